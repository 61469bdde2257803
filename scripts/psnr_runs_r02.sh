#!/bin/bash
# The gpurun calls that produced profiles/psnr_parity_r02_*.json (round 2).  One block = one call (the per-call limit is 3600 s;
# a NeuS-family pair of 2000-iteration runs costs 4 GPU-minutes, a hash-family pair at 1024 rays 12).  Usage on the GPU box:
#   bash scripts/psnr_runs_r02.sh <block>      block in: neus_a neus_b neus_c floors hash_floors hash_a hash_b sampler
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
P="python scripts/psnr_parity.py"
case "$1" in
  neus_a) $P --mode hip_vs_oracle --seeds 11,22,33,44,55,66,77,88 --cross-check --out gpurun_out/psnr_parity_r02_neus_hip_vs_oracle.json
          $P --mode hip_vs_oracle --seeds 11 --iters 1000 --eval-iters 1000 --lockstep 50 --out gpurun_out/psnr_parity_r02_neus_lockstep.json ;;
  neus_b) $P --mode hip_vs_oracle --seeds 99,110,121,132,143,154,165,176 --out gpurun_out/psnr_parity_r02_neus_hip_vs_oracle_b.json ;;
  neus_c) $P --mode hip_vs_oracle --seeds 187,198,209,220,231,242,253,264,275 --out gpurun_out/psnr_parity_r02_neus_hip_vs_oracle_c.json ;;
  floors) $P --mode hip_noise_floor --seeds 11,22,33,44,55,66,77,88 --out gpurun_out/psnr_parity_r02_neus_hip_noise_floor.json
          $P --mode hip_vs_hip_f32 --seeds 11,22,33,44 --out gpurun_out/psnr_parity_r02_neus_hip_vs_hip_f32.json ;;
  hash_floors) $P --family hash --mode hip_noise_floor --seeds 11,22,33,44,55,66,77,88 --out gpurun_out/psnr_parity_r02_hash_hip_noise_floor.json
          $P --family hash --mode hip_scatter --seeds 11,22,33,44,55,66,77,88 --out gpurun_out/psnr_parity_r02_hash_hip_scatter.json ;;
  hash_a) $P --family hash --mode hip_vs_oracle --batch 1024 --seeds 11,22 --out gpurun_out/psnr_parity_r02_hash_hip_vs_oracle_partial.json ;;
  hash_b) $P --family hash --mode hip_vs_oracle --batch 1024 --seeds 33,44,55,66 --out gpurun_out/psnr_parity_r02_hash_hip_vs_oracle_b.json ;;
  sampler) $P --family hash --mode hip_occgrid_vs_hierarchical --seeds 11,22,33,44 --out gpurun_out/psnr_r02_hash_occgrid_vs_hierarchical.json ;;
  *) echo "usage: $0 neus_a|neus_b|neus_c|floors|hash_floors|hash_a|hash_b|sampler"; exit 2 ;;
esac
