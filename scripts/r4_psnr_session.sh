#!/bin/bash
# Round-4 PSNR evidence for the two-piece fp16 arithmetic (one gpurun call each; ~4 GPU-minutes per paired seed):
#   bash scripts/r4_psnr_session.sh a     8 paired seeds HIP (split_f16) vs oracle, final checkpoint also evaluated by the oracle's own renderer
#   bash scripts/r4_psnr_session.sh b     8 more seeds
#   bash scripts/r4_psnr_session.sh c     8 more seeds, on the round's FINAL build (chain I/O rework, reproducible weight gradients)
#   bash scripts/r4_psnr_session.sh lock  lock-step (1000 iterations, 20 segments) + 4 seeds split_f16 vs fp32-MFMA kernels
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
P="python scripts/psnr_parity.py"
case "$1" in
  a) $P --mode hip_vs_oracle --seeds 11,22,33,44,55,66,77,88 --cross-check --out gpurun_out/psnr_parity_r04_neus_hip_vs_oracle_f16_a.json ;;
  b) $P --mode hip_vs_oracle --seeds 99,110,121,132,143,154,165,176 --cross-check --out gpurun_out/psnr_parity_r04_neus_hip_vs_oracle_f16_b.json ;;
  c) $P --mode hip_vs_oracle --seeds 187,198,209,220,231,242,253,264 --cross-check --out gpurun_out/psnr_parity_r04_neus_hip_vs_oracle_f16_c.json ;;
  lock) $P --mode hip_vs_oracle --seeds 11 --iters 1000 --eval-iters 1000 --lockstep 50 --out gpurun_out/psnr_parity_r04_neus_lockstep_f16.json
        $P --mode hip_vs_hip_f32 --seeds 11,22,33,44 --out gpurun_out/psnr_parity_r04_neus_hip_f16_vs_hip_f32.json ;;
  *) echo "usage: $0 a|b|c|lock"; exit 2 ;;
esac
