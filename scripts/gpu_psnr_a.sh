#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python scripts/psnr_parity.py --mode hip_vs_oracle --seeds 11,22,33,44,55,66,77,88 --cross-check --out gpurun_out/psnr_parity_r02_neus_hip_vs_oracle.json > gpurun_out/psnr_a1.log 2>&1
tail -1 gpurun_out/psnr_a1.log | cut -c1-1500
timeout 600 python scripts/psnr_parity.py --mode hip_vs_oracle --seeds 11 --iters 1000 --eval-iters 1000 --lockstep 50 --out gpurun_out/psnr_parity_r02_neus_lockstep.json > gpurun_out/psnr_a2.log 2>&1
tail -1 gpurun_out/psnr_a2.log | cut -c1-1500
