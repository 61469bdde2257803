#!/bin/bash
# Round 5: 8 more paired seeds HIP (final library) vs oracle at 2k iterations + the lock-step on the final library (one gpurun call, ~36 GPU-minutes)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
P="python3 scripts/psnr_parity.py"
$P --mode hip_vs_oracle --seeds 275,286,297,308,319,330,341,352 --cross-check --out gpurun_out/psnr_parity_r05_neus_hip_vs_oracle_f16_d.json > gpurun_out/psnr_r05_d.log 2>&1; tail -2 gpurun_out/psnr_r05_d.log | cut -c1-600
$P --mode hip_vs_oracle --seeds 11 --iters 1000 --eval-iters 1000 --lockstep 50 --out gpurun_out/psnr_parity_r05_neus_lockstep_f16.json > gpurun_out/psnr_r05_lock.log 2>&1; tail -1 gpurun_out/psnr_r05_lock.log | cut -c1-900
