#!/bin/bash
# Round 5: the chaos floor of the PSNR-at-2k protocol on the shipping arithmetic: 32 pairs HIP(weights x (1 + 1e-7 xi)) - HIP,
# same seeds as the 24 HIP-vs-oracle pairs of round 4 plus 8 more; window and final-checkpoint distributions.  ~40 GPU-minutes.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
SEEDS=$(python3 -c "print(','.join(str(11 * i) for i in range(1, ${1:-32} + 1)))")
python3 scripts/psnr_parity.py --mode hip_noise_floor --seeds $SEEDS --out gpurun_out/psnr_r05_hip_noise_floor_${1:-32}.json > gpurun_out/psnr_r05_hip_noise_floor.log 2>&1
tail -3 gpurun_out/psnr_r05_hip_noise_floor.log | cut -c1-600
