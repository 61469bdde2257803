#!/bin/bash
# Round 6, last session: PSNR at equal iterations on the FINAL library (the training forward now saves hi + lo instead of the fp32 activation):
# 8 fresh paired seeds HIP (split_f16) vs the GPU-eager oracle, 2000 iterations, 2048 rays -- the protocol of profiles/psnr_parity_r05_*.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6psnr
timeout 3300 python3 scripts/psnr_parity.py --mode hip_vs_oracle --seeds ${1:-501,502,503,504,505,506,507,508} --out gpurun_out/r6psnr/psnr_parity_r06_neus_hip_vs_oracle_f16_${2:-a}.json > gpurun_out/r6psnr/psnr_${2:-a}.log 2>&1
tail -3 gpurun_out/r6psnr/psnr_${2:-a}.log | cut -c1-600
