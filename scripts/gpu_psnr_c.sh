#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python scripts/psnr_parity.py --mode hip_vs_oracle --seeds 99,110,121,132,143,154,165,176 --out gpurun_out/psnr_parity_r02_neus_hip_vs_oracle_b.json > gpurun_out/psnr_c0.log 2>&1
tail -1 gpurun_out/psnr_c0.log | cut -c1-1200
timeout 3300 python scripts/psnr_parity.py --family hash --mode hip_vs_oracle --batch 1024 --seeds 11,22,33,44,55,66 --out gpurun_out/psnr_parity_r02_hash_hip_vs_oracle.json > gpurun_out/psnr_c1.log 2>&1
tail -1 gpurun_out/psnr_c1.log | cut -c1-1500
