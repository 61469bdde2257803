#!/bin/bash
# hash family, HIP vs eager oracle, the four seeds the earlier call's time limit cut off (12 min each at 1024 rays)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3300 python scripts/psnr_parity.py --family hash --mode hip_vs_oracle --batch 1024 --seeds 33,44,55,66 --out gpurun_out/psnr_parity_r02_hash_hip_vs_oracle_b.json > gpurun_out/psnr_c1.log 2>&1
tail -1 gpurun_out/psnr_c1.log | cut -c1-1500
