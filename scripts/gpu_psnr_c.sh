#!/bin/bash
# NeuS family, HIP vs eager oracle: nine more paired seeds on the final kernels (4 min each)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2500 python scripts/psnr_parity.py --mode hip_vs_oracle --seeds 187,198,209,220,231,242,253,264,275 --out gpurun_out/psnr_parity_r02_neus_hip_vs_oracle_c.json > gpurun_out/psnr_c2.log 2>&1
tail -1 gpurun_out/psnr_c2.log | cut -c1-1500
