#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_occgrid.py -q 2>&1 | tail -15 > gpurun_out/occgrid_test.log; cat gpurun_out/occgrid_test.log
timeout 1500 python scripts/psnr_parity.py --mode hip_noise_floor --seeds 11,22,33,44,55,66,77,88 --out gpurun_out/psnr_parity_r02_neus_hip_noise_floor.json > gpurun_out/psnr_b1.log 2>&1
tail -1 gpurun_out/psnr_b1.log | cut -c1-1200
timeout 1200 python scripts/psnr_parity.py --mode hip_vs_hip_f32 --seeds 11,22,33,44 --out gpurun_out/psnr_parity_r02_neus_hip_vs_hip_f32.json > gpurun_out/psnr_b2.log 2>&1
tail -1 gpurun_out/psnr_b2.log | cut -c1-1200
timeout 900 python scripts/psnr_parity.py --family hash --mode hip_noise_floor --seeds 11,22,33,44,55,66,77,88 --out gpurun_out/psnr_parity_r02_hash_hip_noise_floor.json > gpurun_out/psnr_b3.log 2>&1
tail -1 gpurun_out/psnr_b3.log | cut -c1-1200
timeout 1500 python scripts/psnr_parity.py --family hash --mode hip_scatter --seeds 11,22,33,44,55,66,77,88 --out gpurun_out/psnr_parity_r02_hash_hip_scatter.json > gpurun_out/psnr_b4.log 2>&1
tail -1 gpurun_out/psnr_b4.log | cut -c1-1200
