#!/bin/bash
# Round 5, GPU call 4: DPP wave maximum A/B + stamps, the GPU test suite, then the 32-pair PSNR chaos floor.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
bash scripts/stamps.sh run > gpurun_out/s4/stamps.log 2>&1; cp gpurun_out/chain_phase_stamps.json gpurun_out/s4/ 2>/dev/null
cp dynhor_amd/libdynhor_hip.so dynhor_amd/libdynhor_hip_ship.so
bash scripts/ab_libs.sh shfl ship > gpurun_out/s4/ab_libs.log 2>&1
python3 scripts/ab_table.py shfl ship | tee gpurun_out/s4/ab_table.txt
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/s4/pytest_gpu.log 2>&1; tail -8 gpurun_out/s4/pytest_gpu.log
bash scripts/r5_psnr_noise.sh 32
