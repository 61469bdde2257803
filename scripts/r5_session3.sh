#!/bin/bash
# Round 5, GPU call 3: phase stamps of the fp16 chains, A/B of the weight-chunk preload, GPU tests of the current tree.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
bash scripts/stamps.sh run > gpurun_out/s3/stamps.log 2>&1; cp gpurun_out/chain_phase_stamps.json gpurun_out/s3/ 2>/dev/null; grep -A40 "^color_forward {" gpurun_out/s3/stamps.log | head -80
cp dynhor_amd/libdynhor_hip.so dynhor_amd/libdynhor_hip_ship.so
bash scripts/ab_libs.sh wpre0 ship "$@" > gpurun_out/s3/ab_libs.log 2>&1
python3 scripts/ab_table.py wpre0 ship "$@" | tee gpurun_out/s3/ab_table.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest_gpu.log 2>&1; tail -5 gpurun_out/s3/pytest_gpu.log
