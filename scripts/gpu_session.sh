#!/bin/bash
# one gpurun call (round 2, after the GEMM-core change): profiles of the headline bench + the driver-visible bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/prof.sh > gpurun_out/prof.log 2>&1
python bench.py --steps 100 --warmup 20 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
python bench.py --steps 100 --warmup 20 --arithmetic fp32_mfma --no-cpu-baseline > gpurun_out/bench_n1_fp32_mfma.json 2> gpurun_out/bench_n1_fp32.err
python bench.py --steps 100 --warmup 20 --loss full --no-cpu-baseline > gpurun_out/bench_n1_full_loss.json 2> gpurun_out/bench_n1_full.err
tail -c 300 gpurun_out/bench_n1.json; tail -5 gpurun_out/prof_summary.log
