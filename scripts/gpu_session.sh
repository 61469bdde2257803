#!/bin/bash
# one gpurun call: full GPU test suite + every bench line + rocprofv3 profiles (round 2 measurement session)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest.log )
tail -6 gpurun_out/gputest.log
timeout 900 python bench.py --steps 100 --warmup 20 > gpurun_out/r02_bench_n1.json 2> gpurun_out/bench_neus.err; tail -c 600 gpurun_out/r02_bench_n1.json
timeout 600 python bench.py --steps 60 --warmup 10 --arithmetic fp32_mfma --no-cpu-baseline > gpurun_out/r02_bench_n1_fp32_mfma.json 2>/dev/null; tail -c 300 gpurun_out/r02_bench_n1_fp32_mfma.json
timeout 600 python bench.py --steps 60 --warmup 10 --loss full --no-cpu-baseline > gpurun_out/r02_bench_n1_full_loss.json 2> gpurun_out/bench_full.err; tail -c 300 gpurun_out/r02_bench_n1_full_loss.json; tail -2 gpurun_out/bench_full.err
timeout 600 python bench.py --family hash --steps 100 --warmup 20 > gpurun_out/r02_bench_n1_hash.json 2>/dev/null; tail -c 700 gpurun_out/r02_bench_n1_hash.json
timeout 600 python bench.py --family hash --hash-sampler occgrid --steps 100 --warmup 40 > gpurun_out/r02_bench_n1_hash_occgrid.json 2> gpurun_out/bench_occ.err; tail -c 900 gpurun_out/r02_bench_n1_hash_occgrid.json; tail -2 gpurun_out/bench_occ.err
timeout 300 python scripts/host_overhead.py > gpurun_out/host_overhead_neus.log 2>&1; tail -1 gpurun_out/host_overhead_neus.log
timeout 300 python scripts/host_overhead.py hash > gpurun_out/host_overhead_hash.log 2>&1; tail -1 gpurun_out/host_overhead_hash.log
bash scripts/prof.sh > gpurun_out/prof_sh.log 2>&1; tail -3 gpurun_out/prof_sh.log
mkdir -p gpurun_out/neus_prof; cp gpurun_out/prof_summary.json gpurun_out/neus_prof/ 2>/dev/null; cp gpurun_out/keep/*kernel_stats.csv gpurun_out/neus_prof/ 2>/dev/null
bash scripts/prof.sh --family hash > gpurun_out/prof_sh_hash.log 2>&1
mkdir -p gpurun_out/hash_prof; cp gpurun_out/prof_summary.json gpurun_out/hash_prof/ 2>/dev/null; cp gpurun_out/keep/*kernel_stats.csv gpurun_out/hash_prof/ 2>/dev/null
ls gpurun_out/neus_prof gpurun_out/hash_prof
