#!/bin/bash
# one gpurun call: hash-family profile + bench lines (kernel list of the committed stats was stale), final checks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/prof.sh --family hash > gpurun_out/prof_hash.log 2>&1
cp gpurun_out/prof_summary.json gpurun_out/hash_prof_summary.json
cp $(ls -t gpurun_out/keep/*_kernel_stats.csv | head -1) gpurun_out/hash_kernel_stats_new.csv
python bench.py --family hash --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_n1_hash.json 2> gpurun_out/bench_hash.err
python bench.py --family hash --hash-sampler occgrid --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_n1_hash_occgrid.json 2> gpurun_out/bench_hash_occ.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
tail -c 400 gpurun_out/bench_n1_hash.json; echo; tail -c 400 gpurun_out/bench_n1_hash_occgrid.json
