#!/bin/bash
# Round 5, final GPU call on the final library: whole GPU suite, both families' profile + bench lines (scripts/gpu_session.sh + the
# occupancy-grid line), chain re-launch soak of the two register-resident stages whose waits changed (20,000 launches each).
cd $GRAFT_REPO_ROOT
O=gpurun_out/s12; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
bash scripts/gpu_session.sh > $O/gpu_session.log 2>&1; grep -A1 "^== " $O/gpu_session.log | cut -c1-200
python3 bench.py --family hash --hash-sampler occgrid --no-cpu-baseline --no-secondary > gpurun_out/r05_bench_n1_hash_occgrid.json 2> gpurun_out/bench_hash_occ.err; head -c 200 gpurun_out/r05_bench_n1_hash_occgrid.json; echo
timeout 1500 python3 scripts/det_chain.py 20000 --stages sdf_forward,sdf_nograd --out $O/det_chain_20000.json > $O/det_chain_20000.log 2>&1; tail -2 $O/det_chain_20000.log | cut -c1-600
