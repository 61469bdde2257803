#!/bin/bash
# Round 5, GPU call 9: hash family with the table scatter and the small weight-gradient GEMMs on two streams -- tests, same-box A/B against
# one stream, then the family's profile + bench lines (the hash half of scripts/gpu_session.sh).
cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_hash_reproducible.py tests/test_gpu_hash_family.py tests/test_gpu_occgrid.py tests/test_gpu_launch.py -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
 for smp in hierarchical occgrid; do
  for f in "" "--serial-weight-grads"; do
    tag=${smp}${f:+_serial}_$rep
    timeout 600 python3 bench.py --family hash --hash-sampler $smp $f --steps 100 --no-cpu-baseline --no-secondary > $O/bench_hash_$tag.json 2> $O/bench_hash_$tag.err
    python3 - $O/bench_hash_$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {a: b["ms"] for a, b in k.items() if "weight" in a})
PY
  done
 done
done
R=r05
bash scripts/prof.sh --family hash > gpurun_out/prof_hash.log 2>&1
cp gpurun_out/prof_summary.json gpurun_out/${R}_hash_pmc_summary.json
cp $(ls -t gpurun_out/keep/*_kernel_stats.csv | head -1) gpurun_out/${R}_hash_kernel_stats.csv
python3 scripts/make_traffic_json.py gpurun_out/${R}_hash_pmc_summary.json gpurun_out/pmc_traffic_hash.json hash > $O/traffic_hash.log 2>&1; tail -3 $O/traffic_hash.log
cp gpurun_out/pmc_traffic_hash.json profiles/
python3 bench.py --family hash > gpurun_out/${R}_bench_n1_hash.json 2> gpurun_out/bench_hash.err
python3 bench.py --family hash --hash-sampler occgrid --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_n1_hash_occgrid.json 2> gpurun_out/bench_hash_occ.err
for f in ${R}_bench_n1_hash ${R}_bench_n1_hash_occgrid; do echo "== $f"; head -c 330 gpurun_out/$f.json; echo; done
