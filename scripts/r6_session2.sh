#!/bin/bash
# Round 6, GPU call 2: the tile-PAIR chain forms (bitwise tests against the tile forms, same-process A/B timings), the hash tests on the
# re-scaled fixed-point scatter, the golden-fixture replay + keep_only tests, the gradient-noise diagnosis of the hash family, one bench line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s2; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pair_chains.py -x -q -s > $O/pytest_pair.log 2>&1; tail -6 $O/pytest_pair.log
timeout 600 python3 scripts/ab_forms.py --stages color_forward,sdf_gradient,color_backward --out $O/ab_forms.json > $O/ab_forms.log 2>&1; tail -8 $O/ab_forms.log
timeout 600 python3 -m pytest tests/test_gpu_golden_replay.py -x -q -s > $O/pytest_golden.log 2>&1; tail -8 $O/pytest_golden.log
timeout 900 python3 -m pytest tests/test_gpu_hash_reproducible.py tests/test_gpu_hash_family.py tests/test_gpu_hashgrid.py tests/test_gpu_occgrid.py -q -s > $O/pytest_hash.log 2>&1; tail -5 $O/pytest_hash.log
grep -h "target \|float-atomic launches\|table gradient rel L2" $O/pytest_hash.log
timeout 600 python3 scripts/hash_grad_noise.py --iters 0,500,2000 --out $O/hash_grad_noise.json > $O/hash_grad_noise.log 2>&1; grep "^iter" $O/hash_grad_noise.log
timeout 400 python3 bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6s2/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, d.get("parity_check"))
except Exception as e:
    print("bench missing", e); print(open("gpurun_out/r6s2/bench.err").read()[-2000:])
PY
