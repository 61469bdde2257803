#!/bin/bash
# Round 6, GPU call 8: the whole GPU suite on the build with the pair colour forward as the default form + one bench line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s8; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 400 python3 bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6s8/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, d["roofline"]["kernel"], d.get("parity_check", {}).get("first_step_loss_diff"))
except Exception as e:
    print("bench missing", e); print(open("gpurun_out/r6s8/bench.err").read()[-2000:])
PY
