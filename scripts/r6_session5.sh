#!/bin/bash
# Round 6, GPU call 4: pair forms after the b32 exchange writes / split mid / output-stage rewrite: bitwise tests, A/B, stamps.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pair_chains.py -x -q > $O/pytest_pair.log 2>&1; tail -4 $O/pytest_pair.log
timeout 600 python3 scripts/ab_forms.py --stages color_forward,sdf_gradient,color_backward --out $O/ab_forms.json > $O/ab_forms.log 2>&1; grep "^color_\|^sdf_" $O/ab_forms.log
timeout 300 python3 scripts/pair_stamps.py --out $O/pair_stamps_color_fwd.json > $O/pair_stamps.log 2>&1; python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6s5/pair_stamps_color_fwd.json"))
print({k: (round(v) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (dict, list))})
for l in range(4): print(l, {k: round(v) for k, v in d[f"layer{l}"].items()})
print(d["layer2_phase1_chunks"]); print(d["layer2_phase2_chunks"])
PY
