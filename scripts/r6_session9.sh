#!/bin/bash
# Round 6, GPU call 9: L2 touches (tile chains, pair chains) and the weight-gradient kernel's fourth raw set: bitwise tests, same-box A/Bs.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s9; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pair_chains.py -x -q > $O/pytest_pair.log 2>&1; tail -3 $O/pytest_pair.log
timeout 600 python3 scripts/ab_forms.py --stages color_forward,sdf_gradient,color_backward --out $O/ab_forms.json > $O/ab_forms.log 2>&1; grep "^color_\|^sdf_" $O/ab_forms.log
cp dynhor_amd/libdynhor_hip.so dynhor_amd/libdynhor_hip_touch.so
for round in 1 2; do
  for n in notouch touch dw4; do
    timeout 600 python3 scripts/ab_stage.py --lib dynhor_amd/libdynhor_hip_$n.so --reps 20 --out $O/ab_${n}_r$round.json > $O/ab_${n}_r$round.log 2>&1
    echo "== $n round $round"; grep -E "^(sdf_|color_|weight_|grad checksum)" $O/ab_${n}_r$round.log | sed "s/'median_ms': //; s/'min_ms'.*//" | tr '\n' ' '; echo
  done
done
timeout 300 python3 scripts/pair_stamps.py --out $O/pair_stamps_color_fwd.json > $O/pair_stamps.log 2>&1; python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6s9/pair_stamps_color_fwd.json"))
print({k: (round(v) if isinstance(v, float) else v) for k, v in d.items() if not isinstance(v, (dict, list))}); print({k: round(v) for k, v in d['prologue_parts'].items()})
PY
