#!/bin/bash
# Round 5, GPU call 6: the hash family's reproducible (fixed-point) table scatter -- tests, cost -- and nerfacc's grid refresh schedule
# against the every-cell default (8 paired seeds each, occupancy-grid sampler vs hierarchical sampler).
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_hash_reproducible.py tests/test_gpu_occgrid.py tests/test_gpu_hash_family.py tests/test_gpu_launch.py -x -q -s > $O/pytest.log 2>&1; tail -8 $O/pytest.log
grep -h "float-atomic launches\|table gradient rel L2\|occupied .* of" $O/pytest.log
for f in "" "--float-atomic-table-grad"; do   # (round 6: the flag that exists; the default IS the reproducible form)
  for smp in hierarchical occgrid; do
    tag=${smp}${f:+_float}
    timeout 600 python3 bench.py --family hash --hash-sampler $smp $f --steps 100 --no-cpu-baseline --no-secondary > $O/bench_hash_$tag.json 2> $O/bench_hash_$tag.err
    python3 - $O/bench_hash_$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("per_kernel_ms") or d.get("kernels") or {}
print(sys.argv[1], d["value"], d["ms_per_step"], {a: b for a, b in k.items() if "weight" in a} if isinstance(k, dict) else "")
PY
  done
done
S=11,22,33,44,55,66,77,88
timeout 1500 python3 scripts/psnr_parity.py --family hash --mode hip_occgrid_vs_hierarchical --seeds $S --grid-refresh nerfacc --out $O/psnr_r05_hash_occgrid_nerfacc_refresh_vs_hierarchical.json > $O/psnr_nerfacc.log 2>&1; tail -3 $O/psnr_nerfacc.log
timeout 1500 python3 scripts/psnr_parity.py --family hash --mode hip_occgrid_vs_hierarchical --seeds $S --out $O/psnr_r05_hash_occgrid_vs_hierarchical.json > $O/psnr_all.log 2>&1; tail -3 $O/psnr_all.log
python3 - <<'PY'
import json
for n in ("psnr_r05_hash_occgrid_nerfacc_refresh_vs_hierarchical", "psnr_r05_hash_occgrid_vs_hierarchical"):
    try:
        d = json.load(open(f"gpurun_out/s6/{n}.json"))
        print(n, "window", {k: round(v, 3) if isinstance(v, float) else v for k, v in d["window_delta"].items() if k != "per_seed"},
              "final", {k: round(v, 3) if isinstance(v, float) else v for k, v in d["final_delta"].items() if k != "per_seed"})
    except Exception as e:
        print(n, "missing", e)
PY
