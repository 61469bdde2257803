#!/bin/bash
# Round 5, GPU call 5: the paired-group colour forward kernel: parity tests, A/B against the two-workgroup kernel, phase stamps with ONE
# workgroup per CU (what a chain's GEMM costs without a co-resident partner).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 900 python3 -m pytest tests/test_gpu_mlp_forward.py tests/test_gpu_render_forward.py tests/test_gpu_render_backward.py tests/test_gpu_edge_cases.py tests/test_gpu_bench_config.py tests/test_gpu_range_safety.py tests/test_gpu_pose_refinement.py tests/test_gpu_arithmetic_modes.py -x -q > gpurun_out/s5/pytest.log 2>&1; tail -6 gpurun_out/s5/pytest.log
cp dynhor_amd/libdynhor_hip.so dynhor_amd/libdynhor_hip_ship.so
bash scripts/ab_libs.sh pair0 ship > gpurun_out/s5/ab_libs.log 2>&1
python3 scripts/ab_table.py pair0 ship | tee gpurun_out/s5/ab_table.txt
python3 scripts/ab_stage.py --lib dynhor_amd/libdynhor_hip_stamps1.so --stamps-h --blocks 256 --reps 8 --out gpurun_out/s5/stamps_one_wg_per_cu.json > gpurun_out/s5/stamps1.log 2>&1
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/s5/stamps_one_wg_per_cu.json"))
for k, v in d["stamps_h"].items():
    print(k, {a: round(b["mean"]) for a, b in v.items() if isinstance(b, dict) and "mean" in b})
print({k: round(v["median_ms"], 3) for k, v in d["stages"].items()})
PY
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > gpurun_out/s5/bench_quick.json 2> gpurun_out/s5/bench_quick.err; head -c 300 gpurun_out/s5/bench_quick.json; echo
python3 scripts/dbg_dpts.py > gpurun_out/s5/dbg_dpts.log 2>&1; tail -30 gpurun_out/s5/dbg_dpts.log
