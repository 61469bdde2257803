#!/bin/bash
# Development tool: alternated same-box stage timings of variant libraries (scripts/build_variant.sh).
#   gpurun -- 'bash scripts/ab_libs.sh base v1 v3'
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
for round in 1 2; do
  for n in "$@"; do
    timeout 600 python3 $R/scripts/ab_stage.py --lib dynhor_amd/libdynhor_hip_$n.so --reps 20 --out gpurun_out/ab_${n}_r$round.json > $R/gpurun_out/ab_${n}_r$round.log 2>&1
    echo "== $n round $round"; grep -E "^(sdf_|color_|weight_|grad checksum)" $R/gpurun_out/ab_${n}_r$round.log | sed "s/'median_ms': //; s/'min_ms'.*//"
  done
done
