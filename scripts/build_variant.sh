#!/bin/bash
# Development tool: build an alternative library with extra compile flags, for same-box A/B runs through
# scripts/ab_stage.py --lib.   bash scripts/build_variant.sh NAME [-DFLAG ...]  ->  dynhor_amd/libdynhor_hip_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/dynhor_amd/csrc
N=$1; shift
mkdir -p $C/build_$N
for f in $C/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o $C/build_$N/$(basename $f).o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dynhor_amd/libdynhor_hip_$N.so $C/build_$N/*.o
echo built $R/dynhor_amd/libdynhor_hip_$N.so
