#!/bin/bash
# Development tool: build an alternative library with extra compile flags, for same-box A/B runs through
# scripts/ab_stage.py --lib.   bash scripts/build_variant.sh NAME [-DFLAG ...]  ->  dynhor_amd/libdynhor_hip_NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/dynhor_amd/csrc
N=$1; shift
mkdir -p $C/build_$N
# (the shipping flags of __graft_entry__.HIPCC_FLAGS: no packed-fp32 VALU instructions, csrc/layout.h; PK=1 in the environment leaves
# them out -- the irreproducible variants of profiles/r05_dw_aux_hazard_table.json)
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
[ -n "$PK" ] && NOPK=""
for f in $C/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $NOPK "$@" -c $f -o $C/build_$N/$(basename $f).o 2> >(grep -v "is not a recognized feature" >&2) & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dynhor_amd/libdynhor_hip_$N.so $C/build_$N/*.o
echo built $R/dynhor_amd/libdynhor_hip_$N.so
