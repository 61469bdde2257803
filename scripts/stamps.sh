#!/bin/bash
# Diagnostic build with per-phase s_memtime stamps in the chain kernels (csrc/stamps.h) + the run that summarises them.
# Build here (no GPU needed), run on the GPU box:  gpurun -- 'bash scripts/stamps.sh run'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/dynhor_amd/csrc
if [ "$1" != "run" ]; then
  mkdir -p $C/build_stamps
  for f in $C/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DDH_STAMPS -c $f -o $C/build_stamps/$(basename $f).o 2> >(grep -v "is not a recognized feature" >&2) & done; wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dynhor_amd/libdynhor_hip_stamps.so $C/build_stamps/*.o
  echo built $R/dynhor_amd/libdynhor_hip_stamps.so
else
  mkdir -p $R/gpurun_out
  python3 $R/scripts/ab_stage.py --lib dynhor_amd/libdynhor_hip_stamps.so --stamps --stamps-h --reps 8 --out gpurun_out/chain_phase_stamps.json
fi
