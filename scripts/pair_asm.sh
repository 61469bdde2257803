#!/bin/bash
# Development tool: compile csrc/chain_pair.hip to a device listing (/tmp/pair.s) and print each kernel's spill counts.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/dynhor_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -S --cuda-device-only chain_pair.hip -o /tmp/pair.s 2>&1 | grep -v "packed-fp32\|hip-link" | grep -B2 -A6 "error" | head -40
grep "vgpr_spill_count\|sgpr_spill_count\|\.name:" /tmp/pair.s | tr '\n' ' ' | sed 's/\.name:/\n/g' | cut -c1-150; echo
