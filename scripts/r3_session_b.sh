#!/bin/bash
# Round-3 GPU call: dW job-group A/B (variant libraries of scripts/build_variant.sh), then the register-resident no-grad chain
# against the shipping one (values + time).  Everything lands under gpurun_out/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/ab_libs.sh g1 g8 g4 g15 g8a6 g8a14 > gpurun_out/ab_dw_groups.log 2>&1
tail -60 gpurun_out/ab_dw_groups.log
timeout 300 python scripts/ab_nograd.py --b dynhor_amd/libdynhor_hip_nt.so --out gpurun_out/ab_nograd.json > gpurun_out/ab_nograd.log 2>&1
echo "ab_nograd exit $?"; tail -12 gpurun_out/ab_nograd.log
