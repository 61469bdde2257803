#!/bin/bash
# one gpurun call: occgrid tests, same-box A/B of two build variants, smoke(), sampler quality report (round 2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_occgrid.py -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for lib in "" "--lib dynhor_amd/libdynhor_hip_nt.so" "--lib dynhor_amd/libdynhor_hip_epi2.so" "" "--lib dynhor_amd/libdynhor_hip_nt.so" "--lib dynhor_amd/libdynhor_hip_epi2.so"; do
  echo "== ab_stage $lib"; timeout 300 python scripts/ab_stage.py $lib --reps 24 2>&1 | grep -E "sdf_tangent|sdf_backward|weight_grads_gemm" | sed 's/mean_ms.*//'
done
timeout 900 python scripts/psnr_parity.py --family hash --mode hip_occgrid_vs_hierarchical --seeds 11,22,33,44 --out gpurun_out/psnr_r02_hash_occgrid_vs_hierarchical.json > gpurun_out/psnr_occ.log 2>&1
tail -1 gpurun_out/psnr_occ.log | cut -c1-1300
