#!/bin/bash
# Round 5, GPU call 1: the irreproducible two-piece aux body of the weight-gradient kernel.
#   (a) scripts/micro/hz_*_micro: the product's own kernel source on synthetic operands, one binary per variant -> failure table
#   (b) whole-library variants (no packed fp32 / two-piece aux body / both): stage times on one box, det_dw.py soak of the aux body
# gpurun -- 'bash scripts/r5_hazard_session.sh [launches]'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hz
N=${1:-20000}
for v in base ds scalar pkplain vis nopk visnopk lgkmA nopA lgkmL; do
  timeout 300 scripts/micro/hz_${v}_micro $N 256 4096 3 > gpurun_out/hz/${v}_all.json 2> gpurun_out/hz/${v}_all.err
  echo "== $v all-aux: $(cat gpurun_out/hz/${v}_all.json)"; head -4 gpurun_out/hz/${v}_all.err
done
for v in ds vis scalar; do
  timeout 300 scripts/micro/hz_${v}_micro 8000 24 4096 3 > gpurun_out/hz/${v}_mixed.json 2> gpurun_out/hz/${v}_mixed.err
  echo "== $v mixed (24 aux workgroups + main jobs): $(cat gpurun_out/hz/${v}_mixed.json)"; head -3 gpurun_out/hz/${v}_mixed.err
done
cp dynhor_amd/libdynhor_hip.so dynhor_amd/libdynhor_hip_base.so
bash scripts/ab_libs.sh base nopk nopkaux aux2 > gpurun_out/hz/ab_libs.log 2>&1
python3 scripts/ab_table.py base nopk nopkaux aux2 | tee gpurun_out/hz/ab_table.txt
timeout 900 python3 scripts/det_dw.py 40000 dynhor_amd/libdynhor_hip_aux2.so 2 > gpurun_out/hz/det_dw_aux2.log 2>&1; tail -3 gpurun_out/hz/det_dw_aux2.log
timeout 900 python3 scripts/det_dw.py 100000 dynhor_amd/libdynhor_hip_nopkaux.so 2 > gpurun_out/hz/det_dw_nopkaux.log 2>&1; tail -3 gpurun_out/hz/det_dw_nopkaux.log
timeout 600 python3 scripts/det_chain.py 3000 --out gpurun_out/hz/det_chain_base_3000.json > gpurun_out/hz/det_chain_base.log 2>&1; tail -2 gpurun_out/hz/det_chain_base.log | cut -c1-1500
