#!/bin/bash
# Round 5, GPU call 2: round 2 of the repro (which form of the packed instruction; with / without matrix instructions in the step), the GPU
# test suite on the shipping build (no packed fp32, two-piece aux body), aux-job cost A/B, the >= 3e5 / 1e5-launch soaks, one bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
N=${1:-20000}
for v in ds ophi oplo nomfma ophinomfma pkplain; do
  timeout 300 scripts/micro/hz_${v}_micro $N 256 4096 3 > gpurun_out/s2/hz_${v}.json 2> gpurun_out/s2/hz_${v}.err
  echo "== $v: $(cat gpurun_out/s2/hz_${v}.json)"; sed -n 2,3p gpurun_out/s2/hz_${v}.err
done
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest_gpu.log 2>&1; tail -5 gpurun_out/s2/pytest_gpu.log
cp dynhor_amd/libdynhor_hip.so dynhor_amd/libdynhor_hip_ship.so
bash scripts/ab_libs.sh base ship c8 c10 c16 > gpurun_out/s2/ab_libs.log 2>&1
python3 scripts/ab_table.py base ship c8 c10 c16 | tee gpurun_out/s2/ab_table.txt
timeout 1200 python3 scripts/det_dw.py ${2:-300000} dynhor_amd/libdynhor_hip_ship.so 2 > gpurun_out/s2/det_dw_ship.log 2>&1; tail -2 gpurun_out/s2/det_dw_ship.log
timeout 1500 python3 scripts/det_chain.py ${3:-100000} --out gpurun_out/s2/det_chain_ship.json > gpurun_out/s2/det_chain_ship.log 2>&1; tail -1 gpurun_out/s2/det_chain_ship.log | cut -c1-1800
timeout 900 python3 bench.py > gpurun_out/s2/bench_n1.json 2> gpurun_out/s2/bench_n1.err; head -c 1500 gpurun_out/s2/bench_n1.json; echo
