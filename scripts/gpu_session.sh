#!/bin/bash
# one gpurun call: GPU test suite + bench lines + PSNR script smoke (development, round 2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest.log )
tail -8 gpurun_out/gputest.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_neus.json 2> gpurun_out/bench_neus.err; tail -c 1500 gpurun_out/bench_neus.json; tail -3 gpurun_out/bench_neus.err
timeout 600 python bench.py --family hash --steps 30 --warmup 5 > gpurun_out/bench_hash.json 2> gpurun_out/bench_hash.err; tail -c 1200 gpurun_out/bench_hash.json; tail -3 gpurun_out/bench_hash.err
for m in hip_vs_oracle hip_vs_hip_f32 hip_noise_floor; do
  timeout 600 python scripts/psnr_parity.py --mode $m --seeds 11 --iters 60 --eval-iters 40,60 --frames 8 --lockstep 20 --cross-check --out gpurun_out/psnr_smoke_$m.json > gpurun_out/psnr_smoke_$m.log 2>&1; tail -2 gpurun_out/psnr_smoke_$m.log | cut -c1-900
done
timeout 600 python scripts/psnr_parity.py --family hash --mode hip_scatter --seeds 11 --iters 60 --eval-iters 40,60 --frames 8 --out gpurun_out/psnr_smoke_hash.json > gpurun_out/psnr_smoke_hash.log 2>&1; tail -2 gpurun_out/psnr_smoke_hash.log | cut -c1-900
