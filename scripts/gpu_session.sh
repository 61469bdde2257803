#!/bin/bash
# The round-6 profile (rounds 4 / 5: the same with R=r04 / r05) + bench-line call (one gpurun call, ~12 GPU-minutes): rocprofv3 stats + PMC passes of both families, the
# traffic tables bench.py reads, then the verbatim bench lines that go to profiles/r04_bench_n1*.json.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r06
bash scripts/prof.sh > gpurun_out/prof_neus.log 2>&1
cp gpurun_out/prof_summary.json gpurun_out/${R}_pmc_summary.json
cp $(ls -t gpurun_out/keep/*_kernel_stats.csv | head -1) gpurun_out/${R}_kernel_stats.csv
python scripts/make_traffic_json.py gpurun_out/${R}_pmc_summary.json gpurun_out/pmc_traffic.json > /dev/null
bash scripts/prof.sh --family hash > gpurun_out/prof_hash.log 2>&1
cp gpurun_out/prof_summary.json gpurun_out/${R}_hash_pmc_summary.json
cp $(ls -t gpurun_out/keep/*_kernel_stats.csv | head -1) gpurun_out/${R}_hash_kernel_stats.csv
python scripts/make_traffic_json.py gpurun_out/${R}_hash_pmc_summary.json gpurun_out/pmc_traffic_hash.json hash > /dev/null
cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_hash.json profiles/       # so that the bench lines below carry roofline.traffic
python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/bench_n1.err
python bench.py --arithmetic split_bf16 --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_n1_split_bf16.json 2> gpurun_out/bench_n1_bf16.err
python bench.py --arithmetic fp32_mfma --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_n1_fp32_mfma.json 2> gpurun_out/bench_n1_fp32.err
python bench.py --family hash > gpurun_out/${R}_bench_n1_hash.json 2> gpurun_out/bench_hash.err
python bench.py --gpus 1 --force-dist --backend nccl --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_n1_rccl_1rank.json 2> gpurun_out/bench_rccl.err
python scripts/host_overhead.py > gpurun_out/host_overhead_neus.log 2>&1
# the one-GPU numbers that price the 8-GPU run: the fixed-global-batch share of a rank at 8 GPUs, with and without the collective
for rpr in 2048 256; do
  python bench.py --rays-per-rank $rpr --no-secondary --no-cpu-baseline > gpurun_out/${R}_bench_n1_rpr$rpr.json 2> gpurun_out/bench_rpr$rpr.err
  python bench.py --rays-per-rank $rpr --no-secondary --no-cpu-baseline --force-dist --backend nccl > gpurun_out/${R}_bench_n1_rpr${rpr}_rccl_1rank.json 2> gpurun_out/bench_rpr${rpr}_rccl.err
done
for f in ${R}_bench_n1 ${R}_bench_n1_split_bf16 ${R}_bench_n1_fp32_mfma ${R}_bench_n1_hash ${R}_bench_n1_rccl_1rank ${R}_bench_n1_rpr2048 ${R}_bench_n1_rpr256; do echo "== $f"; head -c 330 gpurun_out/$f.json; echo; done
