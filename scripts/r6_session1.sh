#!/bin/bash
# Round 6, GPU call 1: (a) the hash family's tests on the re-scaled fixed-point scatter + the batched-gather oracle, (b) the one-GPU numbers
# that price the 8-GPU run (--rays-per-rank 256, --force-dist --backend nccl at both batch sizes), (c) the first paired
# seeds HIP vs oracle of the hash family with the batched oracle -- each batch its own file, so a time limit keeps the rest.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s1; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_hash_reproducible.py tests/test_gpu_hash_family.py tests/test_gpu_hashgrid.py tests/test_gpu_occgrid.py -x -q -s > $O/pytest_hash.log 2>&1; tail -5 $O/pytest_hash.log
grep -h "target \|float-atomic launches\|table gradient rel L2" $O/pytest_hash.log
for rpr in 2048 256; do
  timeout 300 python3 bench.py --rays-per-rank $rpr --no-secondary --no-cpu-baseline > $O/r06_bench_n1_rpr$rpr.json 2> $O/bench_rpr$rpr.err
  timeout 300 python3 bench.py --rays-per-rank $rpr --no-secondary --no-cpu-baseline --force-dist --backend nccl > $O/r06_bench_n1_rpr${rpr}_rccl_1rank.json 2> $O/bench_rpr${rpr}_rccl.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6s1/r06_bench_n1_rpr*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("scaling"), (d.get("comm") or {}).get("allreduce_ms"))
    except Exception as e:
        print(f, "missing", e)
PY
P="python3 scripts/psnr_parity.py --family hash --mode hip_vs_oracle"
timeout 1300 $P --seeds 401 --out $O/psnr_parity_r06_hash_hip_vs_oracle_a.json > $O/psnr_a.log 2>&1; tail -2 $O/psnr_a.log | cut -c1-700
# how many further seeds fit into ~35 minutes at the oracle arm's measured pace
NS=$(python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6s1/psnr_parity_r06_hash_hip_vs_oracle_a.json"))
    per_seed = 2000 * (d["sec_per_iter_a"] + d["sec_per_iter_b"]) + 60
    print(max(1, min(7, int(2100 / per_seed))))
except Exception:
    print(1)
PY
)
SEEDS=$(python3 -c "print(','.join(str(402 + i) for i in range($NS)))")
echo "batch b: $NS seeds ($SEEDS)"
timeout 2400 $P --seeds $SEEDS --out $O/psnr_parity_r06_hash_hip_vs_oracle_b.json > $O/psnr_b.log 2>&1; tail -2 $O/psnr_b.log | cut -c1-700
