#!/bin/bash
# Development: rocprofv3 kernel stats of the hash family's step on packed rays (occupancy-grid sampler), one stream
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_occ
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_occ -- python3 $GRAFT_REPO_ROOT/bench.py --family hash --hash-sampler occgrid --serial-weight-grads $PROF_ARGS --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > /tmp/prof_occ.out 2> /tmp/prof_occ.err
tail -3 /tmp/prof_occ.err
f=$(ls /tmp/prof_occ/*/*kernel_stats.csv | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/s10; cp $f $GRAFT_REPO_ROOT/gpurun_out/s10/occ_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dh::" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
