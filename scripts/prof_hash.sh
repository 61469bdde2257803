#!/bin/bash
# rocprofv3 kernel-trace stats of the hash-family training iteration (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT/keep
rm -rf $OUT/prof_hash
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hash -- python3 $R/scripts/bench_hash_family.py --steps 10 --warmup 3 > $OUT/prof_hash.log 2>&1
cp $OUT/prof_hash/*/*kernel_stats.csv $OUT/keep/hash_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof_hash
tail -2 $OUT/prof_hash.log
head -25 $OUT/keep/hash_kernel_stats.csv | cut -c1-200
