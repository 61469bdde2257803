#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes of the bench (run on the GPU box via gpurun); outputs under gpurun_out/prof_*
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof_*
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/prof_pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc3.log 2>&1
python3 $R/scripts/prof_summarize.py $OUT > $OUT/prof_summary.log 2>&1
mkdir -p $OUT/keep; cp $OUT/prof_stats/*/*kernel_stats.csv $OUT/keep/ 2>/dev/null; cp $OUT/prof_stats/*/*domain_stats.csv $OUT/keep/ 2>/dev/null
rm -rf $OUT/prof_stats $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_pmc3
ls -la $OUT $OUT/keep
tail -5 $OUT/prof_summary.log
