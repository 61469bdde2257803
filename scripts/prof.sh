#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes of the bench (run on the GPU box via gpurun); outputs under gpurun_out/prof_*
# Counters are collected in their own passes (no --kernel-trace / --stats beside --pmc), FETCH_SIZE and WRITE_SIZE in separate
# ones (TCC slots) -- /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots".  usage: prof.sh [extra bench.py args]
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof_*
ARGS="$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --kernel-steps 2 $ARGS > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --kernel-steps 2 $ARGS > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --kernel-steps 2 $ARGS > $OUT/prof_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/prof_pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --kernel-steps 2 $ARGS > $OUT/prof_pmc3.log 2>&1
# L2 side of the weight stream: requests the CUs' L1s send to L2 and the L2 hit rate
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --kernel-steps 2 $ARGS > $OUT/prof_pmc4.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/prof_pmc5 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --kernel-steps 2 $ARGS > $OUT/prof_pmc5.log 2>&1
python3 $R/scripts/prof_summarize.py $OUT > $OUT/prof_summary.log 2>&1
mkdir -p $OUT/keep; cp $OUT/prof_stats/*/*kernel_stats.csv $OUT/keep/ 2>/dev/null; cp $OUT/prof_stats/*/*domain_stats.csv $OUT/keep/ 2>/dev/null
rm -rf $OUT/prof_stats $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_pmc3 $OUT/prof_pmc4 $OUT/prof_pmc5
ls -la $OUT $OUT/keep
tail -5 $OUT/prof_summary.log
