cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; rm -rf $OUT/prof_*
rocprofv3 -L 2>/dev/null | grep -oE "SQ_(INSTS|ACTIVE_INST|INST_CYCLES|WAIT|BUSY|LDS)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/prof_pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc4.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_pmc5 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc5.log 2>&1
python3 $R/scripts/prof_summarize.py $OUT > $OUT/prof_summary.log 2>&1
rm -rf $OUT/prof_pmc4 $OUT/prof_pmc5
tail -3 $OUT/prof_pmc4.log $OUT/prof_pmc5.log
