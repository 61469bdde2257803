#!/bin/bash
# Round 6, session 2, call A: the training forward's MFMA transposition against the LDS-patch form (same box: whole workspace compared,
# both timed), the GPU suite on the new library, a short bench line.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s2; mkdir -p $O
timeout 600 python3 scripts/ab_fwdtrain.py --a dynhor_amd/libdynhor_hip_patch.so --b dynhor_amd/libdynhor_hip.so --out $O/r06_ab_fwdtrain_mfma_save_dot.json > $O/ab_fwdtrain.log 2>&1; tail -7 $O/ab_fwdtrain.log | cut -c1-700
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python3 bench.py --steps 100 --warmup 20 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
