#!/bin/bash
# Round 5, final GPU call: the whole GPU suite on the final library, the profile + bench-line session (scripts/gpu_session.sh), two
# 2000-iteration hash-family training runs per sampler compared bit for bit (scripts/det_soak.py), the hash family's lock-step
# against the oracle with the fixed-point table scatter.
cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log
bash scripts/gpu_session.sh > $O/gpu_session.log 2>&1; tail -12 $O/gpu_session.log
timeout 600 python3 scripts/det_soak.py 2000 - hash hierarchical > $O/det_soak_hash_hier.log 2>&1; tail -2 $O/det_soak_hash_hier.log
timeout 600 python3 scripts/det_soak.py 2000 - hash occgrid > $O/det_soak_hash_occ.log 2>&1; tail -2 $O/det_soak_hash_occ.log
timeout 1500 python3 scripts/psnr_parity.py --family hash --mode hip_vs_oracle --seeds 11 --iters 1000 --eval-iters 1000 --lockstep 50 --out $O/psnr_parity_r05_hash_lockstep.json > $O/hash_lockstep.log 2>&1; tail -3 $O/hash_lockstep.log
