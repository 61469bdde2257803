#!/bin/bash
# Round 6, session 2, call B: the training forward's tile-store cache policy (aux 0 / 1 / 16 against the shipping nt = 2) and ring depth
# (2 / 4 against 3) on the matrix-pipe-transposition kernel -- same box, alternating launches (scripts/ab_fwdtrain.py).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s2b; mkdir -p $O
for n in st0 st1 st16 d2 d4; do
  timeout 300 python3 scripts/ab_fwdtrain.py --a dynhor_amd/libdynhor_hip.so --b dynhor_amd/libdynhor_hip_$n.so --reps 40 --out $O/ab_$n.json > $O/ab_$n.log 2>&1
  echo "== $n: $(grep "'npts': 262144" $O/ab_$n.log | sed "s/.*'ws_max_abs_diff': \([^,]*\),.*'ms_a': \([^,]*\), 'ms_b': \([^,]*\),.*/maxdiff \1 ms_a \2 ms_b \3/")"
done
# the review's soak for the new training forward: 100,000 relaunches, every launch's outputs compared with the first launch's
timeout 900 python3 scripts/det_chain.py 100000 --stages sdf_forward --out $O/r06_det_chain_soak_sdf_forward_100k.json > $O/det_fwd.log 2>&1; tail -1 $O/det_fwd.log | cut -c1-400
