#!/bin/bash
# Round 6, GPU call 3: phase stamps of the pair colour forward; the hash tests (guard-band test on a crafted workspace); the hash family's
# gradient-noise diagnosis with the chunked fp64 oracle.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s3; mkdir -p $O
timeout 300 python3 scripts/pair_stamps.py --out $O/pair_stamps_color_fwd.json > $O/pair_stamps.log 2>&1; tail -60 $O/pair_stamps.log
timeout 900 python3 -m pytest tests/test_gpu_hash_reproducible.py -q -s > $O/pytest_hash.log 2>&1; tail -4 $O/pytest_hash.log; grep -h "target " $O/pytest_hash.log
timeout 900 python3 scripts/hash_grad_noise.py --iters 0,500,2000 --out $O/hash_grad_noise.json > $O/hash_grad_noise.log 2>&1; grep "^iter" $O/hash_grad_noise.log
