#!/bin/bash
# Round 6, GPU call 10: pair-chain tests on the build without touches; hash family: the HIP kernels' gradients stepped by torch.optim.Adam against
# the fused dh_adam_step (hybrid arm, 24 paired seeds at HIP speed).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s10; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pair_chains.py tests/test_gpu_golden_replay.py -x -q > $O/pytest_pair.log 2>&1; tail -3 $O/pytest_pair.log
timeout 600 python3 scripts/ab_forms.py --stages color_forward --out $O/ab_forms.json > $O/ab_forms.log 2>&1; grep "^color_" $O/ab_forms.log
S=$(python3 -c "print(','.join(str(i) for i in range(601, 625)))")
timeout 2400 python3 scripts/psnr_parity.py --family hash --mode hip_torch_adam --seeds $S --out $O/psnr_r06_hash_hip_torch_adam_vs_fused.json > $O/psnr_hybrid.log 2>&1; tail -1 $O/psnr_hybrid.log | cut -c1-900
