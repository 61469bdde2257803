#!/bin/bash
# Round 6, final GPU call B: the whole GPU suite, smoke(), relaunch soaks of the final library (the pair colour forward: 20,000 launches; every chain
# stage in its default form: 5,000; the pair forms of the two non-default stages: 5,000; the weight-gradient GEMM: 20,000).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6final2; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python3 scripts/det_chain.py 20000 --stages color_forward --out $O/r06_det_chain_soak_color_forward_pair.json > $O/det_pair.log 2>&1; tail -1 $O/det_pair.log | cut -c1-400
timeout 1500 python3 scripts/det_chain.py 5000 --stages sdf_forward,sdf_gradient,color_backward,sdf_tangent,sdf_backward,sdf_nograd,sdf_gradient_pair,color_backward_pair --out $O/r06_det_chain_soak.json > $O/det_chain.log 2>&1; tail -1 $O/det_chain.log | cut -c1-900
timeout 900 python3 scripts/det_dw.py 20000 > $O/det_dw.log 2>&1; tail -2 $O/det_dw.log | cut -c1-400
