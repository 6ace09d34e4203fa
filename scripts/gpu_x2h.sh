#!/bin/bash
# round 5: the scorer's first layer on two fp16 planes (cham_gemm_f32x2h, cham_dm_mulpred_h2h) - kernel tests, stand-alone times, A/B through the bench step
# (CHAM_S1_H2: 0 = six bf16 products everywhere, g = the forward GEMM and the weight gradient on three fp16 products, 1 = the fused dgrad's own products too)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05x2h; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 600 python -m pytest tests/test_gemm_x2h_gpu.py tests/test_dm_fused_gpu.py tests/test_gemm_h2_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O/pytest_kernel.log; tail -5 $O/pytest_kernel.log
( timeout 300 python scripts/bench_x2h.py 2>&1 | grep -v amdgpu.ids ) > $O/microbench.txt; cat $O/microbench.txt
rm -f $O/ab/*.jsonl
AB="CHAM_S1_H2=0 CHAM_S1_H2=g CHAM_S1_H2=1" REPS=${REPS:-3} TAG=r05x2h/ab bash scripts/gpu_ab.sh 2>&1 | tail -10
( timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/pytest_parity.log; tail -6 $O/pytest_parity.log
