#!/bin/bash
# round 5: the scorer's first layer on two fp16 planes (cham_gemm_f32x2h) - kernel tests, stand-alone times, A/B through the bench step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05x2h; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 600 python -m pytest tests/test_gemm_x2h_gpu.py tests/test_gemm_x3_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O/pytest_kernel.log; tail -5 $O/pytest_kernel.log
( timeout 300 python scripts/bench_x2h.py 2>&1 | grep -v amdgpu.ids ) > $O/microbench.txt; cat $O/microbench.txt
AB="CHAM_S1_H2=0 CHAM_S1_H2=1" REPS=3 TAG=r05x2h/ab bash scripts/gpu_ab.sh 2>&1 | tail -8
( timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/pytest_parity.log; tail -6 $O/pytest_parity.log
