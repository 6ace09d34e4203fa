#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_estimator_gpu.py tests/test_gemm_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -40
