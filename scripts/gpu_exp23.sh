#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
timeout 300 python tests/bench_gemm_ramp.py 2>&1 | tail -8
