#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 200 python tests/bench_gemm_sustained.py 1.0 2>&1 | tail -40
echo ---- small-magnitude data
timeout 200 python tests/bench_gemm_sustained.py 0.05 2>&1 | grep -v smi | tail -14
