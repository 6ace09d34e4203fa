#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest_gpu.log
( timeout 300 python tests/bench_gemm.py 5 6 2 2>&1 | tail -30 ) > $O/gemm_v56.log
tail -8 $O/pytest_gpu.log; cat $O/gemm_v56.log
