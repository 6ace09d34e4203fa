#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 600 python -m pytest tests/test_gemm_b16_dma_gpu.py -m gpu -q 2>&1 | tail -25 ) > $O/run9_pytest.log
( timeout 300 python tests/bench_gemm_b16_dma.py 2>&1 | tail -5 ) > $O/run9_bench.txt
cat $O/run9_pytest.log $O/run9_bench.txt
