#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 600 python -m pytest tests/test_gemm_p3_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > $O/run15_pytest.log
( timeout 300 python tests/bench_gemm_p3.py 2>&1 | tail -14 ) > $O/run15_bench.txt
cat $O/run15_pytest.log $O/run15_bench.txt
