#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 600 python -m pytest tests/test_gemm_b16_gpu.py tests/test_gemm_b16_dma_gpu.py -m gpu -q -x 2>&1 | tail -3 ) > $O/run28_pytest.log
( timeout 600 python -m pytest tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py -m gpu -q -x -k "bf16" 2>&1 | tail -3 ) >> $O/run28_pytest.log
for i in 1 2; do
  echo "bf16: $(timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"], d.get("g1_like_session_lengths",{}).get("value"))')"
done > $O/run28_bf16.txt 2>&1
cat $O/run28_pytest.log $O/run28_bf16.txt
