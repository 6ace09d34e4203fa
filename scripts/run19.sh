#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 600 python -m pytest tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py -m gpu -q -x -k "g1 or ragged or compaction or curve" 2>&1 | tail -3 ) > $O/run19_pytest_default.log
( CHAM_W2_MAIN_ROWS=131072 timeout 600 python -m pytest tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py -m gpu -q -x -k "g1 or ragged or compaction or reproducible" 2>&1 | tail -3 ) > $O/run19_pytest.log
for v in 0 131072 0 131072; do
  echo "CHAM_W2_MAIN_ROWS=$v: $(CHAM_W2_MAIN_ROWS=$v timeout 300 python bench.py --length-dist g1 --steps 200 --warmup 20 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"])')"
done > $O/run19.txt 2>&1
cat $O/run19_pytest_default.log $O/run19_pytest.log $O/run19.txt
