#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -4 ) > $O/pytest_part.log
cat $O/pytest_part.log
i=0
for cfg in "0 16" "1 16" "1 32" "1 64" "0 16" "1 32" "1 64"; do
  set -- $cfg; i=$((i+1))
  ( CHAM_TAIL_ON_SIDE=$1 CHAM_W2_SPLITS=$2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench$i.log
  python - <<PY
import json
d = json.loads(open("$O/bench$i.log").read().strip().splitlines()[-1]); print("tail_on_side=$1 w2_splits=$2", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["config"]["final_loss"])
PY
done
