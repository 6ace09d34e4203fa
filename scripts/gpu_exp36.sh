#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_step_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -4 ) > $O/pytest_part.log
cat $O/pytest_part.log
for i in 1 2 3 4 5 6; do
  if [ $((i % 2)) -eq 0 ]; then export CHAM_WS1_AFTER_DM=0; else export CHAM_WS1_AFTER_DM=1; fi
  ( timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench$i.log
done
python - <<PY
import json
for i in (1, 2, 3, 4, 5, 6):
    d = json.loads(open("$O/bench%d.log" % i).read().strip().splitlines()[-1]); r = d["roofline"]
    print("bench run=%d" % (i % 2), d["value"], d["ms_per_step"], r["achieved"], d["config"]["final_loss"])
PY
