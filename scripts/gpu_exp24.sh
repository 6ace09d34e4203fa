#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
( timeout 900 python -m pytest tests/test_step_gpu.py tests/test_state_gpu.py tests/test_estimator_gpu.py tests/test_dp_gpu.py -m gpu -q 2>&1 | tail -25 ) > $O/pytest_part.log
( timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -3 ) > $O/bench1.log
( CHAM_COMPACT=0 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_nocompact.log
cat $O/pytest_part.log
for f in bench1 bench_nocompact; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.log").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("g1_like_session_lengths"))
except Exception as e:
    print("$f", e, open("$O/$f.log").read()[-1500:])
PY
done
