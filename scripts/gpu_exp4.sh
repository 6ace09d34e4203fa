#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest_gpu.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log
( CHAM_OVERLAP=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_nooverlap.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --state host 2>&1 | tail -1 ) > $O/bench_hoststate.log
( timeout 420 python scripts/stress_large_catalog.py --steps 2 2>&1 | tail -3 ) > $O/stress.log
cat $O/pytest_gpu.log; for f in bench bench_nooverlap bench_hoststate; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.log").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["achieved"])
except Exception as e: print("$f", "ERR", e, open("$O/$f.log").read()[-500:])
PY
done; cat $O/stress.log
