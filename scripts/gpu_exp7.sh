#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/pytest_gpu.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 2>&1 | tail -3 ) > $O/bench_bf16.log
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o r01bf16 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 2>&1 | tail -3 ) > $O/rocprof_bf16.log
cd $R
tail -25 $O/pytest_gpu.log; for f in bench bench_bf16; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.log").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["config"]["final_loss"])
except Exception as e: print("$f", "ERR", e, open("$O/$f.log").read()[-800:])
PY
done
