#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest_gpu.log
( timeout 300 python tests/bench_gemm.py 2 102 4 104 0 100 2>&1 | tail -30 ) > $O/gemm_pipe.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log
tail -12 $O/pytest_gpu.log; cat $O/gemm_pipe.log; python - <<PY
import json
d = json.loads(open("$O/bench.log").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["config"]["final_loss"])
PY
