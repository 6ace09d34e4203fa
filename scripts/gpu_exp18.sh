#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 300 python tests/probe/run_probe.py 2>&1 | tail -10 ) > $O/probe.log
( timeout 300 python tests/bench_gemm.py 2 4 0 2>&1 | tail -14 ) > $O/gemm_valu.log
( timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_step_gpu.py -m gpu -q 2>&1 | tail -4 ) > $O/pytest_part.log
( timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log
cat $O/probe.log $O/gemm_valu.log $O/pytest_part.log; python - <<PY
import json
d = json.loads(open("$O/bench.log").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["achieved"])
PY
