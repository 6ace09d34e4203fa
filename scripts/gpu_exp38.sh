#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_step_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -12 ) > $O/pytest_part.log
cat $O/pytest_part.log
i=0
for cfg in 1 0 1 0; do
  i=$((i+1))
  ( CHAM_FUSE_MULPRED=$cfg timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 ) > $O/bench$i.log
  python - <<PY
import json
d = json.loads(open("$O/bench$i.log").read().strip().splitlines()[-1]); print("fuse=$cfg", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["g1_like_session_lengths"]["ms_per_step"], d["config"]["final_loss"])
PY
done
