#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_step_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -6 ) > $O/pytest_part.log
cat $O/pytest_part.log
for i in 1 2; do ( timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 ) > $O/bench$i.log; done
python - <<PY
import json
for i in (1, 2):
    d = json.loads(open("$O/bench%d.log" % i).read().strip().splitlines()[-1]); r = d["roofline"]
    print("bench", d["value"], d["ms_per_step"], r["achieved"], r["launches_per_step"], r["avg_launch_ms"], {k: d["g1_like_session_lengths"][k] for k in ("value", "ms_per_step")})
PY
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tl -o tl -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench_tl.log
cd $R
python scripts/timeline.py $O/prof_tl/tl_kernel_trace.csv k_sel_count_valid full | awk '$1+0 >= 12.0 || /step wall|GEMM-/' | head -60
