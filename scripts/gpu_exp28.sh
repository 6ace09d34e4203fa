#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_estimator_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -8 ) > $O/pytest_part.log
cat $O/pytest_part.log
( timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 ) > $O/bench1.log
cd /tmp
( CHAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -o g1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --length-dist g1 2>&1 | grep '^{' | tail -1 ) > $O/rocprof_g1.log
cd $R
python - <<PY
import json, csv
d = json.loads(open("$O/bench1.log").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k: d["g1_like_session_lengths"][k] for k in ("value", "ms_per_step", "valid_positions_of_padded", "padded_T")})
d = json.loads(open("$O/rocprof_g1.log").read().strip().splitlines()[-1]); print("g1 no-overlap under rocprof", d["value"], d["ms_per_step"])
rows = list(csv.DictReader(open("$O/prof_g1/g1_kernel_stats.csv")))
for r in rows[:12]:
    print('%-80s calls %5s total_ms %8.3f avg_us %9.2f' % (r['Name'][:80], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
