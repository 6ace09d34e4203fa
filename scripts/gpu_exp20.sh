#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench1.log
( CHAM_OVERLAP=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_noov.log
cd /tmp
( CHAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_noov -o noov -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -2 ) > $O/rocprof.log
cd $R
for f in bench1 bench_noov; do python - <<PY
import json
d = json.loads(open("$O/$f.log").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["all_gemm_ms_per_step"])
PY
done
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_noov/noov_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('sum kernel ms/step', tot/1e6/16)
for r in rows[:14]:
    print('%-95s calls/step %5.1f ms/step %7.3f avg_us %9.2f' % (r['Name'][:95], int(r['Calls'])/16, float(r['TotalDurationNs'])/1e6/16, float(r['AverageNs'])/1e3))
PY
