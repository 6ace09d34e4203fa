#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -15 ) > $O/pytest_part.log
cat $O/pytest_part.log
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -o g1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 ) > $O/rocprof_g1.log
cd $R
python - <<PY
import json, csv
d = json.loads(open("$O/rocprof_g1.log").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k: d["g1_like_session_lengths"][k] for k in ("value", "ms_per_step", "valid_positions_of_padded")})
rows = list(csv.DictReader(open("$O/prof_g1/g1_kernel_stats.csv")))
for r in rows:
    if 'rnn' in r['Name'] or 'gru' in r['Name']:
        print(r['Name'][:40], r['Calls'], 'avg_us', float(r['AverageNs'])/1e3)
PY
