#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q --tb=short -k "fused" 2>&1 | tail -3 )
cd /tmp
( CHAM_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fu -o fu -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench_fu.log
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_fu/fu_kernel_stats.csv")))
for r in rows[:6]:
    print('%-90s calls %4s avg_us %9.2f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
i=0
for cfg in 1 0 1 0; do
  i=$((i+1))
  ( CHAM_FUSE_MULPRED=$cfg timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 ) > $O/bench$i.log
  python - <<PY
import json
d = json.loads(open("$O/bench$i.log").read().strip().splitlines()[-1]); print("fuse=$cfg", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["g1_like_session_lengths"]["ms_per_step"], d["config"]["final_loss"])
PY
done
