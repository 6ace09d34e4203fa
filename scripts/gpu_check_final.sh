#!/bin/bash
# Last check of a tree: the whole GPU suite, smoke(), and a short bench line (no arms / legs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r03}
mkdir -p $O
cd $R
export TMPDIR=/tmp PYTHONPATH=$R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu_$T.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke_$T.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms 2>/dev/null | grep '^{' | tail -1 ) > $O/bench_short_$T.json
tail -3 $O/pytest_gpu_$T.log; cat $O/smoke_$T.log
python -c "
import json; d=json.loads(open('$O/bench_short_$T.json').read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'], d['config']['final_loss'], d['native_fp32_mfma_arm']['ms_per_step'])"
