#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tl -o tl -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench_tl.log
cd $R
python scripts/timeline.py $O/prof_tl/tl_kernel_trace.csv k_sel_count_valid full | head -150
