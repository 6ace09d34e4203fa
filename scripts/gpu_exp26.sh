#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -o g1 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --length-dist g1 2>&1 | tail -1 ) > $O/rocprof_g1.log
cd $R
tail -1 $O/rocprof_g1.log | cut -c1-300
python scripts/timeline.py $O/prof_g1/g1_kernel_trace.csv k_sel_count_valid full | head -170
