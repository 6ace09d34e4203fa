#!/bin/bash
# Kernel trace + stats of the bench step (rocprofv3), step timeline (scripts/timeline.py).  TAG=r05x  EXTRA="--length-dist g1"
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${TAG:-r05t}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
BARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc $EXTRA"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $T -- python $R/bench.py $BARGS 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled.json
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
python $R/scripts/timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) k_sel_count_valid full > $O/kernel_trace_step.txt 2>&1
rm -rf $O/prof
head -${HEADN:-30} $O/kernel_stats.csv | cut -c1-150; head -16 $O/kernel_trace_step.txt
