#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1like -o g1 -- python $R/bench.py --length-dist g1 --steps 40 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm 2>&1 | grep '^{' | tail -1 ) > $O/run18_bench.json
cd $R
cp $(find $O/prof_g1like -name '*kernel_stats.csv' | head -1) $O/run18_kernel_stats.csv
f=$(find $O/prof_g1like -name '*kernel_trace.csv' | head -1)
python scripts/timeline.py $f k_sel_count_valid full > $O/run18_timeline.txt 2>&1
rm -rf $O/prof_g1like
python -c "
import json; d=json.loads(open('$O/run18_bench.json').read()); print(d['value'], d['ms_per_step'])"
head -30 $O/run18_kernel_stats.csv | cut -c1-150
