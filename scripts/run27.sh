#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o bf16 -- python $R/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled_bf16.json
cp $(find $O/prof_bf16 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bf16.csv
rm -rf $O/prof_bf16
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_adr -o adr -- python $R/bench.py --config adressa --steps 6 --warmup 2 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled_adressa.json
cp $(find $O/prof_adr -name '*kernel_stats.csv' | head -1) $O/kernel_stats_adressa.csv
rm -rf $O/prof_adr
head -6 $O/kernel_stats_bf16.csv | cut -c1-160; head -5 $O/kernel_stats_adressa.csv | cut -c1-160
