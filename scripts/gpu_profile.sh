#!/bin/bash
# rocprofv3 kernel stats of the bench command (DTYPE=f32|bf16), summary copied to gpurun_out/prof_<TAG>_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r02}
D=${DTYPE:-f32}
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --dtype $D --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg ${BENCH_ARGS:-} 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled_$T.log
cd $R
f=$(find $O/prof_$T -name '*kernel_stats.csv' | head -1)
cp $f $O/prof_${T}_kernel_stats.csv 2>/dev/null
find $O/prof_$T -name '*kernel_trace*' -size +30M -delete
head -40 $O/prof_${T}_kernel_stats.csv | cut -c1-260
