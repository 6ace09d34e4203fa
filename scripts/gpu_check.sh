#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line, rocprofv3 kernel stats.  Run via gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest_gpu.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -5 ) > $O/smoke.log
( timeout 400 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 ) > $O/bench.log
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${TAG:-r01} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 ) > $O/rocprof.log
cd $R
find $O/prof -name '*kernel_trace*' -size +30M -delete
cat $O/pytest_gpu.log $O/smoke.log $O/bench.log
