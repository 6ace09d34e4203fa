#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line (+ optionally rocprofv3 kernel stats with PROFILE=1).  Run via gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r02}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 1200 python -m pytest tests -m gpu -q -s ${PYTEST_ARGS:-} 2>&1 | grep -v "amdgpu.ids\|^\[Gloo\]\|socket.cpp" ) > $O/pytest_$T.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke_$T.log
( timeout 900 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} 2>&1 | grep -v "amdgpu.ids" | tail -30 ) > $O/bench_$T.log
if [ -n "$PROFILE" ]; then
  cd /tmp
  ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled_$T.log
  cd $R
  find $O -name '*kernel_trace*' -size +30M -delete
fi
grep -n "^E  \|passed\|failed\|kink\|adam state\|^FAILED" $O/pytest_$T.log | head -60; cat $O/smoke_$T.log; tail -3 $O/bench_$T.log
