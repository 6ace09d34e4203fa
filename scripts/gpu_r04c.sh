#!/bin/bash
# round 4, GPU call C: the whole -m gpu suite + smoke on the current tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 1700 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -60; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke.log
tail -45 $O/pytest_gpu.log; cat $O/smoke.log
