#!/bin/bash
# the whole GPU suite on one box, log under gpurun_out/$TAG/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${TAG:-suite}; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS} 2>&1 | grep -v amdgpu.ids | tail -15; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
