#!/bin/bash
# the whole `-m gpu` suite as the driver runs it, with per-test durations (gpurun_out/r04/pytest_gpu.log)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40; echo "pytest rc ${PIPESTATUS[0]}" ) > gpurun_out/r04/pytest_gpu.log
tail -25 gpurun_out/r04/pytest_gpu.log
