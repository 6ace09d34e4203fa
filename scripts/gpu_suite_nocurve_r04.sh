#!/bin/bash
# the `-m gpu` suite without the two loss-curve tests (those: scripts/gpu_curve200_r04.sh) - confirmation run on the last build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
( timeout 700 python -m pytest tests -x -q -m gpu -k "not loss_curve" --durations=8 2>&1 | tail -22; echo "pytest rc ${PIPESTATUS[0]}" ) > gpurun_out/r04/pytest_gpu_last_build.log
tail -16 gpurun_out/r04/pytest_gpu_last_build.log
