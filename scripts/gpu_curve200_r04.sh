#!/bin/bash
# the 200-step loss-curve test alone, with its prints (gpurun_out/r04/curve200.log, gpurun_out/loss_curve_200.json)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py -q -s -k "test_loss_curve_g1_shape_and_hitrate" > gpurun_out/r04/curve200.log 2>&1
echo "rc $?"; grep -E "passed|failed|loss curve|running-max|HitRate|Error|assert |stopped" gpurun_out/r04/curve200.log | tail -10
