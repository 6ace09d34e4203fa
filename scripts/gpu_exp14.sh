#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 600 python -m pytest tests/test_step_gpu.py -m gpu -q -k "rnn_variants" 2>&1 | tail -5 ) > $O/pytest_part.log
( timeout 600 python scripts/trainer_throughput.py --state device 2>&1 | tail -2 ) > $O/trainer_device.log
( timeout 600 python scripts/trainer_throughput.py --state host 2>&1 | tail -2 ) > $O/trainer_host.log
cat $O/pytest_part.log $O/trainer_device.log $O/trainer_host.log
