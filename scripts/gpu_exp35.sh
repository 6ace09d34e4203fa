#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
timeout 600 python scripts/trainer_throughput.py --hours 4 --sessions-per-hour 5120 --length-dist full 2>&1 | grep '^{' | tail -1
timeout 600 python scripts/trainer_throughput.py --hours 8 --sessions-per-hour 10240 --length-dist g1 2>&1 | grep '^{' | tail -1
