#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
( timeout 900 python -m pytest tests/test_state_gpu.py tests/test_step_gpu.py tests/test_dp_gpu.py tests/test_estimator_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^INFO" | tail -6 )
for i in 1 2; do
for cfg in CHAM_PRESAMPLE=1 CHAM_PRESAMPLE=0; do
    env $cfg timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['g1_like_session_lengths']['value'], d['g1_like_session_lengths']['ms_per_step'], d['config']['final_loss'])"
done; done
