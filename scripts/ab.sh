#!/bin/bash
# A/B helper: bash scripts/ab.sh "ENV_A" "ENV_B" [repeats]  - alternates two environments of bench.py inside one gpurun call
# (box-to-box spread is ~2 %, larger than most scheduling effects)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
N=${3:-2}
for i in $(seq 1 $N); do
  for cfg in "$1" "$2"; do
    env $cfg timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['final_loss'])"
  done
done
