#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
for s in 32 14 15 12 24 64; do
  echo "splits $s: $(CHAM_P3_W2_SPLITS=$s timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done > $O/run8_splits.txt 2>&1
cat $O/run8_splits.txt
