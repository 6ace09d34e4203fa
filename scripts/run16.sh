#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
for v in 0 1 0 1 2; do
  echo "CHAM_P3_VARIANT=$v: $(CHAM_P3_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-ragged-leg --no-arms --no-native-arm 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], [(g["kernel"][:34], g["avg_launch_ms"]) for g in d["roofline"]["top_gemms"]])')"
done > $O/run16_variant.txt 2>&1
cat $O/run16_variant.txt
