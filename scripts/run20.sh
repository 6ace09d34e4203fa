#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
run() { echo "$1 steps $2: $(env $1 timeout 300 python bench.py --length-dist g1 --steps $2 --warmup 20 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["final_loss"])')"; }
( run A=1 200; run A=2 200; run A=3 200; run A=1 100; run A=2 100; run A=1 140; run A=2 140; run CHAM_OVERLAP=0 200; run CHAM_OVERLAP=0 200; run CHAM_PRESAMPLE=0 200; run CHAM_PRESAMPLE=0 200 ) > $O/run20.txt 2>&1
cat $O/run20.txt
