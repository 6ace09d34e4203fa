#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
for i in 1 2; do
( CHAM_DGRAD_NN=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_nt$i.log
( CHAM_DGRAD_NN=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_nn$i.log
done
for f in bench_nt1 bench_nn1 bench_nt2 bench_nn2; do python - <<PY
import json
d = json.loads(open("$O/$f.log").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["achieved"])
PY
done
