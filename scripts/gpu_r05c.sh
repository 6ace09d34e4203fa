#!/bin/bash
# Round 5: schedule A/B (third lane for the small weight gradients, head split) + the whole -m gpu suite on this build.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"
for rep in 1 2; do
for cfg in "1 1" "0 0" "1 0" "0 1"; do
  set -- $cfg
  ( CHAM_WGRAD_AUX=$1 CHAM_HEAD_SPLIT=$2 timeout 300 python bench.py $BARGS 2>$O/bench_$1$2.err | grep '^{' | tail -1 ) >> $O/bench_aux$1_head$2.jsonl
done
done
python - <<PY
import json, glob
for fn in sorted(glob.glob("$O/bench_aux*.jsonl")):
    for line in open(fn):
        if line.strip():
            d = json.loads(line)
            print(fn.split("/")[-1], d["value"], d["ms_per_step"], "g1-like", d.get("g1_like_session_lengths", {}).get("value"), d["config"]["final_loss"])
PY
( timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -25; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
