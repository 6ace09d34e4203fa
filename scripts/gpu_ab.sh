#!/bin/bash
# A/B of environment switches through the default bench step.  usage: AB="VAR=a VAR=b ..." (one arm per word; '+' joins several assignments) REPS=2 TAG=r05x
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${TAG:-r05ab}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
BARGS=${BARGS:-"--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"}
for rep in $(seq 1 ${REPS:-2}); do
  for arm in $AB; do
    ( env $(echo $arm | tr '+' ' ') timeout 300 python bench.py $BARGS 2>$O/$arm.err | grep '^{' | tail -1 ) >> $O/$arm.jsonl
  done
done
python - <<PY
import json, glob
for fn in sorted(glob.glob("$O/*.jsonl")):
    for line in open(fn):
        if line.strip():
            d = json.loads(line)
            print("%-44s %9.1f %6.3f ms  g1-like %s  loss %s" % (fn.split("/")[-1][:-6], d["value"], d["ms_per_step"], d.get("g1_like_session_lengths", {}).get("value"), d["config"]["final_loss"]))
PY
if [ -n "$PYTEST" ]; then ( timeout 1500 python -m pytest $PYTEST -x -q 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/pytest.log; tail -12 $O/pytest.log; fi
