#!/bin/bash
# A/B of environment switches WITH a rocprofv3 kernel-stats table per arm.  usage: AB="VAR=a VAR=b" TAG=r06b [PYTEST="tests/..."]
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${TAG:-r06ab}; O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
if [ -n "$PYTEST" ]; then cd $R; ( timeout 1500 python -m pytest $PYTEST -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/pytest.log; tail -12 $O/pytest.log; fi
BARGS=${BARGS:-"--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc --no-ragged-leg"}
for rep in $(seq 1 ${REPS:-2}); do
  for arm in $AB; do
    cd $R
    ( env $(echo $arm | tr '+' ' ') timeout 300 python bench.py $BARGS 2>$O/$arm.err | grep '^{' | tail -1 ) >> $O/$arm.jsonl
  done
done
for arm in $AB; do
  cd /tmp
  ( env $(echo $arm | tr '+' ' ') timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$arm -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/profiled_$arm.json
  f=$(find $O/prof_$arm -name '*kernel_stats.csv' | head -1)
  cp $f "$O/${arm}_kernel_stats.csv" 2>/dev/null
  rm -rf $O/prof_$arm
done
cd $R
python - <<PY
import json, glob, csv
for fn in sorted(glob.glob("$O/*.jsonl")):
    for line in open(fn):
        if line.strip():
            d = json.loads(line)
            print("%-44s %9.1f %6.3f ms  loss %s  gemms %s" % (fn.split("/")[-1][:-6], d["value"], d["ms_per_step"], d["config"]["final_loss"], [g["avg_launch_ms"] for g in d["roofline"]["top_gemms"]]))
tabs = {}
for fn in sorted(glob.glob("$O/*_kernel_stats.csv")):
    arm = fn.split("/")[-1][:-len("_kernel_stats.csv")]
    for r in csv.DictReader(open(fn)):
        tabs.setdefault(r["Name"][:70], {})[arm] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
arms = sorted({a for v in tabs.values() for a in v})
rows = sorted(tabs.items(), key=lambda kv: -max(c * t for c, t in kv[1].values()))[:28]
print("%-72s" % "kernel (avg us x calls)" + "".join("%28s" % a[-26:] for a in arms))
for name, v in rows:
    print("%-72s" % name + "".join(("%18.1f x %-7d" % (v[a][1], v[a][0])) if a in v else "%28s" % "-" for a in arms))
PY
