#!/bin/bash
# round 4, GPU call F: cooperative recurrent kernels (unit tests, step tests that now route ragged batches through them, ragged bench legs),
# then the 200-step curve with the re-stated criteria
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python -m pytest tests/test_rnn_coop_gpu.py -x -q -s > $O/rnn_coop.log 2>&1; U=$?; echo "rnn coop rc $U"; tail -6 $O/rnn_coop.log
if [ $U -eq 0 ]; then
  timeout 900 python -m pytest tests/test_step_gpu.py tests/test_estimator_gpu.py tests/test_state_gpu.py tests/test_dp_gpu.py -x -q > $O/step.log 2>&1; echo "step/estimator rc $?"; tail -4 $O/step.log
  timeout 300 python -m pytest tests/test_g1shape_parity_gpu.py -x -q -k "parity_g1_shape and g1" > $O/g1.log 2>&1; echo "g1 ragged parity rc $?"; tail -3 $O/g1.log
fi
for c in 131072 0; do
  timeout 300 python - > $O/bench_coop_$c.json 2> $O/bench_coop_$c.err <<PY
import sys, json, subprocess, os
sys.argv = ["bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-boundary-leg", "--no-native-arm", "--no-arms"]
import chameleon_recsys_amd.nar.nar_model as M
orig = M.NARRuntime.__init__
def init(self, *a, **k):
    orig(self, *a, **k)
    if $c == 0: self.rnn_coop_rows = -1
M.NARRuntime.__init__ = init
import runpy
runpy.run_path("bench.py", run_name="__main__")
PY
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_coop_$c.json") if l.startswith("{")][-1])
    print("rnn_coop_rows=$c full", d["value"], d["ms_per_step"], "ragged", d.get("g1_like_session_lengths", {}).get("value"), d.get("g1_like_session_lengths", {}).get("ms_per_step"), "timeouts", d["config"].get("rnn_coop_spin_timeouts"))
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_coop_$c.err").read()[-800:])
PY
done
( timeout 300 python scripts/emulate_rank.py 1 8 2>&1 | tail -2; timeout 300 python scripts/emulate_rank.py --strong 8 2>&1 | tail -1 ) > $O/emulated_rank.txt; cat $O/emulated_rank.txt | cut -c1-300
timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py -q -s -k "loss_curve_200" > $O/curve200.log 2>&1; echo "curve200 rc $?"; grep -E "passed|failed|loss curve|running-max|HitRate|Error|assert " $O/curve200.log | tail -8
