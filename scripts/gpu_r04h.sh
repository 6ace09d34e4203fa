#!/bin/bash
# round 4, GPU call H: bf16 twin of the fused scorer dgrad, mid-step state update / staging, bench legs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python -m pytest tests/test_dm_fused_gpu.py -x -q > $O/dmf.log 2>&1; echo "dm_fused rc $?"; tail -4 $O/dmf.log
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_estimator_gpu.py tests/test_state_gpu.py tests/test_g1shape_parity_gpu.py -x -q -k "not loss_curve and not headline" > $O/step.log 2>&1; echo "step/estimator/g1shape rc $?"; tail -4 $O/step.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["final_loss"], "timeouts", d["config"].get("rnn_coop_spin_timeouts"))
for k in ("g1_like_session_lengths", "through_boundary", "through_boundary_g1_like_session_lengths", "bf16_arm", "adressa_arm", "stress_arm"):
    if k in d: print("   ", k, json.dumps(d[k])[:420])
PY
( timeout 300 python scripts/emulate_rank.py --strong 8 2>&1 | tail -1 ) | cut -c1-250
