#!/bin/bash
# round 4, pass l: weight-shadow refresh and row grouping on the side lane at the head of a step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_estimator_gpu.py tests/test_state_gpu.py tests/test_dp_gpu.py tests/test_dp_rccl_gpu.py tests/test_config5_parity_gpu.py tests/test_g1shape_parity_gpu.py -x -q -m gpu -k "not loss_curve" > $O/l_tests.log 2>&1; echo "tests rc $?"; tail -2 $O/l_tests.log
BARGS="--no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm"
for i in 1 2; do
( timeout 300 python bench.py --steps 30 --warmup 5 $BARGS 2>/dev/null | grep '^{' | tail -1 ) > $O/l_bench_$i.json
python - <<PY
import json
d = json.loads(open("$O/l_bench_$i.json").read())
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], "g1-like", d.get("g1_like_session_lengths", {}).get("value"), d["config"]["final_loss"])
PY
done
( timeout 300 python scripts/emulate_rank.py --strong 8 2>&1 | tail -1 ) | cut -c1-250
( timeout 300 python bench.py --steps 20 --warmup 5 --dtype bf16 $BARGS 2>/dev/null | grep '^{' | tail -1 ) | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'], d.get('g1_like_session_lengths', {}).get('value'))"
