#!/bin/bash
# round 4, GPU call A: first contact of the two-fp16-plane GEMMs + the new parity tests + A/B of the step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm_h2_gpu.py -x -q -s > $O/h2_unit.log 2>&1; U=$?
echo "h2 unit rc $U"; tail -5 $O/h2_unit.log
timeout 300 python -m tests.bench_gemm_h2 > $O/h2_microbench.log 2>&1; echo "microbench rc $?"; cat $O/h2_microbench.log | tail -16
if [ $U -eq 0 ]; then
  timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py -q -s -k "test_step_parity_g1_shape or adressa" > $O/g1shape.log 2>&1; echo "g1shape rc $?"; grep -E "passed|failed|kink|un-pinned|Error|assert" $O/g1shape.log | tail -30
  timeout 600 python -m pytest tests/test_dp_gpu.py tests/test_dp_rccl_gpu.py tests/test_dm_fused_gpu.py tests/test_step_gpu.py -q -x > $O/dp_step.log 2>&1; echo "dp+step rc $?"; tail -5 $O/dp_step.log
  timeout 300 python -m pytest tests/test_gemm_gpu.py -q -k narrow > $O/narrow.log 2>&1; echo "narrow rc $?"; tail -3 $O/narrow.log
fi
for h in 1 0; do
  CHAM_GEMM_H2=$h timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-native-arm --no-arms > $O/bench_h2_$h.json 2> $O/bench_h2_$h.err; echo "bench H2=$h rc $?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_h2_$h.json") if l.startswith("{")][-1])
    print("H2=$h", d["value"], d["ms_per_step"], d["roofline"]["frac"], [(g["kernel"][:40], g["avg_launch_ms"]) for g in d["roofline"].get("top_gemms", [])], d.get("g1_like_session_lengths", {}).get("value"))
except Exception as e:
    print("bench parse failed", e)
PY
done
