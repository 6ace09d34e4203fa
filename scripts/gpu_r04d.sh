#!/bin/bash
# round 4, GPU call D: LDS epilogue of the h2 NT GEMMs (unit tests, microbench, in-step A/B) x the side-lane ordering arm
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gemm_h2_gpu.py -x -q > $O/h2_unit.log 2>&1; echo "h2 unit rc $?"; tail -3 $O/h2_unit.log
timeout 300 python -m tests.bench_gemm_h2 > $O/h2_microbench.log 2>&1; grep -v amdgpu.ids $O/h2_microbench.log | head -12
for v in 0 1; do for w in 1 0; do
  CHAM_H2_VARIANT=$v CHAM_W2_LAST=$w timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-native-arm --no-arms > $O/bench_v${v}_w${w}.json 2> $O/bench_v${v}_w${w}.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/bench_v${v}_w${w}.json") if l.startswith("{")][-1])
    print("H2_VARIANT=$v W2_LAST=$w", d["value"], d["ms_per_step"], d["roofline"]["frac"], [(g["kernel"][5:30], g["avg_launch_ms"]) for g in d["roofline"].get("top_gemms", [])], d.get("g1_like_session_lengths", {}).get("value"), d["config"]["final_loss"])
except Exception as e:
    print("bench parse failed", e)
PY
done; done
