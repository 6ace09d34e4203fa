#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( H2_ONLY=1 timeout 200 python -m tests.bench_gemm_h2 2>&1 | grep -v amdgpu.ids ) > $O/h2_micro.txt
( timeout 900 python -m pytest tests/test_gemm_h2_gpu.py tests/test_gemm_gpu.py tests/test_gemm_x3_gpu.py tests/test_gemm_b16_gpu.py tests/test_gemm_b16_dma_gpu.py tests/test_gemm_p3_gpu.py "tests/test_g1shape_parity_gpu.py::test_step_parity_g1_shape_headline_batch" "tests/test_g1shape_parity_gpu.py::test_step_parity_g1_shape_bf16" -x -q 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O/pytest.log
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"
for i in 1 2; do ( timeout 300 python bench.py $BARGS 2>/dev/null | grep '^{' | tail -1 ) >> $O/bench_f32.jsonl; ( timeout 300 python bench.py $BARGS --dtype bf16 2>/dev/null | grep '^{' | tail -1 ) >> $O/bench_bf16.jsonl; done
cat $O/h2_micro.txt; tail -8 $O/pytest.log
python - <<PY
import json
for fn in ("bench_f32", "bench_bf16"):
    for line in open("$O/%s.jsonl" % fn):
        if line.strip():
            d = json.loads(line); print(fn, d["value"], d["ms_per_step"], [(g["kernel"][:30], g["avg_launch_ms"]) for g in d["roofline"]["top_gemms"]], d.get("g1_like_session_lengths", {}).get("value"))
PY
