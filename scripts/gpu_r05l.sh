#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 200 python tests/bench_gemm_b16_dma.py 2>&1 | grep -v amdgpu.ids ) > $O/b16_micro.txt
( timeout 900 python -m pytest tests/test_gemm_b16_dma_gpu.py tests/test_gemm_b16_gpu.py "tests/test_g1shape_parity_gpu.py::test_step_parity_g1_shape_bf16" "tests/test_g1shape_parity_gpu.py::test_loss_curve_50_steps_bf16_g1_shape" tests/test_step_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O/pytest.log
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"
for i in 1 2 3; do ( timeout 300 python bench.py $BARGS --dtype bf16 2>/dev/null | grep '^{' | tail -1 ) >> $O/bench_bf16.jsonl; done
cat $O/b16_micro.txt; tail -6 $O/pytest.log
python - <<PY
import json
for line in open("$O/bench_bf16.jsonl"):
    if line.strip():
        d = json.loads(line); print("bf16", d["value"], d["ms_per_step"], [(g["kernel"][:30], g["avg_launch_ms"], g["frac_of_mfma_peak"]) for g in d["roofline"]["top_gemms"]], d.get("g1_like_session_lengths", {}).get("value"))
PY
