#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( H2_ONLY=1 timeout 200 python -m tests.bench_gemm_h2 2>&1 | grep -v amdgpu.ids ) > $O/h2_micro.txt
( timeout 900 python -m pytest tests/test_gemm_h2_gpu.py "tests/test_g1shape_parity_gpu.py::test_step_parity_g1_shape_headline_batch" tests/test_step_gpu.py::test_training_is_bit_reproducible -x -q 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/pytest.log
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"
for i in 1 2 3; do ( timeout 300 python bench.py $BARGS 2>/dev/null | grep '^{' | tail -1 ) >> $O/bench.jsonl; done
cat $O/h2_micro.txt; tail -5 $O/pytest.log
python - <<PY
import json
for line in open("$O/bench.jsonl"):
    if line.strip():
        d = json.loads(line); print("f32", d["value"], d["ms_per_step"], [(g["kernel"][:30], g["avg_launch_ms"], g["frac_of_mfma_peak"]) for g in d["roofline"]["top_gemms"]], d.get("g1_like_session_lengths", {}).get("value"))
PY
