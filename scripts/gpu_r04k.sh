#!/bin/bash
# round 4, pass k: scale kernels with a capped grid (same-address atomics), rocprofv3 kernel stats of the bf16 arm
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gemm_h2_gpu.py tests/test_step_gpu.py tests/test_dm_fused_gpu.py -x -q -m gpu > $O/k_tests.log 2>&1; echo "tests rc $?"; tail -2 $O/k_tests.log
BARGS="--no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm"
( timeout 300 python bench.py --steps 30 --warmup 5 $BARGS 2>/dev/null | grep '^{' | tail -1 ) > $O/k_bench.json
python - <<PY
import json
d = json.loads(open("$O/k_bench.json").read())
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], "g1-like", d.get("g1_like_session_lengths", {}).get("value"))
PY
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o bf16 -- python $R/bench.py --steps 10 --warmup 3 --dtype bf16 --no-ragged-leg $BARGS 2>&1 | grep '^{' | tail -1 ) > $O/bf16_profiled.json
cp $(find $O/prof_bf16 -name '*kernel_stats.csv' | head -1) $O/bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_bf16
head -8 $O/bf16_kernel_stats.csv | cut -c1-150
( timeout 200 python -m tests.bench_gemm_h2 2>&1 | grep "scale_" )
