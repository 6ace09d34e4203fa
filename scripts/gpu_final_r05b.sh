#!/bin/bash
# Round 5, last pass (the GPU budget of the round was nearly spent): smoke, the default bench line (its roofline.traffic measured live), rocprofv3
# kernel stats + step timeline of the same command - and only then the -m gpu suite, so that a budget cut costs the
# suite's tail and not the measurements.  The other artifacts of scripts/gpu_final_r05.sh (PMC table of every kernel, SQ counters, stress /
# bf16 / shard profiles, host profile) stay from the pass before: the build differs from theirs by the fused dgrad's plane products only.
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${TAG:-r05}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke.log
( timeout 600 python bench.py 2>$O/bench.err | grep '^{' | tail -1 ) > $O/bench.json
cd /tmp
BARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $T -- python $R/bench.py $BARGS 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled.json
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
python $R/scripts/timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) k_sel_count_valid full > $O/kernel_trace_step.txt 2>&1
rm -rf $O/prof
cd $R
cat $O/smoke.log
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"]["final_loss"])
for k in ("g1_like_session_lengths", "through_boundary", "through_boundary_g1_like_session_lengths", "cpu_baseline", "native_fp32_mfma_arm", "bf16_arm", "adressa_arm", "stress_arm", "dp_self_exchange_ms"):
    if k in d: print("   ", k, json.dumps(d[k])[:200])
PY
head -8 $O/kernel_stats.csv | cut -c1-140; head -2 $O/kernel_trace_step.txt
# the files that run the changed kernels first, then the rest of the suite; the log is written as it goes (a budget cut keeps what ran)
FIRST="tests/test_dm_fused_gpu.py tests/test_gemm_h2_gpu.py tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py tests/test_estimator_gpu.py tests/test_dp_gpu.py tests/test_dp_rccl_gpu.py tests/test_launch_contract_gpu.py tests/test_config5_parity_gpu.py"
timeout 1200 python -u -m pytest $FIRST tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -v amdgpu.ids $O/pytest_gpu.log | tail -4
