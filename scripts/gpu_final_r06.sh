#!/bin/bash
# Round-6 measurement pass on ONE MI355X box, ONE build: smoke, PMC passes (separate runs) for the HBM traffic of every kernel, the default
# bench line (arms, CPU baseline, hip_graph_arm), rocprofv3 kernel stats + step timeline of the same command, the bf16 / stress profiles,
# SQ counters + micro-benchmark of the plane GEMMs (row-major and tile-blocked), the A/B arms of this round's switches, the emulated
# rank-of-8 shard (eager and hipGraph replay) with its kernel trace, and LAST the -m gpu suite (its log is written as it goes).
# Everything lands in gpurun_out/$T/; scripts/collect_profiles_r06.sh copies what is judged to profiles/r06_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${TAG:-r06}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke.log
PMCARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc --graph off"
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python $R/bench.py $PMCARGS 2>&1 | tail -2 ) > $O/pmc_fetch.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python $R/bench.py $PMCARGS 2>&1 | tail -2 ) > $O/pmc_write.log
cd $R
PMC_GENERATED_BY=scripts/gpu_final_r06.sh python scripts/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) > $O/pmc_traffic.json 2>$O/pmc_traffic.err
rm -rf $O/pmc_fetch $O/pmc_write
( timeout 1500 python bench.py 2>$O/bench.err | grep '^{' | tail -1 ) > $O/bench.json
cd /tmp
BARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc --graph off"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $T -- python $R/bench.py $BARGS 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled.json
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
python $R/scripts/timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) k_sel_count_valid full > $O/kernel_trace_step.txt 2>&1
rm -rf $O/prof
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stress -o stress -- python $R/scripts/stress_large_catalog.py --steps 2 2>&1 | grep '^{' | tail -1 ) > $O/stress_profiled.json
cp $(find $O/prof_stress -name '*kernel_stats.csv' | head -1) $O/stress_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_stress
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o bf16 -- python $R/bench.py $BARGS --dtype bf16 2>&1 | grep '^{' | tail -1 ) > $O/bf16_profiled.json
cp $(find $O/prof_bf16 -name '*kernel_stats.csv' | head -1) $O/bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_bf16
cd $R
bash scripts/h2_pmc.sh > $O/h2_sq_counters.txt 2>&1
( timeout 300 python scripts/bench_h2_blocked.py 2>&1 | grep -v amdgpu.ids ) > $O/h2_blocked_microbench.txt
( timeout 300 python scripts/gemm_breakdown.py 2>&1 | grep -v amdgpu.ids ) > $O/gemm_breakdown.txt
( timeout 300 python scripts/host_profile.py 200 2>&1 | head -48 ) > $O/host_profile.txt
# A/B arms of this round's switches through the default bench step (alternating, two repetitions)
AB="CHAM_H2_BLOCKED=0 CHAM_H2_BLOCKED=1 CHAM_GEMM_TN_SMALL=0 CHAM_DEV_SCALARS=0" REPS=2 TAG=$T/ab BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc --no-ragged-leg --graph off" bash scripts/gpu_ab.sh > $O/ab_arms.txt 2>&1
( for a in "" "--graph"; do timeout 300 python scripts/emulate_rank.py --strong $a 1 8 2>&1 | grep -v amdgpu.ids; done; timeout 300 python scripts/emulate_rank.py 1 8 2>&1 | grep emulated ) > $O/emulated_rank_of_n.txt
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -o shard -- python $R/scripts/emulate_rank.py --strong 8 2>&1 | grep emulated ) > $O/shard_profiled.txt
cd $R
cp $(find $O/prof_shard -name '*kernel_stats.csv' | head -1) $O/shard_kernel_stats.csv 2>/dev/null
TIMELINE_MIN_GAP_NS=1000000 python scripts/timeline.py $(find $O/prof_shard -name '*kernel_trace.csv' | head -1) k_sel_count_valid full > $O/shard_kernel_trace_step.txt 2>&1
rm -rf $O/prof_shard
cat $O/smoke.log
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"]["final_loss"])
print("    hip_graph", json.dumps(d["config"].get("hip_graph"))[:400])
for k in ("g1_like_session_lengths", "through_boundary", "through_boundary_g1_like_session_lengths", "cpu_baseline", "native_fp32_mfma_arm", "bf16_arm", "adressa_arm", "stress_arm", "dp_self_exchange_ms"):
    if k in d: print("   ", k, json.dumps(d[k])[:300])
PY
head -12 $O/kernel_stats.csv | cut -c1-160; head -3 $O/kernel_trace_step.txt; cat $O/emulated_rank_of_n.txt | cut -c1-200; tail -12 $O/ab_arms.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1700 python -u -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
  grep -v amdgpu.ids $O/pytest_gpu.log | tail -6
  for f in A B C; do [ -s gpurun_out/loss_curve_200_$f.json ] && cp gpurun_out/loss_curve_200_$f.json $O/loss_curve_200_$f.json; done
  [ -s gpurun_out/loss_curve_bf16_50.json ] && cp gpurun_out/loss_curve_bf16_50.json $O/loss_curve_bf16_50.json
fi
