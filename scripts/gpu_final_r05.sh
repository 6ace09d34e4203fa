#!/bin/bash
# Round-5 measurement pass on the MI355X box: (optionally) the parity suite, smoke, the default bench line (with its arms), rocprofv3 kernel
# stats + step timeline of the same command, PMC passes (separate runs) for the HBM traffic of the kernels, SQ counters of the h2 GEMMs,
# host profile, emulated rank-of-8 shard.  Everything lands in gpurun_out/$T/; scripts/collect_profiles_r05.sh copies what is judged to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${TAG:-r05}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
if [ "${SKIP_TESTS:-1}" != "1" ]; then
  ( timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -12; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu.log
fi
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke.log
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc 2>&1 | tail -2 ) > $O/pmc_fetch.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc 2>&1 | tail -2 ) > $O/pmc_write.log
cd $R
PMC_GENERATED_BY=scripts/gpu_final_r05.sh python scripts/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) > $O/pmc_traffic.json 2>$O/pmc_traffic.err
rm -rf $O/pmc_fetch $O/pmc_write
# the bench below reads the dominant kernel's `traffic` from profiles/*_pmc_traffic.json: this pass's file, so that value, bytes and symbol
# come from one box and one build
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json
( timeout 1500 python bench.py 2>$O/bench.err | grep '^{' | tail -1 ) > $O/bench.json
cd /tmp
BARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm --no-pmc"
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
( timeout 300 python -m tests.bench_gemm_h2 2>&1 | grep -v amdgpu.ids ) > $O/gemm_h2_microbench.txt
( timeout 300 python scripts/host_profile.py 200 2>&1 | head -48 ) > $O/host_profile.txt
( timeout 300 python scripts/emulate_rank.py 1 8 2>&1 | tail -2; timeout 300 python scripts/emulate_rank.py --strong 8 2>&1 | tail -1 ) > $O/emulated_rank_of_n.txt
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -o shard -- python $R/scripts/emulate_rank.py --strong 8 2>&1 | tail -1 ) > $O/shard_profiled.txt
cd $R
cp $(find $O/prof_shard -name '*kernel_stats.csv' | head -1) $O/shard_kernel_stats.csv 2>/dev/null
TIMELINE_MIN_GAP_NS=1500000 python scripts/timeline.py $(find $O/prof_shard -name '*kernel_trace.csv' | head -1) k_sel_count_valid full > $O/shard_kernel_trace_step.txt 2>&1
rm -rf $O/prof_shard
cat $O/smoke.log
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"]["final_loss"])
for k in ("g1_like_session_lengths", "through_boundary", "through_boundary_g1_like_session_lengths", "cpu_baseline", "native_fp32_mfma_arm", "bf16_arm", "adressa_arm", "stress_arm", "dp_self_exchange_ms"):
    if k in d: print("   ", k, json.dumps(d[k])[:300])
PY
head -12 $O/kernel_stats.csv | cut -c1-160; head -3 $O/kernel_trace_step.txt; cat $O/emulated_rank_of_n.txt | cut -c1-200
