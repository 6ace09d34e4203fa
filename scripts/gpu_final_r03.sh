#!/bin/bash
# Round-3 measurement pass on the MI355X box: parity suite, smoke, the default bench line (with its arms), rocprofv3 kernel stats of the
# same command, PMC passes (separate runs) for the HBM traffic of the dominant kernels, host profile, large-catalog stress.  Everything lands
# in gpurun_out/; the summaries that are judged are copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r03}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu_$T.log
fi
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke_$T.log
( timeout 1200 python bench.py 2>$O/bench_$T.err | grep '^{' | tail -1 ) > $O/bench_$T.json
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled_$T.json
cp $(find $O/prof_$T -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$T.csv 2>/dev/null
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm 2>&1 | tail -2 ) > $O/pmc_fetch_$T.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm 2>&1 | tail -2 ) > $O/pmc_write_$T.log
cd $R
python scripts/pmc_traffic.py $(find $O/pmc_fetch_$T -name '*counter_collection.csv' | head -1) $(find $O/pmc_write_$T -name '*counter_collection.csv' | head -1) > $O/pmc_traffic_$T.json 2>$O/pmc_traffic_$T.err
find $O -name '*kernel_trace*' -size +30M -delete
rm -rf $O/pmc_fetch_$T $O/pmc_write_$T
( timeout 300 python scripts/host_profile.py 200 2>&1 | head -48 ) > $O/host_profile_$T.txt
( timeout 400 python scripts/stress_large_catalog.py 2>&1 | tail -1 ) > $O/stress_$T.json
( timeout 300 python scripts/emulate_rank.py 1 8 2>&1 | tail -2; timeout 300 python scripts/emulate_rank.py --strong 8 2>&1 | tail -1 ) > $O/emulated_rank_$T.txt
tail -3 $O/pytest_gpu_$T.log; cat $O/smoke_$T.log
python - <<PY
import json
d = json.loads(open("$O/bench_$T.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["config"]["final_loss"])
for k in ("g1_like_session_lengths", "through_boundary", "through_boundary_g1_like_session_lengths", "cpu_baseline", "native_fp32_mfma_arm", "bf16_arm", "adressa_arm", "dp_self_exchange", "host_enqueue_ms_per_step"):
    if k in d: print("   ", k, json.dumps(d[k])[:260])
PY
head -8 $O/kernel_stats_$T.csv | cut -c1-200; head -c 500 $O/stress_$T.json; echo; cat $O/emulated_rank_$T.txt | cut -c1-200
