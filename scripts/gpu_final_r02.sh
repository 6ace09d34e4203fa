#!/bin/bash
# Round-2 measurement pass on the MI355X box (the parity suite is run separately: scripts/run_gpu.sh): smoke, the bench lines of every
# arithmetic mode / config, PMC passes (separate runs) for the HBM traffic of the dominant kernels, the 8-rank functional proof and
# the large-catalog stress.  Everything lands in gpurun_out/; the summaries that are judged are copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r02}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke_$T.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' | tail -1 ) > $O/bench_$T.json
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --dtype f32_native 2>&1 | grep '^{' | tail -1 ) > $O/bench_f32_native_$T.json
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 2>&1 | grep '^{' | tail -1 ) > $O/bench_bf16_$T.json
( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --config adressa 2>&1 | grep '^{' | tail -1 ) > $O/bench_adressa_$T.json
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg 2>&1 | tail -2 ) > $O/pmc_fetch_$T.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg --no-boundary-leg 2>&1 | tail -2 ) > $O/pmc_write_$T.log
cd $R
find $O -name '*kernel_trace*' -size +30M -delete
python scripts/pmc_traffic.py $(find $O/pmc_fetch_$T -name '*counter_collection.csv' | head -1) $(find $O/pmc_write_$T -name '*counter_collection.csv' | head -1) > $O/pmc_traffic_$T.json 2>$O/pmc_traffic_$T.err
TAG=$T bash scripts/gpu_dp_proof.sh > $O/dp_proof_$T.log 2>&1
( timeout 400 python scripts/stress_large_catalog.py 2>&1 | tail -1 ) > $O/stress_$T.json
cat $O/smoke_$T.log
python - <<PY
import json
for f in ("bench", "bench_f32_native", "bench_bf16", "bench_adressa"):
    try:
        d = json.loads(open("$O/%s_$T.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["config"]["final_loss"], d.get("g1_like_session_lengths", {}).get("value"))
        for k in ("through_boundary", "through_boundary_g1_like_session_lengths", "cpu_baseline", "accuracy_vs_cpu_ref"):
            if k in d: print("   ", k, json.dumps(d[k])[:300])
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $O/dp_proof_$T.jsonl; head -c 600 $O/stress_$T.json
