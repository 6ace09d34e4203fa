#!/bin/bash
# Full validation + measurement pass on the MI355X box: parity tests, smoke, bench (with CPU baseline + accuracy leg),
# rocprofv3 kernel stats of the same bench command, PMC passes (separate runs) for HBM traffic of the dominant kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r01k}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest_gpu.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' | tail -1 ) > $O/bench.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 2>&1 | grep '^{' | tail -1 ) > $O/bench_bf16.log
( timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ragged-leg --config adressa 2>&1 | grep '^{' | tail -1 ) > $O/bench_adressa.log
( CHAM_COMPACT=0 timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --length-dist g1 2>&1 | grep '^{' | tail -1 ) > $O/bench_g1_padded.log
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled.log
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg 2>&1 | tail -2 ) > $O/pmc_fetch.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged-leg 2>&1 | tail -2 ) > $O/pmc_write.log
cd $R
find $O -name '*kernel_trace*' -size +30M -delete
tail -6 $O/pytest_gpu.log; cat $O/smoke.log; cat $O/bench.log; python - <<PY
import json
for f in ("bench_bf16", "bench_adressa", "bench_g1_padded", "bench_profiled"):
    try:
        d = json.loads(open("$O/%s.log" % f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["config"]["final_loss"])
    except Exception as e:
        print(f, "FAILED", e)
PY
