#!/bin/bash
# Full validation + measurement pass on the MI355X box: parity tests, smoke, bench (with CPU baseline + accuracy leg),
# rocprofv3 kernel stats of the same bench command, PMC passes (separate runs) for HBM traffic of the dominant kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r01f}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest_gpu.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 ) > $O/bench.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 2>&1 | tail -1 ) > $O/bench_bf16.log
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o $T -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -2 ) > $O/rocprof.log
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > $O/pmc_fetch.log
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > $O/pmc_write.log
cd $R
find $O -name '*kernel_trace*' -size +30M -delete
tail -6 $O/pytest_gpu.log; cat $O/smoke.log; cat $O/bench.log; python - <<PY
import json
d = json.loads(open("$O/bench_bf16.log").read().strip().splitlines()[-1]); print("bf16", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["config"]["final_loss"])
PY
