#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/pytest_gpu.log
( CHAM_RNN_LDS_HOG=120000 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_hog.log
( CHAM_SIDE_PRIORITY=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_prio0.log
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d -o r01d -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 ) > $O/rocprof.log
( timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_clk -o clk -- python -m tests.bench_gemm 2 2>&1 | tail -6 ) > $O/pmc_clk.log
cd $R
cat $O/pytest_gpu.log | tail -15; for f in bench_hog bench_prio0; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.log").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["roofline"]["achieved"])
except Exception as e: print("$f", "ERR", e, open("$O/$f.log").read()[-500:])
PY
done; cat $O/pmc_clk.log
