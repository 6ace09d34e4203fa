#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
export PYTHONPATH=$R
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest_gpu.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e -o r01e -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 ) > $O/rocprof.log
( timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_clk -o clk -- python $R/tests/bench_gemm.py 2 2>&1 | tail -6 ) > $O/pmc_clk.log
cd $R
tail -8 $O/pytest_gpu.log; python - <<PY
import json
d = json.loads(open("$O/bench.log").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline"]["achieved"])
PY
cat $O/pmc_clk.log
