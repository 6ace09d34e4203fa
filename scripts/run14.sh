#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests/test_step_gpu.py tests/test_estimator_gpu.py tests/test_g1shape_parity_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > $O/run14_pytest.log
for v in 1 0 1 0; do
  echo "CHAM_DEFER_W2=$v: $(CHAM_DEFER_W2=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm 2>$O/run14_err_$v.txt | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"], d.get("g1_like_session_lengths",{}).get("value"), [(g["kernel"][:34], g["avg_launch_ms"]) for g in d["roofline"]["top_gemms"]])')"
done > $O/run14_defer.txt 2>&1
cat $O/run14_pytest.log $O/run14_defer.txt; tail -3 $O/run14_err_1.txt
