#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 600 python -m pytest tests/test_gemm_p3_gpu.py -m gpu -q -x 2>&1 | tail -3 ) > $O/run13_pytest.log
for v in 1 0 1 0; do
  echo "CHAM_P3_TAIL_SPLIT=$v: $(CHAM_P3_TAIL_SPLIT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-ragged-leg --no-arms --no-native-arm 2>$O/run13_err_$v.txt | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], [(g["kernel"][:34], g["avg_launch_ms"], g["frac_of_mfma_peak"]) for g in d["roofline"]["top_gemms"]])')"
done > $O/run13_tail.txt 2>&1
cat $O/run13_pytest.log $O/run13_tail.txt; tail -3 $O/run13_err_1.txt
