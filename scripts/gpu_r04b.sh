#!/bin/bash
# round 4, GPU call B: kernel trace + timeline of the h2 step, SQ counters of the h2 kernels, the long loss-curve tests, bf16 x DP
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r04b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ragged-leg --no-boundary-leg --no-arms --no-native-arm 2>&1 | grep '^{' | tail -1 ) > $O/bench_profiled.json
cd $R
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
KT=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/timeline.py $KT k_sel_count_valid full > $O/timeline.txt 2>&1
head -40 $O/timeline.txt
rm -rf $O/prof
head -30 $O/kernel_stats.csv | cut -c1-160
bash scripts/h2_pmc.sh > $O/h2_sq_counters.txt 2>&1; cat $O/h2_sq_counters.txt | head -70
timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py -q -s -k "loss_curve" > $O/loss_curves.log 2>&1; echo "loss curves rc $?"; grep -E "passed|failed|loss curve|HitRate|Error|assert" $O/loss_curves.log | tail -12
timeout 600 python -m pytest tests/test_dp_gpu.py -q -x -k every_arithmetic > $O/dp_arith.log 2>&1; echo "dp arith rc $?"; tail -4 $O/dp_arith.log
