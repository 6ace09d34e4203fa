#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 900 python -m pytest tests/test_g1shape_parity_gpu.py tests/test_step_gpu.py tests/test_dp_gpu.py -m gpu -q -k "bf16 or b16" 2>&1 | tail -8 ) > $O/run10_pytest.log
for v in 0 1; do
  echo "CHAM_B16_DMA=$v: $(CHAM_B16_DMA=$v timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("g1_like_session_lengths",{}).get("value"), [(g["kernel"][:40], g["avg_launch_ms"], g["frac_of_mfma_peak"]) for g in d["roofline"]["top_gemms"]])')"
done > $O/run10_bf16.txt 2>&1
cat $O/run10_pytest.log $O/run10_bf16.txt
