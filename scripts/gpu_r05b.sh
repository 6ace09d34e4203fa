#!/bin/bash
# Round 5: SQ counters of the 64-byte-piece NT kernel, the float64-adjudicated loss curve, the > 4 GiB table test.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 900 python -m pytest tests/test_large_table_gpu.py "tests/test_g1shape_parity_gpu.py::test_loss_curve_g1_shape_and_hitrate" -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 ) > $O/pytest_curve_large.log
bash scripts/h2_pmc.sh > $O/h2_sq_counters.txt 2>&1
tail -30 $O/pytest_curve_large.log; cat $O/h2_sq_counters.txt | cut -c1-200
