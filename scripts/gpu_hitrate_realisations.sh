#!/bin/bash
# HitRate@5 / MRR@5 after the 200-step loss-curve trajectories, per family, under several bit-different but equally valid summation orders of
# the HIP path (environment switches that only change the ORDER of fp32 additions): the spread of the HIP realisations next to the oracle's.
# usage (GPU box): bash scripts/gpu_hitrate_realisations.sh  ->  gpurun_out/hitrate_realisations.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R
cd $R
: > $O/hitrate_realisations.txt
for arm in "CHAM_DGRAD_GROUPSUM=1" "CHAM_DGRAD_GROUPSUM=0" "CHAM_DGRAD_GROUPSUM=1 CHAM_FEATURE_BWD_WS=0" "CHAM_DGRAD_GROUPSUM=0 CHAM_FEATURE_BWD_WS=0" \
           "CHAM_DGRAD_GROUPSUM=1 CHAM_GEMM_TN_SMALL=0" "CHAM_DGRAD_GROUPSUM=0 CHAM_GEMM_TN_SMALL=0"; do
  for fam in A B C; do
    line=$(env $arm timeout 600 python -m pytest "tests/test_g1shape_parity_gpu.py::test_loss_curve_g1_shape_and_hitrate[$fam]" -m gpu -q -s 2>&1 | grep -E "HitRate@5|passed|failed" | tr '\n' ' ' | cut -c1-400)
    echo "[$arm] family $fam: $line" >> $O/hitrate_realisations.txt
  done
done
cat $O/hitrate_realisations.txt
