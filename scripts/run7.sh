#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
cd $R
( timeout 400 python scripts/boundary_profile.py g1 2>&1 | tail -25 ) > $O/run7_boundary.txt


cd $R


rm -rf $O/run7_trace
cat $O/run7_boundary.txt; head -12 $O/run7_timeline.txt
