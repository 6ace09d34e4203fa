#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
echo "---- new rule"; timeout 300 python scripts/gemm_breakdown.py 2>&1 | grep -E "ms/step [0-9]|sum of|M=4864 |M=248064  N=1024  K=1024"
echo "---- old rule"; CHAM_GEMM_TILE_BY_AREA=1 timeout 300 python scripts/gemm_breakdown.py 2>&1 | grep -E "ms/step [0-9]|sum of"
echo "---- new rule"; timeout 300 python scripts/gemm_breakdown.py 2>&1 | grep -E "ms/step [0-9]|sum of"
echo "---- old rule"; CHAM_GEMM_TILE_BY_AREA=1 timeout 300 python scripts/gemm_breakdown.py 2>&1 | grep -E "ms/step [0-9]|sum of"
