#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
timeout 400 python scripts/insitu_gemm.py 2>&1 | tail -30
