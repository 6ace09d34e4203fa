#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
timeout 300 python -c "import bench, json; print('compact', json.dumps(bench.hitrate_parity(42)['hitrate_at_5']))" 2>&1 | tail -1
CHAM_COMPACT=0 timeout 300 python -c "import bench, json; print('padded', json.dumps(bench.hitrate_parity(42)['hitrate_at_5']))" 2>&1 | tail -1
timeout 300 python -c "import bench, json; print('compact seed 7', json.dumps(bench.hitrate_parity(7)))" 2>&1 | tail -1
