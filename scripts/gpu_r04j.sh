#!/bin/bash
# round 4, pass j: embedding gradient with more loads in flight; exchange-mode tests; stress arm
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04j
timeout 600 python -m pytest tests/test_embgrad_gpu.py tests/test_config5_parity_gpu.py tests/test_dp_rccl_gpu.py -x -q -m gpu > gpurun_out/r04j/tests_a.log 2>&1
echo "tests_a exit $?"; tail -3 gpurun_out/r04j/tests_a.log
timeout 300 python scripts/stress_large_catalog.py --n-items 1000000 --batch 1024 --neg 200 --micro 512 --steps 2 > gpurun_out/r04j/stress.json 2> gpurun_out/r04j/stress.err
echo "stress exit $?"; cat gpurun_out/r04j/stress.json | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('sessions_per_s', 'embedding_gradient')})"
