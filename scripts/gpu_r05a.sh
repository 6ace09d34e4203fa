#!/bin/bash
# Round 5, first contact: the 64-byte-piece NT kernel (unit tests, micro-benchmark and step A/B against round 4's kernel), the > 4 GiB table test.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R; export TMPDIR=/tmp PYTHONPATH=$R
( timeout 700 python -m pytest tests/test_gemm_h2_gpu.py -x -q 2>&1 | tail -15 ) > $O/pytest_h2.log
( H2_ONLY=1 H2_NT_WIDE=1 timeout 200 python -m tests.bench_gemm_h2 2>&1 | grep -v amdgpu.ids ) > $O/h2_micro_wide.txt
( H2_ONLY=1 H2_NT_WIDE=0 timeout 200 python -m tests.bench_gemm_h2 2>&1 | grep -v amdgpu.ids ) > $O/h2_micro_narrow.txt
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg --no-arms --no-native-arm --no-pmc"
for w in 1 0 1 0; do
  ( CHAM_H2_NT_WIDE=$w timeout 300 python bench.py $BARGS 2>$O/bench_wide$w.err | grep '^{' | tail -1 ) >> $O/bench_wide$w.jsonl
done
( timeout 900 python -m pytest tests/test_large_table_gpu.py -x -q 2>&1 | tail -25 ) > $O/pytest_large.log
tail -5 $O/pytest_h2.log; cat $O/h2_micro_wide.txt $O/h2_micro_narrow.txt; tail -8 $O/pytest_large.log
python - <<PY
import json
for w in (1, 0):
    for line in open("$O/bench_wide%d.jsonl" % w):
        line = line.strip()
        if not line: continue
        d = json.loads(line)
        print("wide", w, d["value"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], [(g["kernel"][:34], g["avg_launch_ms"]) for g in d["roofline"]["top_gemms"]], d.get("g1_like_session_lengths", {}).get("value"))
PY
