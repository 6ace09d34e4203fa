#!/bin/bash
# Last check of a tree: the whole GPU suite, smoke(), and the default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r03}
mkdir -p $O
cd $R
export TMPDIR=/tmp PYTHONPATH=$R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12; echo "pytest rc ${PIPESTATUS[0]}" ) > $O/pytest_gpu_$T.log
( timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 ) > $O/smoke_$T.log
( timeout 1200 python bench.py 2>$O/bench_$T.err | grep '^{' | tail -1 ) > $O/bench_$T.json
tail -3 $O/pytest_gpu_$T.log; cat $O/smoke_$T.log
python -c "
import json; d=json.loads(open('$O/bench_$T.json').read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], d['config']['final_loss'], d['native_fp32_mfma_arm']['ms_per_step'])
print(d['g1_like_session_lengths']['value'], d['through_boundary']['value'], d['through_boundary_g1_like_session_lengths']['value'], d['bf16_arm']['value'], d['bf16_arm'].get('g1_like_session_lengths'), d['adressa_arm']['value'], d['dp_self_exchange_ms'], d['cpu_baseline']['value'], d['accuracy_vs_cpu_ref']['hitrate_at_5'])"
