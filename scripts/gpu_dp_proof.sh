#!/bin/bash
# Functional proof of the multi-rank bench path on a ONE-GPU box: 8 ranks share cuda:0 over the gloo transport (RCCL refuses two
# ranks per device) - weak and strong scaling, dense all-reduce and the sparse item-table exchange.  Numbers are NOT scaling
# results (8 processes time-share one GPU); the driver measures those on an 8-GPU node.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r02}
mkdir -p $O
cd $R
export TMPDIR=/tmp PYTHONPATH=$R CHAM_DIST_BACKEND=gloo
: > $O/dp_proof_$T.jsonl
for cfg in "allreduce weak" "sparse weak" "allreduce strong" "sparse strong" "sparse_rs weak"; do
  set -- $cfg
  CHAM_DP_MODE=$1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 8 --steps 4 --warmup 2 --no-cpu-baseline --no-boundary-leg --no-ragged-leg --no-native-arm --no-arms --no-pmc --scaling $2 2>&1 | grep '^{' | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(json.dumps(dict(dp_mode='$1', scaling=d['scaling'], transport='gloo, 8 ranks on ONE MI355X (functional proof, not a scaling result)', n_gpus=d['n_gpus'],
                      global_batch=d['config']['global_batch'], sessions_per_gpu_per_step=d['config']['sessions_per_gpu_per_step'],
                      value=d['value'], ms_per_step=d['ms_per_step'], final_loss=d['config']['final_loss'])))" | tee -a $O/dp_proof_$T.jsonl
done
