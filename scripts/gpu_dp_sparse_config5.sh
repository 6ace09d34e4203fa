#!/bin/bash
# Reduced BASELINE config 5 on 8 ranks that share ONE MI355X over gloo (functional run): sparse item-table exchange vs the dense
# all-reduce vs the single-process step on the same global batches.  Output: gpurun_out/dp_sparse_config5_<TAG>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${TAG:-r03}
mkdir -p $O
cd $R
export TMPDIR=/tmp PYTHONPATH=$R CHAM_DIST_BACKEND=gloo
timeout 600 python scripts/dp_sparse_config5.py --out $O/dpc5_ref.json > $O/dpc5_ref.log 2>&1
for mode in sparse allreduce; do
  CHAM_DP_MODE=$mode timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 \
    scripts/dp_sparse_config5.py --out $O/dpc5_$mode.json > $O/dpc5_$mode.log 2>&1
done
python - <<PY
import json
O = "$O"
ref, sp, de = (json.load(open("%s/dpc5_%s.json" % (O, k))) for k in ("ref", "sparse", "allreduce"))
def maxdiff(a, b):
    return max(abs(x - y) for u, v in zip(a, b) for x, y in zip(u, v))
out = dict(what="reduced BASELINE config 5 (%d articles x %d, 128-d ACE, %d negatives, global batch %d, full-length sessions): 8 ranks on ONE MI355X over gloo "
                "(functional run, the ranks time-share the device) vs one process running the same global batches as micro-batches"
                % (ref["n_items"], ref["item_embedding"][1], ref["negatives"], ref["global_batch"]),
           flat_parameter_bytes=ref["flat_parameter_bytes"], item_table_bytes=ref["item_table_bytes"],
           sparse=dict(exchange_bytes_per_step=sp["exchange_bytes_per_step"], touched_item_rows_per_step=sp["touched_item_rows_per_step"],
                       loss_per_step=sp["loss_total_xe_reg_per_step"], ms_per_step=sp["ms_per_step"]),
           dense_allreduce=dict(exchange_bytes_per_step=de["exchange_bytes_per_step"], loss_per_step=de["loss_total_xe_reg_per_step"], ms_per_step=de["ms_per_step"]),
           single_process=dict(loss_per_step=ref["loss_total_xe_reg_per_step"], ms_per_step=ref["ms_per_step"]),
           exchange_ratio_dense_over_sparse=round(de["exchange_bytes_per_step"][-1] / max(1, sp["exchange_bytes_per_step"][-1]), 1),
           max_abs_loss_difference=dict(sparse_vs_single=maxdiff(sp["loss_total_xe_reg_per_step"], ref["loss_total_xe_reg_per_step"]),
                                        sparse_vs_dense=maxdiff(sp["loss_total_xe_reg_per_step"], de["loss_total_xe_reg_per_step"])),
           item_table_checksum=dict(sparse=sp["item_table_checksum"], dense=de["item_table_checksum"], single=ref["item_table_checksum"]),
           dense_checksum=dict(sparse=sp["dense_checksum"], dense=de["dense_checksum"], single=ref["dense_checksum"]))
json.dump(out, open("%s/dp_sparse_config5_$T.json" % O, "w"), indent=1)
print(json.dumps(out)[:1500])
PY
for f in ref sparse allreduce; do tail -n 3 $O/dpc5_$f.log | cut -c1-300; done
