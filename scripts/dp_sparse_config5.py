#!/usr/bin/env python
"""CHAM_DP_MODE=sparse at a REDUCED BASELINE config 5 (default: 500 k articles x item-embedding width 378, 128-d ACE, 200 negatives,
seq_len 20, global batch 512) on several ranks - what the exchange moves per step, and that the sharded step computes the
single-process step.  Launch:
    CHAM_DIST_BACKEND=gloo CHAM_DP_MODE=sparse python -m torch.distributed.run --nproc-per-node 8 ... scripts/dp_sparse_config5.py --out X.json
    CHAM_DP_MODE=allreduce ... (dense exchange of the whole flat buffer, for the byte comparison)
    python scripts/dp_sparse_config5.py --out ref.json      (one process: the same global batches as micro-batches of --micro sessions)
Several ranks may share one GPU over gloo (functional run, not a scaling number: the ranks time-share the device)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-items", type=int, default=500_000)
    ap.add_argument("--emb", type=int, default=378)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--neg", type=int, default=200)
    ap.add_argument("--micro", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from chameleon_recsys_amd.nar import synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime
    from chameleon_recsys_amd.nar.parallel import DataParallelNAR
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    backend = os.environ.get("CHAM_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend)
    ifc = dict(recency=True, novelty=True, article_content_embeddings=True, item_clicked_embeddings=True, items_embedding_size=args.emb)
    p = synthetic.default_params(args.n_items, 128, seq_len=20, batch_size=args.batch, neg=args.neg, neg_from_buffer=3000, buffer_size=20000,
                                 for_norm=2000, C=1024, H=255, internal_features_config=ifc)
    rt = NARRuntime(p, device="cuda:%d" % local, seed=42)
    model = NARModuleModel(ModeKeys.TRAIN, None, None, p['session_features_config'], p['articles_features_config'], args.batch, p['lr'], 1.0,
                           args.neg, 3000, p['content_article_embeddings_matrix'], softmax_temperature=0.1, reg_weight_decay=1e-5,
                           recent_clicks_buffer_max_size=20000, recent_clicks_for_normalization=2000, articles_metadata=p['articles_metadata'],
                           CAR_embedding_size=1024, rnn_units=255, runtime=rt)
    dp = DataParallelNAR(model)
    batches = synthetic.make_batches(args.steps, args.batch, 20, args.n_items, p['session_features_config'], length_dist='full',
                                     sessions_per_hour=args.batch * 2, seed=7)
    state = DeviceClickedItemsState(1.0, 20000, 2000, args.n_items, device="cuda:%d" % local)
    losses, xbytes, rows, ms = [], [], [], []
    for f, l in batches:
        model.feed_state(state, state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if world > 1:
            d = dp.upload(f, l)
            model.train_step(d)
            loss = dp.global_loss().cpu().numpy()
        else:
            loss = model.train_step_microbatched(f, l, args.micro).cpu().numpy()
            d = model._d
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        state.update_from_device_batch(d['aci'], d['g_event_ts'])
        losses.append([float(x) for x in loss])
        xbytes.append(int(dp.last_exchange_bytes)); rows.append(int(getattr(dp, 'last_touched_rows', 0)))
    e = rt.layout.entries['items_embedding']
    w = rt.flat[e.offset:e.offset + e.size].double()
    out = dict(world=world, backend=backend if world > 1 else None, dp_mode=dp.mode if world > 1 else None, n_items=args.n_items, item_embedding=list(e.shape),
               global_batch=args.batch, sessions_per_rank=args.batch // world, negatives=args.neg, flat_parameter_bytes=int(rt.flat.numel() * 4),
               item_table_bytes=int(e.size * 4), exchange_bytes_per_step=xbytes, touched_item_rows_per_step=rows,
               loss_total_xe_reg_per_step=losses, ms_per_step=[round(x, 1) for x in ms],
               item_table_checksum=[float(w.sum()), float((w * w).sum())],
               dense_checksum=float(rt.flat[e.offset + e.size:].double().abs().sum()))
    if rank == 0:
        print(json.dumps(out))
        if args.out:
            json.dump(out, open(args.out, "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
