#!/usr/bin/env python
"""BASELINE config 5 - large-catalog stress: 5 M articles, 128-d ACE, seq_len 20, batch 4096, 200 negatives (1 GPU).

The optimizer step runs as session micro-batches (NARModuleModel.train_step_microbatched) so the B*T*(1+N) candidate-row
activations stay bounded; the embedding tables (item table 5M x 378 fp32 = 7.56 GB) and TF-style dense Adam
(28 B/param/step) make gather / scatter / optimizer the HBM-bound part.  Prints one JSON line with the step time and the
HBM-roofline fractions of the optimizer and gather kernels (HIP-event timed).

  python scripts/stress_large_catalog.py [--n-items 5000000] [--batch 4096] [--neg 200] [--micro 2048] [--steps 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-items", type=int, default=5_000_000)
    ap.add_argument("--ace-dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--neg", type=int, default=200)
    ap.add_argument("--micro", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import torch
    from chameleon_recsys_amd._lib import check, ptr
    from chameleon_recsys_amd.nar import synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.layout import ParamLayout
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime

    t0 = time.time()
    p = synthetic.default_params(args.n_items, args.ace_dim, seq_len=20, batch_size=args.batch, neg=args.neg, neg_from_buffer=3000,
                                 buffer_size=20000, for_norm=2000, C=1024, H=255)
    L = ParamLayout(p['session_features_config'], p['articles_features_config'], args.n_items, args.ace_dim, 1024, 255)
    w = L.init_logical(42, max_random_elems=50_000_000)
    rt = NARRuntime(p, weights=w)
    del w
    model = NARModuleModel(ModeKeys.TRAIN, None, None, p['session_features_config'], p['articles_features_config'], args.batch,
                           p['lr'], 1.0, args.neg, 3000, p['content_article_embeddings_matrix'], softmax_temperature=0.1,
                           reg_weight_decay=1e-5, recent_clicks_buffer_max_size=20000, recent_clicks_for_normalization=2000,
                           articles_metadata=p['articles_metadata'], CAR_embedding_size=1024, rnn_units=255, runtime=rt)
    batches = synthetic.make_batches(2, args.batch, 20, args.n_items, p['session_features_config'], length_dist='full',
                                     sessions_per_hour=args.batch * 2)
    state = DeviceClickedItemsState(1.0, 20000, 2000, args.n_items)
    setup_s = time.time() - t0
    ev = lambda: torch.cuda.Event(enable_timing=True)
    times, adam_ms = [], []
    for i in range(args.steps + 1):
        f, l = batches[i % 2]
        model.feed_state(state, state)
        torch.cuda.synchronize()
        t1 = time.time()
        loss = model.train_step_microbatched(f, l, args.micro)
        torch.cuda.synchronize()
        dt = time.time() - t1
        d = model._d
        state.update_from_device_batch(d['aci'], d['g_event_ts'])
        # Adam alone (re-applied on a scratch copy would change the weights: time a second, identical launch with lr = 0)
        e0, e1 = ev(), ev()
        e0.record()
        check(rt.lib.cham_adam_tf(ptr(rt.flat), ptr(rt.grads), ptr(rt.m), ptr(rt.v), L.total, L.n_reg, 0.0, 0.0, 1.0, 1.0, 1e-8,
                                  torch.cuda.current_stream().cuda_stream), "cham_adam_tf")      # beta=1, lr=0: a no-op pass
        e1.record(); torch.cuda.synchronize()
        if i > 0:
            times.append(dt); adam_ms.append(e0.elapsed_time(e1))
    step_s = float(np.mean(times))
    adam_bytes = 28.0 * L.total
    # ---- gather / scatter of the last micro-batch's item rows, stand-alone (HIP events): HBM-bound stages of the step
    pl, lib = model._plan, rt.lib
    RV, Fi, BT = 2 * pl.P + pl.pmax + 1, L.Fi, pl.P
    s_ = torch.cuda.current_stream().cuda_stream
    src_row_bytes = 4.0 * (L.D + sum(int(sg[2]) for sg in rt.item_segs.cpu().numpy() if sg[0] != 0))     # ACE + embedding rows read per item row
    gather_bytes = RV * (src_row_bytes + 2 * 4.0 * Fi)                                                    # + raw and scaled rows written

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    pg, pb = rt.p('gamma_item'), rt.p('beta_item')
    ms_lds = timed(lambda: check(lib.cham_item_assemble_lds(ptr(pl.ids_all), RV, BT, 2 * BT, ptr(rt.meta_cat), rt.n_items, ptr(rt.ace), L.D,
                                                            ptr(pl.rec_raw), ptr(pl.nov_raw), ptr(pl.stats), ptr(rt.item_desc), Fi, ptr(rt.item_segs),
                                                            rt.n_item_segs, ptr(rt.item_singles), rt.n_item_singles, ptr(rt.flat), ptr(pg), ptr(pb),
                                                            ptr(pl.Xi_raw), ptr(pl.Xi_s), s_), "lds"))
    ms_elem = timed(lambda: check(lib.cham_item_assemble(ptr(pl.ids_all), RV, BT, 2 * BT, ptr(rt.meta_cat), rt.n_items, ptr(rt.ace), L.D,
                                                         ptr(pl.rec_raw), ptr(pl.nov_raw), ptr(pl.stats), ptr(rt.item_desc), Fi, ptr(rt.flat),
                                                         ptr(pg), ptr(pb), ptr(pl.Xi_raw), ptr(pl.Xi_s), s_), "elementwise"))
    grp = [g for g in rt.item_emb_groups if g[0] == 5][0]          # (kind, feat, c0, dim, rows, offset) of the item-embedding table
    ms_group = timed(lambda: check(lib.cham_group_rows(ptr(pl.ids_all), RV, rt.item_id_bits, ptr(pl.perm), ptr(pl.seg), ptr(pl.group_ws), pl.group_ws.numel() * 4, s_), "group"))
    ms_scatter = timed(lambda: check(lib.cham_emb_grad_grouped(ptr(pl.dXi), RV, Fi, grp[2], grp[3], ptr(pg), ptr(pl.ids_all), ptr(pl.perm), ptr(pl.seg),
                                                               rt.grads.data_ptr() + 4 * grp[5], s_), "grouped"))
    scatter_bytes = RV * 4.0 * grp[3] * 2                           # read the rows' embedding-gradient columns, write one table row per distinct id (<= RV)
    roof = lambda by, ms: dict(bound="hbm", algorithmic_bytes=by, ms=round(ms, 4), achieved_gbs=round(by / (ms * 1e-3) / 1e9, 1),
                               peak_gbs=HBM_PEAK_GBS, frac=round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    out = dict(workload="large-catalog stress (BASELINE.json configs[4])", n_items=args.n_items, ace_dim=args.ace_dim,
               batch=args.batch, negatives=args.neg, micro_batch_sessions=args.micro, params=int(L.total),
               item_embedding_dim=L.entries['items_embedding'].shape[1], setup_s=round(setup_s, 1),
               step_s=round(step_s, 4), sessions_per_s=round(args.batch / step_s, 1), loss=[float(x) for x in loss.cpu().numpy()],
               hbm_gb_allocated=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
               adam=dict(bound="hbm", algorithmic_bytes=adam_bytes, ms=round(float(np.mean(adam_ms)), 3),
                         achieved_gbs=round(adam_bytes / (np.mean(adam_ms) * 1e-3) / 1e9, 1), peak_gbs=HBM_PEAK_GBS,
                         frac=round(adam_bytes / (np.mean(adam_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)),
               item_rows_per_micro_batch=int(RV),
               gather_lds_tiles=dict(kernel="k_item_assemble_lds (ACE / embedding rows -> LDS tile -> 16-byte row stores, raw + scaled)", **roof(gather_bytes, ms_lds)),
               gather_one_thread_per_element=dict(kernel="k_item_assemble (round-1 form)", **roof(gather_bytes, ms_elem)),
               embedding_gradient=dict(kernel="k_rs_hist / k_rs_scan / k_rs_scatter x passes + k_seg_table (%.3f ms: stable radix sort of the rows by id and the segment table, forward pass) + "
                                              "k_emb_grad_all (ONE launch: one wave per short segment, one workgroup per 64-row chunk of a long one, the last-arriving chunk adds the chunk sums in chunk order; deterministic, no float atomics)" % ms_group, **roof(scatter_bytes, ms_scatter)))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
