#!/usr/bin/env python
"""Debug: G1-size ragged batches, presampled vs in-step negative sampling (and run-to-run repeatability of each), per-step comparison
of the negatives, the pool and the loss.  python scripts/debug_presample.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from chameleon_recsys_amd.nar import synthetic
from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sync_each = os.environ.get("SYNC_EACH", "1") == "1"
cfg = B.G1
p = synthetic.default_params(cfg['n_items'], cfg['ace_dim'], seq_len=cfg['seq_len'], batch_size=256, neg=cfg['neg'], neg_from_buffer=cfg['neg_from_buffer'],
                             buffer_size=cfg['buffer'], for_norm=cfg['for_norm'], C=cfg['C'], H=cfg['H'], seed=42)
batches = synthetic.make_batches(8, 256, cfg['seq_len'], cfg['n_items'], p['session_features_config'], seed=43, length_dist="g1", sessions_per_hour=512)


def run(use_presample):
    rt = NARRuntime(p, device="cuda:0", seed=42)
    model = NARModuleModel(ModeKeys.TRAIN, None, None, p['session_features_config'], p['articles_features_config'], 256, p['lr'], 1.0, cfg['neg'],
                           cfg['neg_from_buffer'], p['content_article_embeddings_matrix'], softmax_temperature=p['softmax_temperature'],
                           reg_weight_decay=p['reg_weight_decay'], recent_clicks_buffer_max_size=cfg['buffer'],
                           recent_clicks_for_normalization=cfg['for_norm'], articles_metadata=p['articles_metadata'], CAR_embedding_size=cfg['C'],
                           rnn_units=cfg['H'], runtime=rt)
    state = DeviceClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'])
    if os.environ.get("NO_CONSUMED") == "1":       # state update fully ordered behind the step (wait_stream) instead of behind softmax_bwd
        state.note_consumed = lambda aci: None
    dev = [model.upload_batch(f, l) for f, l in batches]
    out = []
    for i in range(steps):
        k = i % 8
        model.feed_state(state, state)
        loss = model.train_step(dev[k])
        pl = model._plan
        rec = (pl.cur_neg_ids.clone(), pl.pool.clone() if hasattr(pl, 'pool') else None, loss.clone(), state.buf_ids.clone(), state.pop_norm.clone())
        state.update_from_device_batch(dev[k]['aci'], dev[k]['g_event_ts'])
        if use_presample:
            if os.environ.get("SYNC_BEFORE_PRE") == "1":
                torch.cuda.synchronize()
            model.presample(dev[(k + 1) % 8])
            if os.environ.get("SYNC_AFTER_PRE") == "1":
                torch.cuda.synchronize()
        if sync_each:
            torch.cuda.synchronize()
        out.append(rec)
    torch.cuda.synchronize()
    return [(a.cpu().numpy(), None if b is None else b.cpu().numpy(), c.cpu().numpy(), d.cpu().numpy(), e.cpu().numpy()) for a, b, c, d, e in out]


runs = {"pre1": run(True), "pre2": run(True), "pre3": run(True), "in1": run(False)}
for a, b in (("pre1", "pre2"), ("pre1", "pre3"), ("pre1", "in1")):
    first = None
    for i, (x, y) in enumerate(zip(runs[a], runs[b])):
        same = [np.array_equal(x[j], y[j]) for j in (0, 1, 2) if x[j] is not None]
        if not all(same):
            first = (i, same, float(x[2][0]), float(y[2][0]))
            break
    print("%s vs %s: %s" % (a, b, "identical over %d steps" % steps if first is None else "first difference at step %d [neg_ids, pool, loss same?] %s loss %.6f vs %.6f" % first), flush=True)
