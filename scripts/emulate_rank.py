"""What ONE rank of an N-GPU weak-scaling run computes, without the collectives (one GPU is enough): the global batch has N x 256
sessions, this process takes rank 0's 256 rows.  Shows how much of the step grows with the GLOBAL batch (candidate pool, state update,
replicated integers) - the part weak scaling cannot hide.  usage: python scripts/emulate_rank.py [--strong] [--graph] [N ...]
--graph: the step replayed from a hipGraph (nar_model.GraphedTrainStep) after the eager warm-up - what the host then submits per step is three calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench as B
from chameleon_recsys_amd.nar import synthetic
from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
from chameleon_recsys_amd.nar.nar_model import GraphedTrainStep, ModeKeys, NARModuleModel, NARRuntime
from chameleon_recsys_amd.nar.parallel import DataParallelNAR


def run(world, steps=20, warmup=6, seed=42, strong=False, graph=False):
    cfg = B.G1
    # weak: 256 rows per rank (global batch 256 x N); strong (BASELINE configs[2]: global batch 256): 256 / N rows per rank
    Bl, Bg = (cfg['batch'] // world, cfg['batch']) if strong else (cfg['batch'], cfg['batch'] * world)
    params = synthetic.default_params(cfg['n_items'], cfg['ace_dim'], seq_len=cfg['seq_len'], batch_size=Bg, neg=cfg['neg'],
                                      neg_from_buffer=cfg['neg_from_buffer'], buffer_size=cfg['buffer'], for_norm=cfg['for_norm'],
                                      C=cfg['C'], H=cfg['H'], seed=seed)
    batches = synthetic.make_batches(8, Bg, cfg['seq_len'], cfg['n_items'], params['session_features_config'], seed=seed,
                                     length_dist="full", sessions_per_hour=Bg * 2)
    rt = NARRuntime(params, device="cuda:0", seed=seed)
    model = NARModuleModel(ModeKeys.TRAIN, None, None, params['session_features_config'], params['articles_features_config'],
                           Bg, params['lr'], 1.0, cfg['neg'], cfg['neg_from_buffer'], params['content_article_embeddings_matrix'],
                           softmax_temperature=params['softmax_temperature'], reg_weight_decay=params['reg_weight_decay'],
                           recent_clicks_buffer_max_size=cfg['buffer'], recent_clicks_for_normalization=cfg['for_norm'],
                           articles_metadata=params['articles_metadata'], CAR_embedding_size=cfg['C'], rnn_units=cfg['H'], runtime=rt)
    dp = DataParallelNAR(model)
    dp.rank, dp.world = 0, world                    # slicing only: no process group, no collectives
    rt.dp_rank, rt.dp_world = 0, world
    state = DeviceClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'], device="cuda:0")
    dev = [dp.upload(f, l) for f, l in batches]
    for k in range(2):
        state.update_from_device_batch(dev[k]['aci'], dev[k]['g_event_ts'])

    def step(i):
        k = i % 8
        model.feed_state(state, state)
        model.train_step(dev[k])
        state.update_from_device_batch(dev[k]['aci'], dev[k]['g_event_ts'])
        model.presample(dev[(k + 1) % 8])
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if graph:
        eager_step, gs = step, GraphedTrainStep(model, state)

        def step(i):
            k = i % 8
            gs.step(dev[k], dev[(k + 1) % 8])
        for i in range(warmup, warmup + 3):          # capture + two replays outside the clock (the timed loop starts at the same batch)
            step(i)
        for i in range(warmup + 3, warmup + 8):
            step(i)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for i in range(3):
        step(warmup + steps + i)
    host = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    print("emulated rank 0 of %d, %s scaling, %s (global batch %d, local %d): %.3f ms/step (host enqueue %.3f ms) -> %.0f sessions/s per GPU, x%d = %.0f "
          "if the collectives hide" % (world, "strong" if strong else "weak", "hipGraph replay" if graph else "eager", Bg, Bl, dt * 1e3, host * 1e3, Bl / dt,
                                       world, world * Bl / dt), flush=True)
    return dt


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = {}
    for w in [int(x) for x in args] or [1, 2, 4, 8]:
        res[w] = run(w, strong="--strong" in sys.argv, graph="--graph" in sys.argv)
    if "--strong" in sys.argv and 1 in res:
        print("strong scaling emulated (compute only): " + ", ".join("N=%d %.2fx" % (w, res[1] / t) for w, t in sorted(res.items())))
