"""Manual tool: per-shape GEMM time inside the G1 full-length training step (HIP events around every cham_gemm_* launch, lanes
overlapped as in production).  usage: python scripts/gemm_breakdown.py"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from chameleon_recsys_amd.nar import synthetic
from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime
from chameleon_recsys_amd.nar.parallel import DataParallelNAR


def main():
    cfg = bench.G1
    Bg = cfg['batch']
    params = synthetic.default_params(cfg['n_items'], cfg['ace_dim'], seq_len=cfg['seq_len'], batch_size=Bg, neg=cfg['neg'],
                                      neg_from_buffer=cfg['neg_from_buffer'], buffer_size=cfg['buffer'],
                                      for_norm=cfg['for_norm'], C=cfg['C'], H=cfg['H'], seed=42)
    batches = synthetic.make_batches(4, Bg, cfg['seq_len'], cfg['n_items'], params['session_features_config'], seed=42,
                                     length_dist="full", sessions_per_hour=Bg * 2)
    rt = NARRuntime(params, device="cuda:0", seed=42)
    model = NARModuleModel(ModeKeys.TRAIN, None, None, params['session_features_config'], params['articles_features_config'],
                           Bg, params['lr'], 1.0, cfg['neg'], cfg['neg_from_buffer'], params['content_article_embeddings_matrix'],
                           softmax_temperature=params['softmax_temperature'], reg_weight_decay=params['reg_weight_decay'],
                           recent_clicks_buffer_max_size=cfg['buffer'], recent_clicks_for_normalization=cfg['for_norm'],
                           articles_metadata=params['articles_metadata'], CAR_embedding_size=cfg['C'], rnn_units=cfg['H'], runtime=rt)
    dp = DataParallelNAR(model)
    state = DeviceClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'], device="cuda:0")
    dev = [dp.upload(f, l) for f, l in batches]

    def step(i):
        k = i % len(dev)
        model.feed_state(state, state)
        model.train_step(dev[k])
        state.update_from_device_batch(dev[k]['aci'], dev[k]['g_event_ts'])
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(20):
        step(8 + i)
    torch.cuda.synchronize()
    print("ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
    rt.profile = []
    n = 6
    for i in range(n):
        step(28 + i)
    torch.cuda.synchronize()
    prof, rt.profile = rt.profile, None
    agg = collections.OrderedDict()
    for r in prof:
        key = (r['M'], r['N'], r['K'], r['transA'], r['transB'], r['act'], int(r['dref']), r['splits'])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += r['ev'][0].elapsed_time(r['ev'][1])
    tot = 0.0
    for key, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * key[0] * key[1] * key[2]
        tot += ms / n
        print("M=%-7d N=%-5d K=%-7d tA=%d tB=%d act=%d dref=%d splits=%d: %5.1f/step %8.3f ms/step  %6.1f TFLOP/s" % (
            key + (c / n, ms / n, fl * c / (ms * 1e-3) / 1e12 if ms > 0 else 0)))
    print("sum of GEMM launch times %.3f ms/step" % tot)


if __name__ == "__main__":
    main()
