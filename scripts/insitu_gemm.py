"""Manual experiment: are the three CAR layer-2 GEMMs slower inside the training step than in tests/bench_gemm.py, and is
that the data (values / placement) or the surrounding kernels?  Times them (HIP events, no overlap) (1) inside the step,
(2) back to back on the step's own buffers, (3) back to back on fresh randn buffers."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CHAM_OVERLAP"] = "0"
import torch
import bench
from chameleon_recsys_amd._lib import ptr
from chameleon_recsys_amd.nar import synthetic
from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime, ACT_TANH, ACT_LEAKY
from chameleon_recsys_amd.nar.parallel import DataParallelNAR


def main():
    cfg = bench.G1
    Bg = cfg['batch']
    params = synthetic.default_params(cfg['n_items'], cfg['ace_dim'], seq_len=cfg['seq_len'], batch_size=Bg, neg=cfg['neg'],
                                      neg_from_buffer=cfg['neg_from_buffer'], buffer_size=cfg['buffer'],
                                      for_norm=cfg['for_norm'], C=cfg['C'], H=cfg['H'], seed=42)
    batches = synthetic.make_batches(4, Bg, cfg['seq_len'], cfg['n_items'], params['session_features_config'], seed=42,
                                     length_dist="g1", sessions_per_hour=Bg * 2)
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
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    rt.profile = []
    for i in range(4):
        step(6 + i)
    torch.cuda.synchronize()
    prof, rt.profile = rt.profile, None
    big = [r for r in prof if 2.0 * r['M'] * r['N'] * r['K'] > 1e11]
    for r in big[:3] + big[-3:]:
        ms = r['ev'][0].elapsed_time(r['ev'][1])
        print("in-step  M=%d N=%d K=%d tA=%d tB=%d act=%d dref=%d: %.3f ms %.1f TFLOP/s" % (
            r['M'], r['N'], r['K'], r['transA'], r['transB'], r['act'], r['dref'], ms, 2.0 * r['M'] * r['N'] * r['K'] / ms / 1e9))
    pl = model._plan
    p, g = rt.p, rt.g
    C, BT, Rc, Rall = rt.layout.C, pl.BT, pl.Rc, pl.Rall

    def timeit(name, fn, flops):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 6
        print("%-34s %.3f ms %.1f TFLOP/s" % (name, ms, flops / ms / 1e9), flush=True)

    def trio(tag, Z1, Z2, dZ2, dZ1, W2, b2, gW2):
        timeit(tag + " fwd  NN tanh", lambda: rt.gemm(Z1[BT:], W2, Z2[BT:], Rc, C, C, C, C, C, bias=b2, act=ACT_TANH), 2.0 * Rc * C * C)
        timeit(tag + " dgrad NT leaky'", lambda: rt.gemm(dZ2[BT:], W2, dZ1[BT:], Rc, C, C, C, C, C, transB=1, dref=Z1[BT:], ldr=C, dact=ACT_LEAKY), 2.0 * Rc * C * C)
        timeit(tag + " wgrad TN", lambda: rt.gemm(Z1, dZ2, gW2, C, C, Rall, C, C, C, transA=1, splits=0), 2.0 * Rall * C * C)
    print("Z1 |mean| %.3g zero-frac %.3f ; dZ2 |mean| %.3g zero-frac %.3f" % (
        pl.Z1.abs().mean().item(), (pl.Z1 == 0).float().mean().item(), pl.dZ2.abs().mean().item(), (pl.dZ2 == 0).float().mean().item()))
    print("ptr %% 4096: Z1 %d Z2 %d dZ2 %d dZ1 %d W2 %d" % tuple(t.data_ptr() % 4096 for t in (pl.Z1, pl.Z2, pl.dZ2, pl.dZ1, p('W2'))))
    gsave = g('W2').clone()
    trio("step buffers", pl.Z1, pl.Z2, pl.dZ2, pl.dZ1, p('W2'), p('b2'), g('W2'))
    # same buffers, random contents
    for t in (pl.Z1, pl.dZ2):
        t.normal_()
    trio("step buffers, randn data", pl.Z1, pl.Z2, pl.dZ2, pl.dZ1, p('W2'), p('b2'), g('W2'))
    fresh = [torch.randn(Rall, C, device="cuda") for _ in range(4)]
    W = torch.randn(C, C, device="cuda") * 0.03; b = torch.randn(C, device="cuda"); gw = torch.empty(C, C, device="cuda")
    trio("fresh buffers", fresh[0], fresh[1], fresh[2], fresh[3], W, b, gw)
    trio("fresh data, param-buffer W2", fresh[0], fresh[1], fresh[2], fresh[3], p('W2'), p('b2'), g('W2'))


if __name__ == "__main__":
    main()
