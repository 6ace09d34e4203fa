"""Diagnostic (not a test): per-stage max errors HIP vs oracle on the tiny config.  python -m tests.debug_stages"""
import numpy as np
import torch

from chameleon_recsys_amd.nar import synthetic
from tests import helpers as H


def main(empty=False):
    p = H.tiny_params()
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, [] if empty else batches[:3])
    model, orc = H.make_pair(p)
    f, l = batches[0] if empty else batches[3]
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    model.feed_state(pop, buf)
    d = model.upload_batch(f, l)
    pl = model.forward(d)
    torch.cuda.synchronize()
    ref = orc.forward(f, l, buf, pop, 'train')
    L = model.rt.layout
    B, T, N, BT, NC = pl.B, pl.T, pl.N, pl.BT, pl.NC
    mask = ref['mask'].numpy()
    e = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max())
    print("neg equal:", np.array_equal(pl.neg_ids.cpu().numpy(), ref['neg_items'].numpy()), "P", pl.meta.cpu().numpy())
    fc, fi = L.f_ctx, L.f_item
    Xc = pl.Xc_s.cpu().numpy()[:, :fc].reshape(B, T, fc)
    Xi = pl.Xi_s.cpu().numpy()
    x_in = ref['x_in'].detach().numpy(); x_pos = ref['x_pos'].detach().numpy()
    print("ctx feats err", e(Xc, x_in[..., :fc]))
    print("item feats (clicked) err", e(Xi[:BT, :fi].reshape(B, T, fi), x_in[..., fc:]))
    err_cols = np.abs(Xi[:BT, :fi].reshape(B, T, fi) - x_in[..., fc:]).max((0, 1))
    print("   worst cols", np.argsort(-err_cols)[:5], np.sort(-err_cols)[:5])
    print("item feats (positive) err", e(Xi[BT:2 * BT, :fi].reshape(B, T, fi), x_pos[..., fc:]))
    Z2 = pl.Z2.cpu().numpy()
    C = L.C
    print("car_in err", e(Z2[:BT].reshape(B, T, C), ref['car_in'].detach().numpy()))
    Z2c = Z2[BT:].reshape(B, T, NC, C)
    print("car_pos err", e(Z2c[:, :, 0], ref['car_pos'].detach().numpy()))
    print("car_neg err (masked)", e(Z2c[:, :, 1:][mask], ref['car_neg'].detach().numpy()[mask]))
    H_ = L.H
    print("rnn_out err", e(pl.rnn_out[-1].cpu().numpy().reshape(B, T, -1)[..., :H_], ref['rnn_out'].detach().numpy()))
    print("pred err", e(pl.pred.cpu().numpy().reshape(B, T, C), ref['pred'].detach().numpy()))
    print("logits err (masked)", e(pl.logits.cpu().numpy().reshape(B, T, NC)[mask], ref['logits'].detach().numpy()[mask]))
    print("loss hip", pl.loss.cpu().numpy(), "ref", float(ref['total_loss']), float(ref['xe_loss']), float(ref['reg_loss']))
    for v in orc.w.values():
        v.grad = None
    ref['xe_loss'].backward()
    model.backward()
    torch.cuda.synchronize()
    g = model.rt.logical_grads()
    for k, v in orc.w.items():
        rg = v.grad.numpy() if v.grad is not None else np.zeros_like(v.detach().numpy())
        print("grad %-28s err %.3e  scale %.3e" % (k, e(g[k], rg), float(np.abs(rg).max())))


if __name__ == "__main__":
    import sys
    main(empty="--empty" in sys.argv)
