"""Diagnostic (not a test): per-stage max errors HIP vs oracle on the tiny config (padded [B, T] layout: valid-position
compaction switched off).  python -m tests.debug_stages"""
import numpy as np
import torch

from chameleon_recsys_amd.nar import synthetic
from tests import helpers as H


def main(empty=False):
    p = H.tiny_params()
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, [] if empty else batches[:3])
    model, orc = H.make_pair(p)
    model.rt.compact = False          # the per-stage comparisons below index the plan buffers in the padded [B, T] layout
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


if __name__ == "__main__" and "--self" not in __import__("sys").argv and "--taps" not in __import__("sys").argv:
    import sys
    main(empty="--empty" in sys.argv)


def selfcheck(empty=False):
    """Backward kernels vs torch ops on the GPU, fed with the kernels' OWN forward tensors."""
    p = H.tiny_params()
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, [] if empty else batches[:3])
    model, orc = H.make_pair(p)
    model.rt.compact = False          # the per-stage comparisons below index the plan buffers in the padded [B, T] layout
    f, l = batches[0] if empty else batches[3]
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    model.feed_state(pop, buf)
    d = model.upload_batch(f, l)
    pl = model.forward(d)
    model.backward()
    torch.cuda.synchronize()
    rt, L = model.rt, model.rt.layout
    P, G = rt.p, rt.g
    B, T, N, BT, NC, Rc = pl.B, pl.T, pl.N, pl.BT, pl.NC, pl.Rc
    C = L.C
    lk = lambda y: torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    e = lambda a, b: "%.3e (scale %.3e)" % (float((a - b).abs().max()), float(b.abs().max()))
    mask = pl.mask.bool()
    tau = model.softmax_temperature
    probs = pl.probs.view(BT, NC)
    onehot = torch.zeros_like(probs); onehot[:, 0] = 1
    ds = ((probs - onehot) * mask[:, None].float() / (tau * d['sum_mask'])).reshape(-1)
    print("ds", e(pl.ds, ds))
    # oracle d xe / d logits
    ref = orc.forward(f, l, buf, pop, 'train')
    ref['logits'].retain_grad()
    ref['xe_loss'].backward()
    print("ds vs oracle dlogits", e(pl.ds.cpu().view(B, T, NC), ref['logits'].grad))
    dS3 = ds[:, None] * P('Ws4')[None, :] * lk(pl.S3)
    print("dS3", e(pl.dS3, dS3))
    print("g Ws4", e(G('Ws4'), (pl.S3 * ds[:, None]).sum(0)), " g bs3", e(G('bs3'), dS3.sum(0)))
    print("g Ws3", e(G('Ws3'), pl.S2.t() @ dS3))
    dS2 = (dS3 @ P('Ws3').t()) * lk(pl.S2)
    print("dS2", e(pl.dS2, dS2))
    print("g Ws2", e(G('Ws2'), pl.S1.t() @ dS2), " g bs2", e(G('bs2'), dS2.sum(0)))
    dS1 = (dS2 @ P('Ws2').t()) * lk(pl.S1)
    print("dS1", e(pl.dS1, dS1))
    Z2c = pl.Z2[BT:]
    predr = pl.pred.repeat_interleave(NC, 0)
    print("g Ws1", e(G('Ws1'), (Z2c * predr).t() @ dS1), " g bs1", e(G('bs1'), dS1.sum(0)))
    dM = dS1 @ P('Ws1').t()
    dZ2c = dM * predr * (1 - Z2c * Z2c)
    print("dZ2c", e(pl.dZ2[BT:], dZ2c))
    dpred = (dM * Z2c).view(BT, NC, C).sum(1) * (1 - pl.pred * pl.pred)
    print("dpred_pre", e(pl.dpred, dpred))
    print("g Wf2", e(G('Wf2'), pl.FC1.t() @ dpred), " g bf2", e(G('bf2'), dpred.sum(0)))
    dFC1 = (dpred @ P('Wf2').t()) * lk(pl.FC1)
    print("dFC1", e(pl.dFC1, dFC1))
    print("g Wf1", e(G('Wf1'), pl.rnn_out[-1].t() @ dFC1))
    drnn = dFC1 @ P('Wf1').t()
    # UGRNN BPTT in torch
    Hp = L.Hp
    Wh = P('rnn0/Wh')
    g_, c_, hp_ = pl.G[0].view(B, T, Hp), pl.Cc[0].view(B, T, Hp), pl.hprev[0].view(B, T, Hp)
    dout = drnn.view(B, T, Hp)
    carry = torch.zeros(B, Hp, device=drnn.device)
    dx = torch.zeros(B, T, 2 * Hp, device=drnn.device)
    sl = pl.seq_len.long()
    for t in range(T - 1, -1, -1):
        valid = (t < sl)[:, None].float()
        dh = (dout[:, t] + carry)
        dzg = dh * (hp_[:, t] - c_[:, t]) * g_[:, t] * (1 - g_[:, t]) * valid
        dzc = dh * (1 - g_[:, t]) * (1 - c_[:, t] ** 2) * valid
        dz = torch.cat([dzg, dzc], 1)
        dx[:, t] = dz
        newc = dh * g_[:, t] + dz @ Wh.t()
        carry = valid * newc + (1 - valid) * carry
    print("dxproj", e(pl.dxproj, dx.view(BT, 2 * Hp)))
    dxf = dx.view(BT, 2 * Hp)
    print("g rnn Wx", e(G('rnn0/Wx'), pl.Z2[:BT].t() @ dxf), " g rnn Wh", e(G('rnn0/Wh'), pl.hprev[0].t() @ dxf))
    dZ2in = (dxf @ P('rnn0/Wx').t()) * (1 - pl.Z2[:BT] ** 2)
    print("dZ2 in", e(pl.dZ2[:BT], dZ2in))
    dZ2 = torch.cat([dZ2in, dZ2c], 0)
    print("g W2", e(G('W2'), pl.Z1.t() @ dZ2), " g b2", e(G('b2'), dZ2.sum(0)))
    dZ1 = (dZ2 @ P('W2').t()) * lk(pl.Z1)
    print("dZ1", e(pl.dZ1, dZ1))
    dU = dZ1[:BT] + dZ1[BT:].view(BT, NC, C).sum(1)
    print("dU", e(pl.dU, dU))
    pmax = pl.pmax
    dV = torch.zeros_like(pl.dV)
    dV[:BT] = dZ1[:BT]
    dV[BT:2 * BT] = dZ1[BT:].view(BT, NC, C)[:, 0]
    slot = pl.neg_slot.view(BT, N).long()
    rows = dZ1[BT:].view(BT, NC, C)[:, 1:].reshape(-1, C)
    sl_flat = slot.reshape(-1)
    ok = sl_flat >= 0
    dV.index_add_(0, 2 * BT + sl_flat[ok], rows[ok])
    print("dV", e(pl.dV, dV))
    print("g W1c", e(G('W1c'), pl.Xc_s.t() @ dU), " g W1i", e(G('W1i'), pl.Xi_s.t() @ dV))
    print("dXc", e(pl.dXc, dU @ P('W1c').t()), " dXi", e(pl.dXi, dV @ P('W1i').t()))
    print("g gamma_item", e(G('gamma_item'), (pl.dXi * pl.Xi_raw).sum(0)), " g beta_ctx", e(G('beta_ctx'), pl.dXc.sum(0)))


if __name__ == "__main__" and "--self" in __import__("sys").argv:
    selfcheck(empty="--empty" in __import__("sys").argv)


def taps(empty=False):
    p = H.tiny_params()
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, [] if empty else batches[:3])
    model, orc = H.make_pair(p)
    model.rt.compact = False          # the per-stage comparisons below index the plan buffers in the padded [B, T] layout
    f, l = batches[0] if empty else batches[3]
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    model.feed_state(pop, buf)
    d = model.upload_batch(f, l)
    pl = model.forward(d)
    model.backward()
    torch.cuda.synchronize()
    orc.debug_taps = []
    ref = orc.forward(f, l, buf, pop, 'train')
    ref['xe_loss'].backward()
    B, T, N, BT, NC = pl.B, pl.T, pl.N, pl.BT, pl.NC
    (p1, p2, p3), (n1, n2, n3) = orc.debug_taps
    mask = ref['mask'].numpy()
    for name, gp, gn, hip, dhip in [("S3", p3, n3, pl.S3, pl.dS3), ("S2", p2, n2, pl.S2, pl.dS2), ("S1", p1, n1, pl.S1, pl.dS1)]:
        K = gp.shape[-1]
        act = torch.cat([gp.detach().unsqueeze(2), gn.detach()], 2).numpy()           # [B,T,NC,K]
        grad = torch.cat([gp.grad.unsqueeze(2), gn.grad], 2).numpy()
        h = hip.cpu().numpy().reshape(B, T, NC, K)
        dh = dhip.cpu().numpy().reshape(B, T, NC, K)
        # hip d* buffers hold gradient w.r.t. the PRE-activation; oracle .grad is w.r.t. the POST-activation
        lk = np.where(act > 0, 1.0, 0.2)
        dpre_ref = grad * lk
        err = np.abs(dh - dpre_ref)
        print(name, "act err(masked)", np.abs(h - act)[mask].max(), "dpre err", err.max(), "scale", np.abs(dpre_ref).max(),
              "sign flips", int(((h > 0) != (act > 0))[mask].sum()), "of", int(mask.sum()) * NC * K)
        idx = np.unravel_index(np.argmax(err), err.shape)
        print("   worst at", idx, "mask", mask[idx[0], idx[1]], "hip act", h[idx], "ref act", act[idx], "hip d", dh[idx], "ref d", dpre_ref[idx])


if __name__ == "__main__" and "--taps" in __import__("sys").argv:
    taps(empty="--empty" in __import__("sys").argv)
