"""A second, independent restatement of the oracle's riskiest pieces - the only pin available without TensorFlow.

`oracle/nar_oracle.py` is float32 PyTorch with autograd; everything here is float64 numpy with HAND-DERIVED closed-form gradients,
written from the TF 1.12 definitions, not from the oracle's code:
  * UGRNNCell / GRUCell time steps under dynamic_rnn's length masking (nar_model.py:1308-1361): outputs + gradients w.r.t. the
    inputs, kernels and biases by explicit back-propagation through time;
  * -sum(log softmax(logits / tau)[..., 0] * mask) / sum(mask) (nar_model.py:511-515, 660-664) and its gradient
    (p - onehot_0) * mask / (tau * sum(mask));
  * tf.train.AdamOptimizer (nar_model.py:708-722): epsilon OUTSIDE the bias correction, correction folded into the step size.
The oracle is asserted against these, so it is not only checked against its own stored output (tests/test_golden.py)."""
import math

import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import synthetic
from oracle.nar_oracle import NAROracle
from tests import helpers as H


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def ugrnn_bptt(x, lengths, K, b, R):
    """tf.contrib.rnn.UGRNNCell (TF 1.12 rnn_cell.py): [g_act, c_act] = [x, h] K + b; c = tanh(c_act); g = sigmoid(g_act + 1);
    h' = g h + (1 - g) c.  dynamic_rnn: beyond a row's length the output is zero and the state is carried.  loss = sum(out * R).
    Returns out and d loss / d (x, K, b)."""
    B, T, I = x.shape
    H_ = K.shape[1] // 2
    h = np.zeros((B, H_)); hs, gs, cs, outs = [], [], [], []
    for t in range(T):
        z = np.concatenate([x[:, t], h], 1) @ K + b
        c, g = np.tanh(z[:, H_:]), _sig(z[:, :H_] + 1.0)
        hn = g * h + (1 - g) * c
        v = (t < lengths)[:, None]
        hs.append(h); gs.append(g); cs.append(c)
        outs.append(np.where(v, hn, 0.0)); h = np.where(v, hn, h)
    out = np.stack(outs, 1)
    dx, dK, db, dh = np.zeros_like(x), np.zeros_like(K), np.zeros_like(b), np.zeros((B, H_))
    for t in range(T - 1, -1, -1):
        v = (t < lengths)[:, None]
        dhn = np.where(v, R[:, t] + dh, 0.0)             # output path + state path (both only where the step is valid)
        carry = np.where(v, 0.0, dh)                      # invalid step: state carried through unchanged
        g, c, hp = gs[t], cs[t], hs[t]
        dg, dc = dhn * (hp - c), dhn * (1 - g)
        dz = np.concatenate([dg * g * (1 - g), dc * (1 - c * c)], 1)
        xin = np.concatenate([x[:, t], hp], 1)
        dK += xin.T @ dz; db += dz.sum(0)
        dxin = dz @ K.T
        dx[:, t] = dxin[:, :I]
        dh = carry + dhn * g + dxin[:, I:]
    return out, dx, dK, db


def gru_bptt(x, lengths, Kg, bg, Kc, bc, R):
    """tf.nn.rnn_cell.GRUCell (TF 1.12): [r, u] = sigmoid([x, h] Kg + bg); c = tanh([x, r h] Kc + bc); h' = u h + (1 - u) c."""
    B, T, I = x.shape
    H_ = Kc.shape[1]
    h = np.zeros((B, H_)); st, outs = [], []
    for t in range(T):
        ru = _sig(np.concatenate([x[:, t], h], 1) @ Kg + bg)
        r, u = ru[:, :H_], ru[:, H_:]
        c = np.tanh(np.concatenate([x[:, t], r * h], 1) @ Kc + bc)
        hn = u * h + (1 - u) * c
        v = (t < lengths)[:, None]
        st.append((h, r, u, c))
        outs.append(np.where(v, hn, 0.0)); h = np.where(v, hn, h)
    out = np.stack(outs, 1)
    dx = np.zeros_like(x); dKg, dbg, dKc, dbc = np.zeros_like(Kg), np.zeros_like(bg), np.zeros_like(Kc), np.zeros_like(bc)
    dh = np.zeros((B, H_))
    for t in range(T - 1, -1, -1):
        v = (t < lengths)[:, None]
        hp, r, u, c = st[t]
        dhn = np.where(v, R[:, t] + dh, 0.0)
        carry = np.where(v, 0.0, dh)
        du, dc = dhn * (hp - c), dhn * (1 - u)
        dzc = dc * (1 - c * c)
        xc = np.concatenate([x[:, t], r * hp], 1)
        dKc += xc.T @ dzc; dbc += dzc.sum(0)
        dxc = dzc @ Kc.T
        drh = dxc[:, I:]
        dr = drh * hp
        dzg = np.concatenate([dr * r * (1 - r), du * u * (1 - u)], 1)
        xg = np.concatenate([x[:, t], hp], 1)
        dKg += xg.T @ dzg; dbg += dzg.sum(0)
        dxg = dzg @ Kg.T
        dx[:, t] = dxc[:, :I] + dxg[:, :I]
        dh = carry + dhn * u + drh * r + dxg[:, I:]
    return out, dx, (dKg, dbg, dKc, dbc)


def _oracle(cell, layers=1, Hn=24, C=16):
    p = H.tiny_params(C=C, H=Hn, neg=5, batch_size=6, rnn_cell=cell, rnn_num_layers=layers, n_items=200, ace_dim=8)
    return NAROracle(p, seed=4), p


@pytest.mark.parametrize("cell", ["ugrnn", "gru"])
def test_recurrent_cell_forward_and_bptt(cell):
    orc, p = _oracle(cell)
    rng = np.random.default_rng(1)
    B, T, C, Hn = 6, 7, 16, 24
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    lengths = np.array([7, 3, 1, 5, 7, 2])
    R = rng.standard_normal((B, T, Hn)).astype(np.float32)
    for k, v in orc.w.items():          # non-trivial biases
        if k.startswith('rnn/') and k.endswith('bias'):
            v.data = torch.from_numpy((0.3 * rng.standard_normal(v.shape)).astype(np.float32) + (1.0 if 'gates' in k else 0.0))
    xt = torch.from_numpy(x).requires_grad_(True)
    out = orc._rnn(xt, torch.from_numpy(lengths))
    (out * torch.from_numpy(R)).sum().backward()
    w = {k: v.detach().numpy().astype(np.float64) for k, v in orc.w.items()}
    if cell == 'ugrnn':
        ref_out, dx, dK, db = ugrnn_bptt(x.astype(np.float64), lengths, w['rnn/0/kernel'], w['rnn/0/bias'], R.astype(np.float64))
        grads = {'rnn/0/kernel': dK, 'rnn/0/bias': db}
    else:
        ref_out, dx, (dKg, dbg, dKc, dbc) = gru_bptt(x.astype(np.float64), lengths, w['rnn/0/gates/kernel'], w['rnn/0/gates/bias'],
                                                      w['rnn/0/candidate/kernel'], w['rnn/0/candidate/bias'], R.astype(np.float64))
        grads = {'rnn/0/gates/kernel': dKg, 'rnn/0/gates/bias': dbg, 'rnn/0/candidate/kernel': dKc, 'rnn/0/candidate/bias': dbc}
    assert np.abs(out.detach().numpy() - ref_out).max() < 2e-6
    assert not out.detach().numpy()[1, 3:].any() and not out.detach().numpy()[2, 1:].any()          # zero output past the length
    assert np.abs(xt.grad.numpy() - dx).max() < 2e-5 * max(1.0, np.abs(dx).max())
    for k, gref in grads.items():
        assert np.abs(orc.w[k].grad.numpy() - gref).max() < 2e-5 * max(1.0, np.abs(gref).max()), k


def test_sampled_softmax_loss_and_gradient_closed_form():
    p = H.tiny_params(C=16, H=24, neg=5, batch_size=12, n_items=200, ace_dim=8, softmax_temperature=0.1)
    batches = synthetic.make_batches(3, 12, 8, 200, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:2])
    orc = NAROracle(p, seed=2)
    f, l = batches[2]
    ref = orc.forward(f, l, st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm(), 'train')
    logits = ref['logits']
    logits.retain_grad()
    ref['xe_loss'].backward()
    z = logits.detach().numpy().astype(np.float64) / 0.1
    mask = ref['mask'].numpy().astype(np.float64)
    zs = z - z.max(-1, keepdims=True)
    prob = np.exp(zs) / np.exp(zs).sum(-1, keepdims=True)
    xe = -(np.log(prob[..., 0]) * mask).sum() / mask.sum()
    assert abs(float(ref['xe_loss'].detach()) - xe) < 1e-5
    assert np.abs(ref['probs'].detach().numpy() - prob).max() < 1e-5
    onehot = np.zeros_like(prob); onehot[..., 0] = 1.0
    dlogits = (prob - onehot) * mask[..., None] / (0.1 * mask.sum())
    assert np.abs(logits.grad.numpy() - dlogits).max() < 1e-5 * max(1.0, np.abs(dlogits).max())
    # the regulariser: sum over kernels / embeddings / gamma / beta of lambda * |w|^2 / 2, no biases, no recurrent weights
    lam = p['reg_weight_decay']
    reg = sum(lam * 0.5 * float((v.detach().numpy().astype(np.float64) ** 2).sum()) for k, v in orc.w.items()
              if not k.endswith('bias') and not k.startswith('rnn/'))
    assert abs(float(ref['reg_loss'].detach()) - reg) < 1e-6 * max(1.0, reg)


def test_tf_adam_two_steps_closed_form():
    orc, p = _oracle('ugrnn')
    rng = np.random.default_rng(3)
    lr, b1, b2, eps = p['lr'], 0.9, 0.999, 1e-8
    w0 = {k: v.detach().numpy().astype(np.float64).copy() for k, v in orc.w.items()}
    g1 = {k: rng.standard_normal(v.shape) * 1e-2 for k, v in w0.items()}
    g2 = {k: rng.standard_normal(v.shape) * 1e-2 for k, v in w0.items()}
    for g in (g1, g2):
        orc.adam_update({k: torch.from_numpy(v.astype(np.float32)) for k, v in g.items()})
    for k in w0:
        ga, gb = g1[k].astype(np.float32).astype(np.float64), g2[k].astype(np.float32).astype(np.float64)
        m1, v1 = (1 - b1) * ga, (1 - b2) * ga * ga
        lr1 = lr * math.sqrt(1 - b2) / (1 - b1)
        w1 = w0[k] - lr1 * m1 / (np.sqrt(v1) + eps)               # eps is NOT bias-corrected (tf.train.AdamOptimizer)
        m2, v2 = b1 * m1 + (1 - b1) * gb, b2 * v1 + (1 - b2) * gb * gb
        lr2 = lr * math.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
        w2 = w1 - lr2 * m2 / (np.sqrt(v2) + eps)
        assert np.abs(orc.w[k].detach().numpy() - w2).max() < 2e-6 * max(1.0, np.abs(w2).max()) + 3e-7 * lr / 1e-4, k
        assert np.abs(orc.m[k].numpy() - m2).max() < 1e-6 * max(1e-3, np.abs(m2).max())
        # torch.optim.Adam would give a different step: eps inside the corrected denominator (the two differ where |g| ~ eps)
    assert orc.global_step == 2


def test_leaky_site_branch_override_is_leaky_relu_with_pinned_branches():
    """NAROracle._leaky_site (the hook the GPU gradient-parity tests use to run the oracle on the HIP path's branches): without an
    override it is tf.nn.leaky_relu; with one, the value changes only where the pinned branch disagrees with the sign (and then by
    0.8 |x|, i.e. nothing for the |x| ~ 1e-7 elements it is used on) and the gradient factor is the pinned one; elements outside
    `valid` keep their own branch."""
    orc, p = _oracle('ugrnn')
    x = torch.tensor([[-2.0, -1e-7, 1e-7, 3.0], [0.5, -0.5, 2e-7, -2e-7]], requires_grad=True)
    y = orc._leaky_site('Z1', x)
    assert torch.equal(y, torch.nn.functional.leaky_relu(x, 0.2))
    sign = torch.tensor([[False, True, False, True], [False, False, False, True]])          # flips at [0,1] [0,2] [1,0] [1,3]; agrees elsewhere
    valid = torch.tensor([[True], [False]])                                                   # second row: override ignored
    orc.leaky_signs = {'Z1': [(sign, valid)]}
    x2 = x.detach().clone().requires_grad_(True)
    y2 = orc._leaky_site('Z1', x2)
    y2.sum().backward()
    assert orc.leaky_signs['Z1'] == []                                                        # consumed in call order
    assert torch.equal(x2.grad[0], torch.tensor([0.2, 1.0, 0.2, 1.0]))                        # pinned branches on the valid row
    assert torch.equal(x2.grad[1], torch.tensor([1.0, 0.2, 1.0, 0.2]))                        # own branches on the other
    assert float((y2 - y.detach()).abs().max()) < 1e-6 and torch.equal(y2[1], y.detach()[1])
    orc.leaky_signs = None
    assert torch.equal(orc._leaky_site('Z1', x.detach()), torch.nn.functional.leaky_relu(x.detach(), 0.2))


def test_grad_significance_is_a_running_and_over_steps():
    g1 = {'a': torch.tensor([1.0, 1e-6, 0.5]), 'b': torch.tensor([0.0, 0.0])}
    g2 = {'a': torch.tensor([1e-6, 1.0, 0.5]), 'b': torch.tensor([0.0, 0.0])}
    s1 = H.grad_significance(g1)
    s2 = H.grad_significance(g2, acc=s1)
    assert s1['a'].tolist() == [True, False, True] and s2['a'].tolist() == [False, False, True]
    assert not s2['b'].any()
