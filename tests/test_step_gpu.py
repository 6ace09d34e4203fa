"""NAR training step: HIP path vs the CPU oracle on identical inputs / weights / state.

Tolerances (BASELINE.json north_star): negative-sample indices bit-exact; logits / loss within 1e-3 (fp32)."""
import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


def _compare_step(model, orc, f, l, st, check_grads=True):
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    model.feed_state(pop, buf)
    d = model.upload_batch(f, l)
    model.forward(d)
    out = model.outputs_numpy()
    for v in orc.w.values():
        v.grad = None
    orc.debug_taps = {}
    ref = orc.forward(f, l, buf, pop, 'train')
    assert np.array_equal(out['neg_items'], ref['neg_items'].numpy()), "negative samples must be bit exact"
    mask = ref['mask'].numpy()
    assert np.abs(out['logits'] - ref['logits'].detach().numpy())[mask].max() < LOGIT_TOL
    assert np.abs(out['probs'] - ref['probs'].detach().numpy())[mask].max() < LOGIT_TOL
    assert abs(out['loss'][1] - float(ref['xe_loss'])) < LOGIT_TOL
    assert abs(out['loss'][2] - float(ref['reg_loss'])) < 1e-5
    assert abs(out['loss'][0] - float(ref['total_loss'])) < LOGIT_TOL
    flips = _kink_flips(model, orc, mask)
    if check_grads:
        ref['xe_loss'].backward()
        model.backward()
        torch.cuda.synchronize()
        g = model.rt.logical_grads()
        # leaky_relu'(x) jumps 0.2 -> 1 at 0: a pre-activation within ~1e-7 of zero may take either branch in two
        # correct fp32 evaluations.  Tight tolerance when no such flip happened on this batch, loose otherwise.
        tol = 3e-4 if flips == 0 else 0.25
        for k, v in orc.w.items():
            rg = v.grad.numpy() if v.grad is not None else np.zeros_like(v.detach().numpy())
            scale = max(1e-6, float(np.abs(rg).max()))
            err = float(np.abs(g[k] - rg).max())
            assert err < tol * scale + 2e-5, "grad %s: err %g scale %g (kink flips %d)" % (k, err, scale, flips)
    return flips


def _kink_flips(model, orc, mask):
    """Number of leaky-ReLU outputs whose SIGN differs between the HIP path and the oracle (gradient-carrying rows)."""
    pl, taps = model._plan, orc.debug_taps
    B, T, NC, BT, P = pl.B, pl.T, pl.NC, pl.BT, pl.P       # P = rows actually computed (valid positions); full_rows -> [B*T] layout
    n = 0
    for name, hip in (('S1', pl.S1), ('S2', pl.S2), ('S3', pl.S3)):
        pos, neg = taps[name]
        ref = torch.cat([pos.unsqueeze(2), neg], 2).numpy()
        h = pl.full_rows(hip, NC).cpu().numpy().reshape(ref.shape)
        n += int(((h > 0) != (ref > 0))[mask].sum())
    zin, zpos, zneg = taps['Z1']
    C = zin.shape[-1]
    Zin = pl.full_rows(pl.Z1[:P]).cpu().numpy()
    Zc = pl.full_rows(pl.cand_Z1(P), NC).cpu().numpy()
    n += int(((Zin.reshape(B, T, C) > 0) != (zin.numpy() > 0))[mask].sum())
    zc = torch.cat([zpos.unsqueeze(2), zneg], 2).numpy()
    n += int(((Zc.reshape(B, T, NC, C) > 0) != (zc > 0))[mask].sum())
    n += int(((pl.full_rows(pl.FC1).cpu().numpy().reshape(B, T, -1) > 0) != (taps['FC1'][0].numpy() > 0))[mask].sum())
    return n


def hip_leaky_signs(model, mask):
    """The branch the HIP path took at every leaky-ReLU output (valid positions), in the form NAROracle.leaky_signs expects."""
    pl = model._plan
    B, T, NC, P = pl.B, pl.T, pl.NC, pl.P
    valid = torch.from_numpy(np.asarray(mask, bool))
    out = {}
    for name, hip in (('S1', pl.S1), ('S2', pl.S2), ('S3', pl.S3)):
        h = pl.full_rows(hip, NC).float().cpu().reshape(B, T, NC, -1) > 0
        out[name] = [(h[:, :, 0], valid[:, :, None]), (h[:, :, 1:], valid[:, :, None, None])]
    zin = pl.full_rows(pl.Z1[:P]).float().cpu().reshape(B, T, -1) > 0
    zc = pl.full_rows(pl.cand_Z1(P), NC).float().cpu().reshape(B, T, NC, -1) > 0
    out['Z1'] = [(zin, valid[:, :, None]), (zc[:, :, 0], valid[:, :, None]), (zc[:, :, 1:], valid[:, :, None, None])]
    out['FC1'] = [(pl.full_rows(pl.FC1).float().cpu().reshape(B, T, -1) > 0, valid[:, :, None])]
    return out


def test_step_parity_tiny_warm_state(gpu):
    p = H.tiny_params()
    batches = synthetic.make_batches(7, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p)
    flips = [_compare_step(model, orc, *batches[i], st) for i in (3, 4, 5, 6)]
    assert sum(1 for x in flips if x == 0) >= 2, "tight gradient parity needs kink-free batches: %r" % flips


def test_step_parity_first_batch_empty_buffer(gpu):
    """nar_model.py:1080-1084: with an empty recent-clicks buffer the normalisation stats come from the batch."""
    p = H.tiny_params()
    batches = synthetic.make_batches(1, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, [])
    model, orc = H.make_pair(p)
    _compare_step(model, orc, *batches[0], st)


def test_step_parity_full_length_sessions(gpu):
    p = H.tiny_params(C=256, H=128, neg=12, batch_size=40)
    batches = synthetic.make_batches(3, 40, 8, 1000, p['session_features_config'], length_dist='full')
    st = H.warm_state(p, batches[:2])
    model, orc = H.make_pair(p, seed=11)
    _compare_step(model, orc, *batches[2], st)


def test_training_curve_matches_oracle(gpu):
    """Several optimizer steps with the state evolving: loss curve within 1e-3, Adam slots agree."""
    p = H.tiny_params()
    batches = synthetic.make_batches(8, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, [])
    model, orc = H.make_pair(p)
    for i, (f, l) in enumerate(batches):
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        model.feed_state(pop, buf)
        loss = model.train_step(model.upload_batch(f, l)).cpu().numpy()
        ref = orc.train_step(f, l, buf, pop)
        assert np.array_equal(model._plan.neg_ids.cpu().numpy(), ref['neg_items'].numpy())
        assert abs(loss[0] - float(ref['total_loss'])) < LOGIT_TOL, (i, loss, float(ref['total_loss']))
        H.update_state(st, f, l)
    # first moments everywhere (match4/bias: true gradient is 0), weights where the first moment is above the noise floor
    H.assert_adam_state_close(model, orc, p['lr'], n_steps=len(batches), w_tol=0.2)


def test_row_shards_sum_to_full_batch(gpu):
    """Data-parallel launch parameters: two row shards (row_begin, GLOBAL ids / denominators) reproduce the
    full-batch negatives and logits, and their gradient buffers SUM to the full-batch gradient (what the RCCL all-reduce
    computes)."""
    from chameleon_recsys_amd.nar import parallel
    p = H.tiny_params()
    batches = synthetic.make_batches(4, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, _ = H.make_pair(p)
    f, l = batches[3]
    model.feed_state(st.get_articles_recent_pop_norm(), st.get_recent_clicks_buffer())
    model.forward(model.upload_batch(f, l)); model.backward()
    torch.cuda.synchronize()
    full = model.outputs_numpy()
    g_full = model.rt.grads.clone()
    g_sum = torch.zeros_like(g_full)
    xe, negs, logits = 0.0, [], []
    for r in range(3):
        b, e = parallel.shard_rows(64, r, 3)
        fl, ll = parallel.slice_batch(f, l, b, e)
        model.forward(model.upload_batch(fl, ll, f, l, row_begin=b)); model.backward()
        torch.cuda.synchronize()
        out = model.outputs_numpy()
        g_sum += model.rt.grads
        xe += out['loss'][1]; negs.append(out['neg_items']); logits.append(out['logits'])
    assert np.array_equal(np.concatenate(negs), full['neg_items'])
    assert np.abs(np.concatenate(logits) - full['logits']).max() < 1e-5
    assert abs(xe - full['loss'][1]) < 1e-5
    scale = float(g_full.abs().max())
    assert float((g_sum - g_full).abs().max()) < 2e-5 * scale + 1e-7


@pytest.mark.parametrize("cell,layers,Hn", [("gru", 1, 100), ("gru", 2, 256), ("ugrnn", 2, 255), ("ugrnn", 1, 1000)])   # 1000 -> step-wise path
def test_step_parity_rnn_variants(gpu, cell, layers, Hn):
    """GRU (north-star / BASELINE config 4: 2-layer GRU, hidden 256) and stacked cells vs the oracle."""
    p = H.tiny_params(C=128, H=Hn, neg=9, batch_size=40, rnn_cell=cell, rnn_num_layers=layers)
    batches = synthetic.make_batches(4, 40, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:2])
    model, orc = H.make_pair(p, seed=5)
    flips = [_compare_step(model, orc, *batches[i], st) for i in (2, 3)]
    assert min(flips) == 0, flips


def test_adressa_shape_tiny(gpu):
    """Adressa feature schema (3 metadata embeddings, 3 large context embeddings), GRU x2, seq_len 30 -> T = 29."""
    p = synthetic.default_params(800, 32, seq_len=30, batch_size=24, neg=20, neg_from_buffer=200, buffer_size=1500, for_norm=300,
                                 C=128, H=128, dataset='adressa', rnn_cell='gru', rnn_num_layers=2, softmax_temperature=0.2,
                                 reg_weight_decay=1e-4, lr=3e-4)
    batches = synthetic.make_batches(3, 24, 30, 800, p['session_features_config'], length_dist='g1', seed=3)
    st = H.warm_state(p, batches[:2])
    model, orc = H.make_pair(p, seed=2)
    _compare_step(model, orc, *batches[2], st)


def test_microbatched_step_equals_whole_batch_step(gpu):
    """Gradient accumulation over session micro-batches (BASELINE config 5 path) == one step on the whole batch."""
    p = H.tiny_params()
    batches = synthetic.make_batches(4, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    m1, _ = H.make_pair(p)
    m2, _ = H.make_pair(p)
    f, l = batches[3]
    for m in (m1, m2):
        m.feed_state(st.get_articles_recent_pop_norm(), st.get_recent_clicks_buffer())
    a = m1.train_step(m1.upload_batch(f, l)).cpu().numpy()
    b = m2.train_step_microbatched(f, l, 24).cpu().numpy()          # 24 + 24 + 16 sessions
    assert np.abs(a - b).max() < 1e-5, (a, b)
    assert m1.rt.global_step == m2.rt.global_step == 1
    # first Adam step: dw = lr * g / (|g| + eps') is ill-conditioned where |g| ~ eps: compare the moments tightly, weights loosely
    H.assert_runtimes_close(m1.rt, m2.rt, p['lr'])


@pytest.mark.parametrize("cell,layers", [("ugrnn", 1), ("gru", 2)])
def test_valid_position_compaction_equals_padded_masked_path(gpu, cell, layers):
    """Row-wise stages on the non-padded positions only (default) == computing every padded position and masking it
    (CHAM_COMPACT=0, the reference's formulation): same negatives, logits at valid positions, loss, gradients, Adam step."""
    p = H.tiny_params(C=128, H=96, neg=9, batch_size=40, rnn_cell=cell, rnn_num_layers=layers)
    batches = synthetic.make_batches(4, 40, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:2])
    mc, _ = H.make_pair(p, seed=5)
    mp, _ = H.make_pair(p, seed=5)
    mp.rt.compact = False
    for f, l in batches[2:4]:
        outs = []
        for m in (mc, mp):
            m.feed_state(st.get_articles_recent_pop_norm(), st.get_recent_clicks_buffer())
            d = m.upload_batch(f, l)
            m.forward(d); m.backward()
            torch.cuda.synchronize()
            outs.append((m.outputs_numpy(), m.rt.grads.clone(), d))
        (oc, gc, dc), (op, gp, dp_) = outs
        assert dc['pos'] is not None and dp_['pos'] is None and dc['P'] < dp_['P']
        mask = np.arange(f['item_clicked'].shape[1])[None, :] < (np.asarray(f['session_size']).reshape(-1, 1) - 1)
        assert np.array_equal(oc['neg_items'], op['neg_items'])
        assert np.abs(oc['logits'] - op['logits'])[mask].max() < 1e-5
        assert np.abs(oc['loss'] - op['loss']).max() < 1e-5
        assert float((gc - gp).abs().max()) < 2e-5 * float(gp.abs().max()) + 1e-7
        for m in (mc, mp):
            m.apply_gradients()
        H.update_state(st, f, l)
        H.assert_runtimes_close(mc.rt, mp.rt, p['lr'])
        # Adam's first steps amplify roundoff where the true gradient is 0 (match4/bias: softmax shift invariance), which would
        # shift every logit of the next batch: restart both models from the same weights / slots
        for name in ('flat', 'm', 'v'):
            getattr(mp.rt, name).copy_(getattr(mc.rt, name))


@pytest.mark.parametrize("n_sessions", [1, 3])
def test_step_parity_handful_of_sessions(gpu, n_sessions):
    """The short last batch of an hourly file (datasets.py:136: no drop_remainder): a few sessions, a few valid positions -
    every wgrad reduction is shorter than one split-K chunk."""
    from chameleon_recsys_amd.nar import parallel
    p = H.tiny_params()
    batches = synthetic.make_batches(4, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p)
    f, l = parallel.slice_batch(*batches[3], 5, 5 + n_sessions)
    T = int(np.asarray(f['session_size']).max()) - 1          # padded_batch pads to the longest session OF THE BATCH
    f = {k: (v[:, :T] if np.asarray(v).ndim == 2 else v) for k, v in f.items()}
    l = {k: (v[:, :T] if k == 'label_next_item' else v) for k, v in l.items()}
    _compare_step(model, orc, f, l, st)


def _mode_parity(p, grad_tol, loss_of, check_fwd, flip_sensitive=True):
    """Forward + gradient parity on two batches; gradients are compared tightly on batches without a leaky-ReLU kink flip
    (see _compare_step) - at least one of the two must be flip-free."""
    batches = synthetic.make_batches(6, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p)
    clean = 0
    for f, l in batches[3:6]:
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        model.feed_state(pop, buf)
        model.forward(model.upload_batch(f, l))
        out = model.outputs_numpy()
        for v in orc.w.values():
            v.grad = None
        orc.debug_taps = {}
        ref = orc.forward(f, l, buf, pop, 'train')
        mask = ref['mask'].numpy()
        assert np.array_equal(out['neg_items'], ref['neg_items'].numpy())
        check_fwd(model, orc, out, ref, mask, f, l, buf, pop)
        flips = _kink_flips(model, orc, mask)
        loss_of(ref).backward()
        model.backward()
        torch.cuda.synchronize()
        g = model.rt.logical_grads()
        tol = grad_tol if (flips == 0 or not flip_sensitive) else 0.25
        clean += (flips == 0 or not flip_sensitive)
        for k, v in orc.w.items():
            rg = v.grad.numpy() if v.grad is not None else np.zeros_like(v.detach().numpy())
            scale = max(1e-6, float(np.abs(rg).max()))
            assert float(np.abs(g[k] - rg).max()) < tol * scale + 2e-5, (k, flips)
        H.update_state(st, f, l)
    assert clean >= 1


def test_step_parity_bf16_compute_mode(gpu):
    """BASELINE config 3 arithmetic: every Dense / matmul with bf16-rounded operands + fp32 accumulation (fp32 storage,
    softmax, loss, Adam).  Forward: against the oracle emulating exactly that rounding - logits 2e-3 / loss 1e-3 (fp32
    accumulation order only) - and within 3e-2 of the fp32 oracle's loss.  Gradients: bf16 noise is amplified by the
    cancellation in the small context / embedding gradients, and the HIP path rounds a few operands after summing them over a
    click's candidates (dU, dV) where autograd rounds per row, so two correct bf16 evaluations differ by more than they each
    differ from fp32; the check is therefore accuracy against the FP32 oracle: the HIP bf16 gradient must be as close to it as
    the emulated bf16 gradient is (x3 + 2 % of the tensor's max)."""
    from oracle.nar_oracle import NAROracle
    p = H.tiny_params(gemm_dtype='bf16')
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p)
    assert model.rt.gemm_dtype == 'bf16' and orc.gemm_dtype == 'bf16'
    p32 = dict(p); p32['gemm_dtype'] = 'f32'
    orc32 = NAROracle(p32, weights=orc.weights_numpy())
    for f, l in batches[3:5]:
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        model.feed_state(pop, buf)
        model.forward(model.upload_batch(f, l))
        out = model.outputs_numpy()
        grads = {}
        for name, o in (("bf16", orc), ("f32", orc32)):
            for v in o.w.values():
                v.grad = None
            ref = o.forward(f, l, buf, pop, 'train')
            if name == "bf16":
                mask = ref['mask'].numpy()
                assert np.array_equal(out['neg_items'], ref['neg_items'].numpy())
                assert np.abs(out['logits'] - ref['logits'].detach().numpy())[mask].max() < 2e-3
                assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < 1e-3
            else:
                assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < 3e-2
            ref['xe_loss'].backward()
            grads[name] = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros_like(v.detach().numpy())) for k, v in o.w.items()}
        model.backward()
        torch.cuda.synchronize()
        g = model.rt.logical_grads()
        for k in g:
            scale = max(1e-6, float(np.abs(grads["f32"][k]).max()))
            e_hip = float(np.abs(g[k] - grads["f32"][k]).max())
            e_emu = float(np.abs(grads["bf16"][k] - grads["f32"][k]).max())
            assert e_hip < 3.0 * e_emu + 2e-2 * scale + 2e-5, (k, e_hip, e_emu, scale)
        H.update_state(st, f, l)


def test_step_parity_novelty_regularised_loss(gpu):
    """--novelty_reg_factor > 0 (nar_model.py:673-683): loss and gradients with the novelty term."""
    p = H.tiny_params(novelty_reg_factor=0.3)

    def check_fwd(model, orc, out, ref, mask, f, l, buf, pop):
        plain = float(ref['xe_loss'].detach()) + float(ref['reg_loss'].detach())
        assert abs(plain - float(ref['total_loss'].detach())) > 0.05        # the term is not negligible in this set-up
        assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < LOGIT_TOL
    _mode_parity(p, 3e-4, lambda ref: ref['total_loss'] - ref['reg_loss'], check_fwd)


def test_training_is_bit_reproducible(gpu):
    """Two runs of the same four optimizer steps (device-resident state, two-lane stream schedule, presampled negatives) end with
    BIT-IDENTICAL weights and Adam slots: every reduction of the step has a fixed order (split-K partials, column sums, the PreCAR
    slot scatter, the embedding-table gradients) - no float atomics anywhere.  SURVEY section 5 asks for exactly this mode."""
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    p = H.tiny_params()
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    runs = []
    for _ in range(2):
        model, _o = H.make_pair(p)
        st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'],
                                     p['recent_clicks_for_normalization'], 1000)
        dev = [model.upload_batch(f, l) for f, l in batches]
        losses = []
        for i, d in enumerate(dev):
            model.feed_state(st, st)
            losses.append(model.train_step(d).clone())
            st.update_from_device_batch(d['aci'], d['g_event_ts'])
            if i + 1 < len(dev):
                model.presample(dev[i + 1])
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu(), model.rt.flat.cpu().clone(), model.rt.m.cpu().clone(), model.rt.v.cpu().clone()))
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)


def test_step_parity_fewer_candidates_than_negatives(gpu):
    """A catalog too small to offer N unique valid candidates: every click's negatives end in zero padding (nar_model.py:1252) and
    the zero-padding item row collects the gradient of ALL of them (the reference backpropagates through padded negatives too)."""
    p = H.tiny_params(n_items=32, neg=40, neg_from_buffer=30, buffer_size=300, for_norm=100)
    batches = synthetic.make_batches(5, 64, 8, 32, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p)
    for i in (3, 4):
        _compare_step(model, orc, *batches[i], st)
        neg = model._plan.neg_ids.cpu().numpy()
        valid = np.asarray(batches[i][1]['label_next_item']) != 0
        assert (neg[valid] == 0).mean() > 0.125, "this case is meant to exercise > 12.5 % padded negatives"


@pytest.mark.parametrize("cell,layers,keep", [("ugrnn", 1, 0.8), ("gru", 2, 0.9)])
def test_step_parity_dropout(gpu, cell, layers, keep):
    """dropout_keep_prob < 1 (the reference's hyper-tuning config searches 0.8 / 0.9: nar_mlengine_hypertuning.yaml:40-45): dropout on
    the three feature tensors (nar_model.py:338, 352, 368 - the dense, un-factorised PreCAR path), on FC1 (:418) and on every recurrent
    layer's output (:1331), with the counter-based masks both sides share: bit-exact negatives, logits / loss 1e-3, gradients."""
    p = H.tiny_params(C=128, H=96, neg=9, batch_size=40, rnn_cell=cell, rnn_num_layers=layers, dropout_keep_prob=keep)
    batches = synthetic.make_batches(5, 40, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:2])
    model, orc = H.make_pair(p, seed=5)
    assert model.keep_prob == keep
    flips = []
    for i in (2, 3, 4):
        flips.append(_compare_step(model, orc, *batches[i], st))
        x_neg = orc.forward(*batches[i], st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm(), 'train')['x_neg']
        mask = np.asarray(batches[i][1]['label_next_item']) != 0
        dropped = float((x_neg.detach().numpy()[mask] == 0).mean())
        assert dropped > (1.0 - keep) * 0.8, dropped            # the masks really are applied (one-hot zeros come on top)
        model.rt.global_step += 1; orc.global_step += 1          # a new mask / sample key per step
    assert min(flips) == 0, flips


def test_dropout_is_identity_in_eval_and_independent_of_row_shards(gpu):
    """EVAL never drops (nar_trainer_gcom.py:245); TRAIN masks are keyed by the GLOBAL session row and time step, so two row shards
    reproduce the full batch's logits (data-parallel / micro-batched steps see the same masks)."""
    from chameleon_recsys_amd.nar import parallel
    p = H.tiny_params(dropout_keep_prob=0.8)
    batches = synthetic.make_batches(4, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, _ = H.make_pair(p)
    f, l = batches[3]
    model.feed_state(st.get_articles_recent_pop_norm(), st.get_recent_clicks_buffer())
    model.forward(model.upload_batch(f, l)); model.backward()
    torch.cuda.synchronize()
    full = model.outputs_numpy()
    g_full = model.rt.grads.clone()
    g_sum, logits = torch.zeros_like(g_full), []
    for r in range(2):
        b, e = parallel.shard_rows(64, r, 2)
        fl, ll = parallel.slice_batch(f, l, b, e)
        model.forward(model.upload_batch(fl, ll, f, l, row_begin=b)); model.backward()
        torch.cuda.synchronize()
        logits.append(model.outputs_numpy()['logits']); g_sum += model.rt.grads
    assert np.abs(np.concatenate(logits) - full['logits']).max() < 1e-5
    assert float((g_sum - g_full).abs().max()) < 2e-5 * float(g_full.abs().max()) + 1e-7


def test_step_parity_non_default_log_bases(gpu):
    """NARModuleModel(elapsed_days_smooth_log_base, popularity_smooth_log_base) (nar_model.py:122-123; log_base :28-34): the bases of
    the recency feature (:1071-1075), the novelty feature (:1148) and the novelty regulariser (:544) are launch scalars - steps with
    bases far from the defaults (and the regulariser on) against the oracle, forward and gradients; and the bases really are used:
    the default-base oracle disagrees on the same inputs."""
    from oracle.nar_oracle import NAROracle
    p = H.tiny_params(novelty_reg_factor=0.3, elapsed_days_smooth_log_base=2.5, popularity_smooth_log_base=10.0)
    p0 = {k: v for k, v in p.items() if k not in ('elapsed_days_smooth_log_base', 'popularity_smooth_log_base')}
    seen = []

    def check_fwd(model, orc, out, ref, mask, f, l, buf, pop):
        assert model.elapsed_days_smooth_log_base == 2.5 and model.popularity_smooth_log_base == 10.0
        assert np.abs(out['logits'] - ref['logits'].detach().numpy())[mask].max() < LOGIT_TOL
        assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < LOGIT_TOL
        ref0 = NAROracle(p0, weights=orc.weights_numpy()).forward(f, l, buf, pop, 'train')
        seen.append(abs(float(ref0['total_loss'].detach()) - float(out['loss'][0])))
    _mode_parity(p, 3e-4, lambda ref: ref['total_loss'] - ref['reg_loss'], check_fwd)
    assert max(seen) > 1e-2, seen
    with pytest.raises(ValueError):
        H.make_pair(H.tiny_params(popularity_smooth_log_base=1.0))


def test_step_parity_numerical_article_metadata(gpu):
    """'numerical' article features (nar_model.py:755-757 reached from get_item_features :926-939): a float-valued one (stored as
    float32 bits in the metadata table) and an integer-valued one, next to the categorical column - forward and gradients."""
    p = H.tiny_params()
    rng = np.random.default_rng(5)
    n = p['content_article_embeddings_matrix'].shape[0]
    acfg = p['articles_features_config']
    acfg['text_length_z'] = {'type': 'numerical', 'dtype': 'float'}
    acfg['n_images'] = {'type': 'numerical', 'dtype': 'int'}
    p['articles_metadata']['text_length_z'] = rng.standard_normal(n).astype(np.float32)
    p['articles_metadata']['n_images'] = rng.integers(0, 5, size=n).astype(np.int64)
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p)
    L = model.rt.layout
    assert L.meta_is_float == {'text_length_z'} and L.f_item == 37 + 2 + 64 + 44 + 2
    _compare_step(model, orc, *batches[3], st)
    bad = dict(p); bad['articles_features_config'] = dict(acfg); bad['articles_features_config']['text_length_z'] = {'type': 'numerical', 'dtype': 'int'}
    from chameleon_recsys_amd.nar.nar_model import NARRuntime
    with pytest.raises(ValueError):
        NARRuntime(bad)


def test_step_parity_bf16_with_dropout(gpu):
    """dropout_keep_prob < 1 in the bf16 configuration (nar_model.py:338, 352, 368, 418, 1331 x BASELINE configs[2] arithmetic): the dense
    PreCAR layer on bf16-rounded operands, its candidate rows stored bf16.  Forward against the oracle that emulates the rounding with the
    same counter-based masks (logits 4e-3, loss 1e-3); gradients as accurate against the FP32 oracle (same masks) as the emulation is
    (the criterion of test_step_parity_bf16_compute_mode)."""
    from oracle.nar_oracle import NAROracle
    p = H.tiny_params(gemm_dtype='bf16', dropout_keep_prob=0.85, C=128, H=96, neg=9, batch_size=40)
    batches = synthetic.make_batches(5, 40, 8, 1000, p['session_features_config'], length_dist='g1')
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p, seed=5)
    assert model.rt.gemm_dtype == 'bf16' and model.keep_prob == 0.85
    p32 = dict(p); p32['gemm_dtype'] = 'f32'
    orc32 = NAROracle(p32, weights=orc.weights_numpy())
    for f, l in batches[3:5]:
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        model.feed_state(pop, buf)
        model.forward(model.upload_batch(f, l))
        out = model.outputs_numpy()
        grads = {}
        for name, o in (("bf16", orc), ("f32", orc32)):
            for v in o.w.values():
                v.grad = None
            o.global_step = model.rt.global_step
            ref = o.forward(f, l, buf, pop, 'train')
            if name == "bf16":
                mask = ref['mask'].numpy()
                assert np.array_equal(out['neg_items'], ref['neg_items'].numpy())
                assert np.abs(out['logits'] - ref['logits'].detach().numpy())[mask].max() < 4e-3
                assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < 1e-3
                dropped = float((ref['x_neg'].detach().numpy()[np.asarray(l['label_next_item']) != 0] == 0).mean())
                assert dropped > 0.15 * 0.8, dropped
            else:
                assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < 3e-2
            ref['xe_loss'].backward()
            grads[name] = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros_like(v.detach().numpy())) for k, v in o.w.items()}
        model.backward()
        torch.cuda.synchronize()
        g = model.rt.logical_grads()
        for k in g:
            scale = max(1e-6, float(np.abs(grads["f32"][k]).max()))
            e_hip = float(np.abs(g[k] - grads["f32"][k]).max())
            e_emu = float(np.abs(grads["bf16"][k] - grads["f32"][k]).max())
            assert e_hip < 3.0 * e_emu + 2e-2 * scale + 2e-5, (k, e_hip, e_emu, scale)
        H.update_state(st, f, l)
        model.rt.global_step += 1


def test_training_with_tile_blocked_planes_is_bit_identical(gpu, monkeypatch):
    """CHAM_H2_BLOCKED=1 (round 6: the candidate rows' fp16 planes stored [row tiles of 256][C / 32][256][32]; both producers and the three
    plane GEMMs in that layout) against the default row-major planes: the SAME four optimizer steps, ragged G1-like batches (valid
    positions vary from step to step: a row tile keeps rows of an earlier, longer step beyond the current ones - the TN form masks them),
    bit-identical losses, weights and Adam slots; the launch counters prove the blocked kernels ran."""
    import ctypes
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    p = H.tiny_params(C=256, neg=40)
    batches = synthetic.make_batches(5, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    runs, counts = [], []
    for blocked in ("0", "1"):
        monkeypatch.setenv("CHAM_H2_BLOCKED", blocked)
        model, _o = H.make_pair(p)
        assert model.rt.h2 and model.rt.h2_blocked == (blocked == "1")
        c = (ctypes.c_longlong * 8)()
        model.rt.lib.cham_gemm_h2_launch_counts(c, 1)
        st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'], p['recent_clicks_for_normalization'], 1000)
        dev = [model.upload_batch(f, l) for f, l in batches]
        losses = []
        for i, d in enumerate(dev):
            model.feed_state(st, st)
            losses.append(model.train_step(d).clone())
            st.update_from_device_batch(d['aci'], d['g_event_ts'])
        torch.cuda.synchronize()
        model.rt.lib.cham_gemm_h2_launch_counts(c, 0)
        counts.append(list(c))
        z1 = model._plan.cand_Z1().cpu()
        runs.append((torch.stack(losses).cpu(), model.rt.flat.cpu().clone(), model.rt.m.cpu().clone(), model.rt.v.cpu().clone(), z1))
    assert counts[0][3] == 0 and counts[0][4] == 0
    assert counts[1][3] == 2 * len(batches) and counts[1][4] == len(batches), counts      # NT forward + dgrad, TN weight gradient per step
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)


def _train_with_env(monkeypatch, env, p, batches, full_lanes=False):
    """Four optimizer steps under the given environment switches -> (losses, weights, Adam m, v), all on the host."""
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model, _o = H.make_pair(p)
    st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'], p['recent_clicks_for_normalization'], 1000)
    dev = [model.upload_batch(f, l) for f, l in batches]
    losses = []
    for d in dev:
        model.feed_state(st, st)
        losses.append(model.train_step(d).clone())
        st.update_from_device_batch(d['aci'], d['g_event_ts'])
    torch.cuda.synchronize()
    for k in env:
        monkeypatch.delenv(k)
    return model, (torch.stack(losses).cpu(), model.rt.flat.cpu().clone(), model.rt.m.cpu().clone(), model.rt.v.cpu().clone())


@pytest.mark.parametrize("length_dist", ["g1", "full"])
def test_round6_schedule_switches_are_bit_identical_and_summation_switches_close(gpu, monkeypatch, length_dist):
    """The round-6 switches of the backward pass against their `=0` arms over four optimizer steps with the state evolving.
    SCHEDULE only (which lane a kernel runs on) - CHAM_TAIL_SPLIT, CHAM_LOSS_SIDE: bit-identical losses, weights and Adam slots.
    Another fixed SUMMATION ORDER - CHAM_DGRAD_GROUPSUM (dU from the CAR dgrad's epilogue), CHAM_FEATURE_BWD_WS (coalesced dgamma / dbeta
    column sums): same training to fp32 roundoff (losses 2e-5, Adam first moments 1e-4 of the largest), and the group-sum launch counter proves which
    path produced dU.  `full`: every session full length (the full-batch lane schedule: third lane in use); `g1`: ragged lengths."""
    import ctypes
    p = H.tiny_params(C=256, neg=40)
    batches = synthetic.make_batches(4, 64, 8, 1000, p['session_features_config'], length_dist=length_dist)
    c = (ctypes.c_longlong * 8)()
    base_model, base = _train_with_env(monkeypatch, {}, p, batches)
    base_model.rt.lib.cham_gemm_h2_launch_counts(c, 1)
    assert base_model.rt.dgrad_groupsum and base_model.rt.loss_side and base_model.rt.tail_split and base_model.rt.feature_bwd_ws
    for name in ("CHAM_TAIL_SPLIT", "CHAM_LOSS_SIDE"):
        _m, run = _train_with_env(monkeypatch, {name: "0"}, p, batches)
        for a, b in zip(base, run):
            assert torch.equal(a, b), name
    base_model.rt.lib.cham_gemm_h2_launch_counts(c, 1)
    m1, _ = _train_with_env(monkeypatch, {}, p, batches)
    m1.rt.lib.cham_gemm_h2_launch_counts(c, 1)
    assert c[5] == len(batches), list(c)                     # default: every step's dgrad left the per-click sums
    for name in ("CHAM_DGRAD_GROUPSUM", "CHAM_FEATURE_BWD_WS"):
        m0, run = _train_with_env(monkeypatch, {name: "0"}, p, batches)
        m0.rt.lib.cham_gemm_h2_launch_counts(c, 1)
        assert c[5] == (0 if name == "CHAM_DGRAD_GROUPSUM" else len(batches)), (name, list(c))
        assert float((base[0] - run[0]).abs().max()) < 2e-5, name
        # (weights are no yardstick: one TF-Adam step moves an entry by ~lr whatever |g| is, so where the true gradient is ~0 the sign of the
        # roundoff decides - tests/helpers.py; Adam's first moments are linear in the gradients)
        assert float((base[2] - run[2]).abs().max()) < 1e-4 * float(base[2].abs().max()), (name, float((base[2] - run[2]).abs().max()), float(base[2].abs().max()))
