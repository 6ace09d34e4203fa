"""BASELINE configs[1] at FULL size (46k articles, 250-d ACE, seq_len 20, batch 256, 50 negatives, C=1024, H=255): the
size-independent properties of the HIP path (the oracle comparison at this very batch size - ~10 s of oracle time per evaluation - is
tests/test_g1shape_parity_gpu.py::test_step_parity_g1_shape_headline_batch).
  * row-shard additivity: two half-batches (global ids / denominators) reproduce the full batch's negatives and logits, their
    gradient buffers sum to the full-batch gradient (the property data parallelism and micro-batching rest on);
  * softmax rows sum to 1, padded clicks contribute nothing, negatives never contain the session's own items and are unique
    per click (the reference's own sampler test properties, candidate_sampling_tests.py);
  * padded parameter lanes (rnn_units 255 -> 256, feature widths -> multiples of 4) stay exactly zero through Adam steps;
  * the loss goes down over a few optimizer steps on a fixed batch."""
import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import parallel, synthetic
from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g1():
    p = synthetic.default_params(46000, 250, seq_len=20, batch_size=256, neg=50, neg_from_buffer=3000, buffer_size=20000,
                                 for_norm=2000, C=1024, H=255)
    batches = synthetic.make_batches(4, 256, 20, 46000, p['session_features_config'], length_dist='g1', sessions_per_hour=512)
    return p, batches


def test_full_size_shard_additivity_and_sampler_properties(gpu, g1):
    p, batches = g1
    model, _ = H.make_pair(p, seed=1)
    st = DeviceClickedItemsState(1.0, 20000, 2000, 46000)
    for f, l in batches[:3]:
        aci = np.concatenate([f['item_clicked'], l['label_last_item']], 1)
        st.update_from_device_batch(torch.from_numpy(aci).cuda(), torch.from_numpy(f['event_timestamp']).cuda())
    f, l = batches[3]
    model.feed_state(st, st)
    model.forward(model.upload_batch(f, l)); model.backward()
    torch.cuda.synchronize()
    full = model.outputs_numpy()
    g_full = model.rt.grads.clone()
    # --- softmax / mask / sampler properties at full size
    mask = np.arange(f['item_clicked'].shape[1])[None, :] < (f['session_size'] - 1)[:, None]
    assert np.abs(full['probs'].sum(-1) - 1.0)[mask].max() < 1e-5
    neg = full['neg_items']
    assert not neg[~mask].any()                                        # padded clicks -> all-zero negatives
    aci = np.concatenate([f['item_clicked'], l['label_last_item']], 1)
    for b in range(0, 256, 17):
        sess = set(aci[b].tolist()) - {0}
        for t in np.nonzero(mask[b])[0]:
            nz = neg[b, t][neg[b, t] != 0]
            assert len(nz) == 50 and len(np.unique(nz)) == 50 and not (set(nz.tolist()) & sess)
    assert np.isfinite(full['loss']).all() and 0.0 < full['loss'][1] < 8.0
    # --- two row shards
    g_sum = torch.zeros_like(g_full)
    negs, logits, xe = [], [], 0.0
    for r in range(2):
        b, e = parallel.shard_rows(256, r, 2)
        fl, ll = parallel.slice_batch(f, l, b, e)
        model.forward(model.upload_batch(fl, ll, f, l, row_begin=b)); model.backward()
        torch.cuda.synchronize()
        out = model.outputs_numpy()
        g_sum += model.rt.grads
        negs.append(out['neg_items']); logits.append(out['logits']); xe += out['loss'][1]
    assert np.array_equal(np.concatenate(negs), neg)
    assert np.abs(np.concatenate(logits) - full['logits']).max() < 1e-4
    assert abs(xe - full['loss'][1]) < 1e-5
    assert float((g_sum - g_full).abs().max()) < 1e-4 * float(g_full.abs().max()) + 1e-7


def test_full_size_training_steps_keep_pads_zero_and_reduce_loss(gpu, g1):
    p, batches = g1
    model, _ = H.make_pair(p, seed=2)
    model.lr = 1e-3
    L, rt = model.rt.layout, model.rt
    st = DeviceClickedItemsState(1.0, 20000, 2000, 46000)
    f, l = batches[0]
    losses = []
    d = None
    for i in range(6):
        model.feed_state(st, st)
        d = model.upload_batch(f, l)
        losses.append(float(model.train_step(d).cpu().numpy()[1]))
        if i == 0:
            st.update_from_device_batch(d['aci'], d['g_event_ts'])     # leave the empty-buffer regime after the first step
    assert losses[-1] < losses[1] - 0.02, losses                       # same batch, same state from step 1 on
    H_, Hp = L.H, L.Hp
    Wx, Wh, b = rt.p('rnn0/Wx'), rt.p('rnn0/Wh'), rt.p('rnn0/b')
    assert not Wx[:, H_:Hp].any() and not Wx[:, Hp + H_:].any()
    assert not Wh[H_:, :].any() and not Wh[:, H_:Hp].any() and not Wh[:, Hp + H_:].any()
    assert not b[H_:Hp].any() and not b[Hp + H_:].any()
    assert not rt.p('Wf1')[H_:, :].any()
    assert not rt.p('W1c')[L.f_ctx:, :].any() and not rt.p('W1i')[L.f_item:, :].any()
    assert not rt.flat[sum(e.size for e in L.entries.values()):].any()
