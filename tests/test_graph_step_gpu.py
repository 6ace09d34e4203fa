"""The training step replayed from a hipGraph (nar_model.GraphedTrainStep, round 6; the reference runs a step as ONE session.run,
/root/reference/nar_module/nar/nar_model.py:1434-1470) against the eager step: BIT-IDENTICAL losses at every step, weights, Adam slots, recent-clicks
state and negatives - same kernels, same arguments, same order per lane; the only difference is who submits them.  Also: the device step-scalar
record against the by-value entry points (CHAM_DEV_SCALARS=0), and the conditions under which a step is not capturable."""
import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(p, batches, n_eager, graphed, monkeypatch=None, dev_scalars="1"):
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.nar_model import GraphedTrainStep
    if monkeypatch is not None:
        monkeypatch.setenv("CHAM_DEV_SCALARS", dev_scalars)
    model, _o = H.make_pair(p)
    n_items = p['content_article_embeddings_matrix'].shape[0]
    st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'], p['recent_clicks_for_normalization'], n_items)
    dev = [model.upload_batch(f, l) for f, l in batches]
    gs = GraphedTrainStep(model, st)
    losses = []
    for i, d in enumerate(dev[:-1]):
        if graphed and i >= n_eager:
            assert gs.supports(d) is None, gs.supports(d)
            losses.append(gs.step(d, dev[i + 1]).clone())
        else:
            model.feed_state(st, st)
            losses.append(model.train_step(d).clone())
            st.update_from_device_batch(d['aci'], d['g_event_ts'])
            model.presample(dev[i + 1])
    torch.cuda.synchronize()
    pl = model._plan
    # the negatives drawn for the NEXT batch (key = the next step): eagerly they sit in the sampler set that is not the current one, after a
    # replay in the set the captured forward reads (the graph's last nodes copy them there)
    neg_next = pl._samp[pl._samp_cur if (graphed and gs.replays) else 1 - pl._samp_cur]['neg_ids']
    out = dict(neg_next=neg_next.cpu().clone(), losses=torch.stack(losses).cpu(), flat=model.rt.flat.cpu().clone(), m=model.rt.m.cpu().clone(), v=model.rt.v.cpu().clone(),
               buf=st.buf_ids.cpu().clone(), buf_ts=st.buf_ts.cpu().clone(), pop=st.pop_norm.cpu().clone(),
               logits=pl.logits.cpu().clone(), step=model.rt.global_step, n_updates=st.n_updates)
    return out, gs, model, st, dev


@pytest.mark.parametrize("C,neg", [(128, 10), (256, 40)])
def test_graph_replay_is_bit_identical_to_the_eager_step(gpu, C, neg):
    """C = 128: six-product arithmetic, unfused scorer dgrad; C = 256, 41 candidates: two-fp16-plane CAR GEMMs + the fused scorer dgrad (the
    default arithmetic at the G1 shape).  Ten steps of full-length sessions; the graphed run switches to replays after two eager steps."""
    p = H.tiny_params(C=C, neg=neg)
    batches = synthetic.make_batches(11, 64, 8, 1000, p['session_features_config'], length_dist='full')
    eager, _, _, _, _ = _run(p, batches, 10, False)
    graph, gs, model, st, dev = _run(p, batches, 2, True)
    assert gs.graph is not None and gs.replays == 8
    assert eager['step'] == graph['step'] == 10 and eager['n_updates'] == graph['n_updates']
    for k in ('losses', 'flat', 'm', 'v', 'buf', 'buf_ts', 'pop', 'neg_next', 'logits'):
        assert torch.equal(eager[k], graph[k]), k
    # ... and an EAGER step after the replays continues the same trajectory (python-side bookkeeping of the replays: global step, state)
    model.feed_state(st, st)
    l_after = model.train_step(dev[-1]).clone().cpu()
    ref_model_run, _, _, _, _ = _run(p, batches + [batches[0]], 11, False)
    assert torch.equal(ref_model_run['losses'][-1], l_after)


def test_device_scalar_record_matches_the_by_value_entry_points(gpu, monkeypatch):
    """CHAM_DEV_SCALARS=1 (default: sampler key, max time stamp, sum(mask), lr_t read from the device record) against =0 (by value): same bits,
    on ragged G1-like batches (sum(mask) and the valid-position count change from step to step) with presampling."""
    p = H.tiny_params(C=256, neg=40)
    batches = synthetic.make_batches(7, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    a, _, ma, _, _ = _run(p, batches, 6, False, monkeypatch, "1")
    b, _, mb, _, _ = _run(p, batches, 6, False, monkeypatch, "0")
    assert ma.rt.dev_scalars and not mb.rt.dev_scalars
    for k in ('losses', 'flat', 'm', 'v', 'buf', 'pop', 'neg_next'):
        assert torch.equal(a[k], b[k]), k


def test_what_is_not_capturable_is_refused_with_a_reason(gpu):
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, DeviceClickedItemsState
    from chameleon_recsys_amd.nar.nar_model import GraphedTrainStep
    p = H.tiny_params()
    model, _o = H.make_pair(p)
    ragged = synthetic.make_batches(2, 64, 8, 1000, p['session_features_config'], length_dist='g1')
    full = synthetic.make_batches(2, 64, 8, 1000, p['session_features_config'], length_dist='full')
    st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'], p['recent_clicks_for_normalization'], 1000)
    gs = GraphedTrainStep(model, st)
    d_full, d_ragged = model.upload_batch(*full[0]), model.upload_batch(*ragged[0])
    assert "empty recent-clicks state" in gs.supports(d_full)
    st.update_from_device_batch(d_full['aci'], d_full['g_event_ts'])
    assert "no eager training step" in gs.supports(d_full)            # the shape is cold: buffers / kernel attributes are set up by an eager step
    model.feed_state(st, st)
    model.train_step(d_full)
    assert gs.supports(d_full) is None
    assert "compacted" in gs.supports(d_ragged)
    with pytest.raises(RuntimeError, match="compacted"):
        gs.step(d_ragged, d_ragged)
    host = ClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'], p['recent_clicks_for_normalization'], 1000)
    assert "host-side" in GraphedTrainStep(model, host).supports(d_full)


def test_graph_replay_with_an_unexpected_batch_order(gpu):
    """step(d, d_next) with d other than the previous call's d_next (the replay had drawn the negatives of THAT batch): the slot is reloaded
    and the negatives are drawn eagerly - still bit-identical to the eager run over the same batch order."""
    p = H.tiny_params(C=256, neg=40)
    batches = synthetic.make_batches(6, 64, 8, 1000, p['session_features_config'], length_dist='full')
    order = [0, 1, 2, 4, 3, 5, 1]              # (the last entry is only the "next" of the last step)
    seq = [batches[i] for i in order]
    eager, _, _, _, _ = _run(p, seq, 10, False)
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.nar_model import GraphedTrainStep
    model, _o = H.make_pair(p)
    st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'], p['recent_clicks_for_normalization'], 1000)
    dev = [model.upload_batch(f, l) for f, l in batches]
    gs = GraphedTrainStep(model, st)
    losses = []
    for n, i in enumerate(order[:-1]):
        if n < 2:
            model.feed_state(st, st)
            losses.append(model.train_step(dev[i]).clone())
            st.update_from_device_batch(dev[i]['aci'], dev[i]['g_event_ts'])
        else:      # the announced next batch is ALWAYS wrong here (i + 1 is never what follows in `order` from n = 2 on)
            losses.append(gs.step(dev[i], dev[(i + 1) % 6]).clone())
    torch.cuda.synchronize()
    assert torch.equal(torch.stack(losses).cpu(), eager['losses']) and torch.equal(model.rt.flat.cpu(), eager['flat'])
