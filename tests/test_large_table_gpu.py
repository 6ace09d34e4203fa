"""BASELINE.json configs[4] catalog size: the trainable item table of 5 M articles x floor(8 * 5e6^0.25) = 378 columns is 7.56 GB
(nar_model.py:25-26, 911-919) - every row above 2 840 947 starts beyond the 2^32-byte offset.  The other config-5 parity tests run the
config's WIDTHS on a 50 k-article catalog; nothing there touches such a row.

Here the SAME optimizer step runs twice on the full-size table: once with the batch's item ids in rows 1 .. 40 000 ("low"), once with
every non-pad id moved up by 4 500 000 ("high": byte offsets 6.8 - 6.9 GB into the table, 5.1 G into the ACE matrix) and the catalog /
table rows moved with them.  The remap is monotone (0 stays 0), so the sampler's integer path, every gather, the radix-sorted row
grouping, the embedding-gradient segments, L2 and TF-Adam see the same values in another place: negatives equal after the remap,
logits / probabilities / cross-entropy, every dense gradient, the touched rows' gradients and post-Adam rows + slots BIT-identical (the
L2 loss - a sum over all 1.9 G table entries in another order - to 1e-5), no row outside the remapped set written.  A 32-bit byte offset in a buffer descriptor, a radix sort with too few digit passes for 23-bit ids or an int32
element index in k_adam_tf / k_sumsq_partial breaks one of these."""
import gc

import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import config, synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu

N_BIG, K_SMALL, OFFSET, ACE_DIM, NEG, B = 5_000_000, 40_000, 4_500_000, 128, 200, 24
FILL = np.float32(0.01)          # every untouched row of the table (Adam + L2 move all of them, nar_model.py:708-722)


def _remap(ids, off):
    ids = np.asarray(ids)
    return np.where(ids != 0, ids + off, 0).astype(ids.dtype)


def _params(off):
    """Estimator params for the 5 M-article catalog whose only populated rows are remap(0 .. K_SMALL-1); the rest of the catalog is
    constant (zero ACE rows, one creation time stamp, category 0) - never read by the step."""
    p = synthetic.default_params(K_SMALL, ACE_DIM, seq_len=20, batch_size=B, neg=NEG, neg_from_buffer=3000, buffer_size=20000,
                                 for_norm=2000, C=1024, H=255, seed=3)
    ace_s, meta_s = p['content_article_embeddings_matrix'], p['articles_metadata']
    rows = _remap(np.arange(K_SMALL, dtype=np.int64), off)
    ace = np.zeros((N_BIG, ACE_DIM), np.float32)
    ace[rows] = ace_s
    meta = {}
    for k, v in meta_s.items():
        a = np.full(N_BIG, v[0], dtype=v.dtype)
        a[rows] = v
        meta[k] = a
    meta['article_id'] = np.arange(N_BIG, dtype=np.int64)
    p['content_article_embeddings_matrix'], p['articles_metadata'] = ace, meta
    p['session_features_config'] = config.get_session_features_config_gcom(N_BIG)
    p['articles_features_config'] = config.get_articles_features_config_gcom(N_BIG)
    return p, rows


def _run(off):
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, batch_clicks_for_state
    from chameleon_recsys_amd.nar.layout import ParamLayout
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime
    p, rows = _params(off)
    L = ParamLayout(p['session_features_config'], p['articles_features_config'], N_BIG, ACE_DIM, 1024, 255)
    assert L.entries['items_embedding'].shape == (N_BIG, 378) and L.entries['items_embedding'].size * 4 > (7 << 30)
    w = L.init_logical(7, max_random_elems=50_000_000)            # dense weights random (seeded), the 1.9 G-element table zero ...
    rng = np.random.default_rng(8)
    small = rng.uniform(-0.05, 0.05, size=(K_SMALL, 378)).astype(np.float32)
    tab = w['items_embedding']
    tab[:] = FILL                                                 # ... -> constant, with the K_SMALL populated rows at their (re)mapped place
    tab[rows] = small
    rt = NARRuntime(p, weights=w)
    del w, tab
    gc.collect()
    model = NARModuleModel(ModeKeys.TRAIN, None, None, p['session_features_config'], p['articles_features_config'], B, p['lr'], 1.0, NEG,
                           3000, p['content_article_embeddings_matrix'], softmax_temperature=0.1, reg_weight_decay=1e-5,
                           recent_clicks_buffer_max_size=20000, recent_clicks_for_normalization=2000,
                           articles_metadata=p['articles_metadata'], CAR_embedding_size=1024, rnn_units=255, runtime=rt)
    batches = synthetic.make_batches(4, B, 20, K_SMALL, p['session_features_config'], length_dist='g1', sessions_per_hour=4 * B, seed=31)
    st = ClickedItemsState(1.0, 20000, 2000, N_BIG)
    mapped = []
    for f, l in batches:
        f, l = dict(f), dict(l)
        f['item_clicked'] = _remap(f['item_clicked'], off)
        l['label_next_item'], l['label_last_item'] = _remap(l['label_next_item'], off), _remap(l['label_last_item'], off)
        mapped.append((f, l))
    for f, l in mapped[:3]:
        st.update_items_state(*batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp']))
    f, l = mapped[3]
    model.feed_state(st.get_articles_recent_pop_norm().copy(), st.get_recent_clicks_buffer().copy())
    model.forward(model.upload_batch(f, l))
    out = model.outputs_numpy()
    model.backward()
    torch.cuda.synchronize()
    e = L.entries['items_embedding']
    gtab = rt.g('items_embedding')
    ridx = torch.from_numpy(rows).to(rt.device)
    res = dict(rows=rows, neg=out['neg_items'], logits=out['logits'], probs=out['probs'], loss=out['loss'],
               g_rows=gtab[ridx].cpu().numpy(), g_touched=torch.nonzero(gtab.abs().amax(1) != 0).flatten().cpu().numpy(),
               g_dense={k: rt.g(k).cpu().numpy() for k in L.entries if k != 'items_embedding'})
    model.apply_gradients()
    torch.cuda.synchronize()
    wtab = rt.p('items_embedding')
    untouched = wtab[N_BIG - 1].cpu().numpy()                      # last row of the table: byte offset 7.56 GB - 1.5 KB
    res.update(w_rows=wtab[ridx].cpu().numpy(), m_rows=rt.view(rt.m, 'items_embedding')[ridx].cpu().numpy(),
               v_rows=rt.view(rt.v, 'items_embedding')[ridx].cpu().numpy(), untouched=untouched,
               # rows that differ from the common value of an untouched row after Adam: must be a subset of the populated rows
               w_changed=torch.nonzero((wtab != torch.from_numpy(untouched).to(rt.device)).any(1)).flatten().cpu().numpy(),
               w_dense={k: rt.p(k).cpu().numpy() for k in L.entries if k != 'items_embedding'}, e_offset=e.offset)
    del model, rt, gtab, wtab, ridx
    gc.collect()
    torch.cuda.empty_cache()
    return res


def test_item_rows_beyond_4gib_match_the_compact_id_run(gpu):
    lo = _run(0)
    hi = _run(OFFSET)
    assert int(hi['rows'][1]) * 378 * 4 > (1 << 32) and int(hi['rows'][1]) * ACE_DIM * 4 > (1 << 31)
    assert np.array_equal(_remap(lo['neg'], OFFSET), hi['neg']), "negatives differ after the id remap"
    for k in ('logits', 'probs'):
        assert np.array_equal(lo[k], hi[k]), "%s not bit-identical: max |d| %g" % (k, float(np.abs(lo[k] - hi[k]).max()))
    # loss = [total, cross-entropy, L2]: the cross-entropy is bit-identical; the L2 term sums the squares of ALL 1.9 G table entries - the
    # same multiset of values at other positions, i.e. in another order of fp32 partial sums (measured: 1.0303863 vs 1.0303853)
    assert lo['loss'][1] == hi['loss'][1], (lo['loss'], hi['loss'])
    assert abs(float(lo['loss'][2]) - float(hi['loss'][2])) <= 1e-5 * float(lo['loss'][2]) and float(lo['loss'][2]) > 0.5, (lo['loss'], hi['loss'])
    assert abs(float(lo['loss'][0]) - float(hi['loss'][0])) <= 1e-5 * float(lo['loss'][0])
    for k, v in lo['g_dense'].items():
        assert np.array_equal(v, hi['g_dense'][k]), "gradient of %s differs" % k
    assert lo['g_touched'].size > 100 and np.array_equal(_remap(lo['g_touched'], OFFSET), hi['g_touched']), \
        "rows of the item table with a gradient: %d (low ids) vs %d (high ids)" % (lo['g_touched'].size, hi['g_touched'].size)
    assert int(hi['g_touched'][hi['g_touched'] != 0].min()) > OFFSET and np.abs(lo['g_rows']).max() > 0      # (row 0 = the pad item, not moved)
    assert np.array_equal(lo['g_rows'], hi['g_rows']), "touched rows' gradients differ"
    # after TF-Adam (dense over every row): populated rows + both slots bit-identical, every other row holds ONE common value
    for k in ('w_rows', 'm_rows', 'v_rows', 'untouched'):
        assert np.array_equal(lo[k], hi[k]), k
    assert np.all(lo['untouched'] == lo['untouched'][0]) and lo['untouched'][0] != FILL      # L2 + Adam moved the untouched rows, all alike
    populated_lo, populated_hi = set(lo['rows'].tolist()), set(hi['rows'].tolist())
    assert set(lo['w_changed'].tolist()) <= populated_lo and set(hi['w_changed'].tolist()) <= populated_hi
    assert np.array_equal(_remap(lo['w_changed'], OFFSET)[lo['w_changed'] != 0], hi['w_changed'][hi['w_changed'] != 0])
    for k, v in lo['w_dense'].items():
        assert np.array_equal(v, hi['w_dense'][k]), "post-Adam %s differs" % k
