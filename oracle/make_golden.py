#!/usr/bin/env python
"""Generates the committed golden fixtures under tests/golden/ by EXECUTING the reference's own importable
modules in the build container (they cannot travel to the GPU box: /root/reference does not exist there).

TEST INFRASTRUCTURE (see oracle/__init__.py).

  python oracle/make_golden.py            # rewrites tests/golden/*.npz

What the reference can run here (TensorFlow 1.12 is not installable, so nothing inside the TF graph):
  * nar_module/nar/clicked_items_state.py  ClickedItemsState  -> state_trace.npz
  * nar_module/nar/metrics.py              HitRate, MRR        -> metrics_hitrate_mrr.npz
  * metrics.HitRateBySessionPosition, evaluation.ColdStartAnalysisState, the state's first-click bookkeeping -> eval_extras.npz
  * nar_module/nar/benchmarks/candidate_sampling.py (numpy clone of the graph's negative sampler; loaded by file
    path because the package __init__ pulls TF)              -> sampler_clone_stats.npz (distribution pin)
The restated oracle's own outputs for one full tiny training step (weights, inputs, negatives, logits, loss,
gradients, post-Adam weights) are stored in nar_step_tiny.npz: they pin the HIP path AND guard the oracle
against accidental edits ("parity unpinned at the TF boundary" still applies to those numbers).
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CHAMELEON_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _ref_nar():
    sys.path.insert(0, os.path.join(REF, "nar_module"))
    from nar import clicked_items_state, metrics          # noqa: E402  (reference modules, executed unmodified)
    return clicked_items_state, metrics


def _ref_sampler_clone():
    path = os.path.join(REF, "nar_module", "nar", "benchmarks", "candidate_sampling.py")
    spec = importlib.util.spec_from_file_location("ref_candidate_sampling", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def state_trace():
    """12 batches through the reference ClickedItemsState (two buffer geometries: one that overflows max_size, one
    where the time-window truncation bites)."""
    from chameleon_recsys_amd.nar import synthetic
    from oracle.state import batch_clicks_for_state
    cis, _ = _ref_nar()
    out = {}
    for tag, (hours, max_size, for_norm, B, per_hour) in {"a": (1.0, 600, 200, 48, 96), "b": (0.5, 5000, 300, 32, 32)}.items():
        p = synthetic.default_params(1000, 16, seq_len=8, batch_size=B)
        batches = synthetic.make_batches(12, B, 8, 1000, p['session_features_config'], seed=5, length_dist='g1',
                                         sessions_per_hour=per_hour)
        st = cis.ClickedItemsState(hours, max_size, for_norm, 1000)
        out[tag + "_cfg"] = np.array([hours, max_size, for_norm, 1000], dtype=np.float64)
        for i, (f, l) in enumerate(batches):
            ids, ts = batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp'])
            st.update_items_state(ids, ts)
            out["%s_ids_%d" % (tag, i)] = ids
            out["%s_ts_%d" % (tag, i)] = ts
            out["%s_buffer_%d" % (tag, i)] = st.pop_recent_clicks_buffer.copy()
            out["%s_popnorm_%d" % (tag, i)] = st.get_articles_recent_pop_norm().copy()
            out["%s_pop_%d" % (tag, i)] = st.get_articles_pop().copy()
            if i == 5:     # snapshot / restore around an "evaluation" (clicked_items_state.py:49-79)
                st.save_state_checkpoint()
                st.update_items_state(ids[::-1].copy(), ts[::-1].copy())
                st.restore_state_checkpoint()
                out["%s_buffer_restored" % tag] = st.pop_recent_clicks_buffer.copy()
    np.savez_compressed(os.path.join(GOLD, "state_trace.npz"), **out)


def metrics_fixture():
    _, metrics = _ref_nar()
    rng = np.random.default_rng(11)
    B, T, K = 24, 6, 12
    labels = rng.integers(1, 40, size=(B, T)).astype(np.int64)
    labels[rng.random((B, T)) < 0.3] = 0
    preds = np.stack([[rng.permutation(40)[:K] + 1 for _ in range(T)] for _ in range(B)]).astype(np.int64)
    out = dict(labels=labels, preds=preds)
    for n in (1, 5, 10):
        hr, mrr = metrics.HitRate(n), metrics.MRR(n)
        hr.add(preds[:10], labels[:10]); hr.add(preds[10:], labels[10:])       # streaming: two adds
        mrr.add(preds[:10], labels[:10]); mrr.add(preds[10:], labels[10:])
        out["hitrate_at_%d" % n] = np.float64(hr.result())
        out["mrr_at_%d" % n] = np.float64(mrr.result())
    np.savez_compressed(os.path.join(GOLD, "metrics_hitrate_mrr.npz"), **out)


def eval_extras_fixture():
    """HitRateBySessionPosition (metrics.py:136-168), ColdStartAnalysisState (evaluation.py:50-90) and the state's first-click
    bookkeeping (clicked_items_state.py:97-104, 196-203), executed from the reference on random streams."""
    cis, metrics = _ref_nar()
    from nar import evaluation as ref_eval                       # noqa: E402  (reference module)
    rng = np.random.default_rng(23)
    B, T, K, n_items, steps = 16, 7, 9, 60, 6
    out = dict(cfg=np.array([B, T, K, n_items, steps]))
    hr = metrics.HitRateBySessionPosition(5)
    st = cis.ClickedItemsState(1.0, 100, 50, n_items)
    for i in range(steps):
        labels = rng.integers(1, n_items, size=(B, T)).astype(np.int64)
        lens = rng.integers(1, T + 1, size=B)
        labels[np.arange(T)[None, :] >= lens[:, None]] = 0
        clicked = rng.integers(1, n_items, size=(B, T)).astype(np.int64)
        clicked[labels == 0] = 0
        preds = np.stack([[rng.permutation(n_items - 1)[:K] + 1 for _ in range(T)] for _ in range(B)]).astype(np.int64)
        pop = rng.random((B, T))
        hr.add(preds, labels, pop)
        # ItemsStateUpdaterHook.update_items_cold_start_state, nar_model.py:1480-1494
        nz = set(clicked.reshape(-1)).union(set(labels.reshape(-1))).difference(set([0]))
        st.increment_current_step()
        st.update_items_first_click_step(nz)
        st.get_cold_start_state().update_items_num_steps_before_first_rec(preds[:, :, :5], st.items_first_click_step, st.get_current_step())
        out.update({"labels_%d" % i: labels, "clicked_%d" % i: clicked, "preds_%d" % i: preds, "pop_%d" % i: pop})
    hitrate, avg_pop, total = hr.result()
    keys = sorted(total.keys())
    out.update(pos_keys=np.array(keys), pos_hitrate=np.array([hitrate[k] for k in keys]), pos_avg_pop=np.array([avg_pop[k] for k in keys]),
               pos_total=np.array([total[k] for k in keys]))
    res = ref_eval.compute_metrics_results([hr], recommender='chameleon')
    out.update(result_keys=np.array(sorted(res.keys())))
    fc = st.items_first_click_step
    ids = sorted(fc.keys())
    out.update(first_click_ids=np.array(ids), first_click_step=np.array([fc[k] for k in ids]))
    nb = st.get_cold_start_state().items_num_steps_before_first_rec
    ids = sorted(nb.keys())
    out.update(first_rec_ids=np.array(ids), first_rec_steps=np.array([nb[k] for k in ids]))
    stats = st.get_cold_start_state().get_statistics()
    skeys = sorted(stats.keys())
    out.update(stats_keys=np.array(skeys), stats_values=np.array([float(stats[k]) for k in skeys]))
    np.savez_compressed(os.path.join(GOLD, "eval_extras.npz"), **out)


def sampler_clone_stats():
    """Distribution pin against the reference's numpy clone of the sampler (its RNG is un-seedable
    np.random.permutation inside -> we store first-pick frequencies over many draws with np.random.seed)."""
    mod = _ref_sampler_clone()
    np.random.seed(1234)
    cand = np.array([7] * 6 + [8] * 3 + [9] * 1, dtype=np.int64)
    mgr = mod.CandidateSamplingManager(lambda: np.zeros(1, np.int64))
    first = [int(mgr.get_neg_items_click(cand, 1)[0]) for _ in range(20000)]
    freq = np.bincount(first, minlength=10)[7:] / len(first)
    np.savez_compressed(os.path.join(GOLD, "sampler_clone_stats.npz"), cand=cand, first_pick_freq=freq)


def nar_step_tiny():
    import torch
    from chameleon_recsys_amd.nar import synthetic
    from oracle.nar_oracle import NAROracle
    from tests import helpers as H
    torch.set_num_threads(1)
    p = H.tiny_params(C=64, H=40, neg=6, batch_size=12, buffer_size=400, for_norm=60, neg_from_buffer=40, n_items=300,
                      ace_dim=16)
    batches = synthetic.make_batches(4, 12, 8, 300, p['session_features_config'], seed=9, length_dist='g1')
    st = H.warm_state(p, batches[:3])
    orc = NAROracle(p, seed=17)
    rng = np.random.default_rng(17)
    with torch.no_grad():
        for k, v in orc.w.items():
            if k.endswith('bias') or k == 'beta':
                v.copy_(torch.from_numpy((0.05 * rng.standard_normal(tuple(v.shape))).astype(np.float32)))
    w0 = orc.weights_numpy()
    f, l = batches[3]
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    ref = orc.train_step(f, l, buf, pop, return_grads=True)
    out = {"w0/" + k: v for k, v in w0.items()}
    out.update({"w1/" + k: v for k, v in orc.weights_numpy().items()})
    out.update({"g/" + k: v.numpy() for k, v in ref['grads'].items()})
    out.update({"f/" + k: np.asarray(v) for k, v in f.items()})
    out.update({"l/" + k: np.asarray(v) for k, v in l.items()})
    out.update(buffer=buf, pop_norm=pop, neg_items=ref['neg_items'].numpy(), logits=ref['logits'].numpy(),
               probs=ref['probs'].numpy(), mask=ref['mask'].numpy(),
               loss=np.array([float(ref['total_loss']), float(ref['xe_loss']), float(ref['reg_loss'])]))
    np.savez_compressed(os.path.join(GOLD, "nar_step_tiny.npz"), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    state_trace()
    metrics_fixture()
    eval_extras_fixture()
    sampler_clone_stats()
    nar_step_tiny()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
