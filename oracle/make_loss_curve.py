#!/usr/bin/env python
"""Generates tests/golden/loss_curve_200.npz: the 200-step G1-width training trajectory of the CPU oracle in FLOAT64 and the
deviation of several equally correct FLOAT32 realisations from it (nar_trainer_gcom.py:511-525 train loop, nar_model.py:708-722
optimizer; north_star: "loss curve matching CPU reference within 1e-3").

TEST INFRASTRUCTURE (see oracle/__init__.py).  Runs in the build container (CPU, ~1 h on 8 cores); the GPU test
tests/test_g1shape_parity_gpu.py::test_loss_curve_g1_shape_and_hitrate then runs only the HIP arms and compares them with the
committed float64 losses, with the fp32 realisations' own spread as the yardstick - the 480 s of CPU oracle the test used to spend
on the GPU box are gone.

  python oracle/make_loss_curve.py run <arm> [steps] [family]    # one trajectory -> gpurun_out/loss_curve_arm_[<family>_]<arm>.json
  python oracle/make_loss_curve.py merge [family]                # all arms of a family -> tests/golden/loss_curve_200[_<family>].npz

Families (tests/helpers.py LOSS_CURVE_FAMILIES; round 6): A = the round-5 trajectory (thirteen arms), B / C = other initial weights and
another batch stream, float64 + four fp32 realisations each - the envelope factor of the GPU test is fitted on held-out families.

Arms: f64 (float64 everything, the graph's own float32 quantisations kept), f32 (the oracle as every parity test uses it),
f32_p1..p11 (fp32 with every contraction summed in a permuted order: NAROracle(sum_perm_seed=k)) - twelve fp32 realisations: four were
too small a sample of a chaotic divergence (the level two of them reach at step 22 a HIP arm reached at step 19).
Inputs / weights: tests/helpers.py loss_curve_setup() (seeded, host only) - exactly what the GPU test feeds the HIP path.
Per step: total / cross-entropy / regularisation loss; SHA-1 of the drawn negatives (integer path: identical in every arm);
after the last step: top-5 ranked ids of four held-out batches in EVAL mode (HitRate@5 / MRR@5 of the trained weights).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
GOLD = os.path.join(ROOT, "tests", "golden", "loss_curve_200.npz")
ARMS = {"f64": (torch.float64, None), "f32": (torch.float32, None)}
ARMS.update({"f32_p%d" % k: (torch.float32, k) for k in range(1, 12)})          # eleven permuted-summation realisations (p4..p11 added late in round 5)


FAMILY_ARMS = ("f64", "f32", "f32_p1", "f32_p2", "f32_p3")         # families B, C: float64 + four fp32 realisations (A: all thirteen)


def _arm_path(arm, family):
    return os.path.join(OUT, "loss_curve_arm_%s.json" % arm if family == "A" else "loss_curve_arm_%s_%s.json" % (family, arm))


def run(arm, steps=None, family="A"):
    from oracle.nar_oracle import NAROracle
    from tests import helpers as H
    from chameleon_recsys_amd.nar import metrics
    from chameleon_recsys_amd.nar.nar_model import NARModuleModel
    dtype, perm = ARMS[arm]
    p, batches, st, w = H.loss_curve_setup(family=family)
    steps = H.LOSS_CURVE['steps'] if steps is None else steps
    orc = NAROracle(p, weights=w, dtype=dtype, sum_perm_seed=perm)
    rec = dict(arm=arm, family=family, total=[], xe=[], reg=[], neg_sha1=[], T=[], n_valid=[], seconds=[])
    path = _arm_path(arm, family)
    for i, (f, l) in enumerate(batches[2:2 + steps]):
        t0 = time.time()
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        ref = orc.train_step(f, l, buf, pop)
        rec['total'].append(float(ref['total_loss'].double())); rec['xe'].append(float(ref['xe_loss'].double()))
        rec['reg'].append(float(ref['reg_loss'].double()))
        rec['neg_sha1'].append(hashlib.sha1(np.ascontiguousarray(ref['neg_items'].numpy().astype(np.int64)).tobytes()).hexdigest())
        rec['T'].append(int(f['item_clicked'].shape[1])); rec['n_valid'].append(int(ref['mask'].sum()))
        rec['seconds'].append(time.time() - t0)
        H.update_state(st, f, l)
        if i % 5 == 4 or i == steps - 1:
            print("%s step %d loss %.6f (%.1f s/step)" % (arm, i, rec['total'][-1], float(np.mean(rec['seconds'][-5:]))), flush=True)
            with open(path + ".tmp", "w") as fh:
                json.dump(rec, fh)
            os.replace(path + ".tmp", path)
    if steps == H.LOSS_CURVE['steps']:
        hr, mrr, top5 = metrics.HitRate(5), metrics.MRR(5), []
        for j, (f, l) in enumerate(batches[2 + steps:]):
            buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
            key = NARModuleModel.eval_step_key(orc.global_step, j)
            pred = orc.forward(f, l, buf, pop, mode='eval', step=key)['predicted_item_ids'].numpy()
            hr.add(pred, l['label_next_item']); mrr.add(pred, l['label_next_item'])
            top5.append(pred[:, :, :5].astype(np.int64).tolist())
            H.update_state(st, f, l)
        rec['hitrate5'], rec['mrr5'], rec['eval_top5'] = float(hr.result()), float(mrr.result()), top5
        with open(path, "w") as fh:
            json.dump(rec, fh)
        print("%s HitRate@5 %.5f MRR@5 %.5f" % (arm, rec['hitrate5'], rec['mrr5']))


def merge(family="A"):
    from tests import helpers as H
    recs = {}
    GOLD = H.loss_curve_fixture_path(family)
    for arm in (ARMS if family == "A" else FAMILY_ARMS):
        with open(_arm_path(arm, family)) as fh:
            recs[arm] = json.load(fh)
        assert len(recs[arm]['total']) == H.LOSS_CURVE['steps'] and 'hitrate5' in recs[arm], arm
    sha = recs['f64']['neg_sha1']
    for arm, r in recs.items():
        assert r['neg_sha1'] == sha, "integer path differs between arms: %s" % arm
    f64 = np.asarray(recs['f64']['total'], np.float64)
    out = dict(config=np.array(json.dumps(dict(H.LOSS_CURVE, **H.LOSS_CURVE_FAMILIES[family]))), loss_f64=f64, xe_f64=np.asarray(recs['f64']['xe'], np.float64),
               reg_f64=np.asarray(recs['f64']['reg'], np.float64), neg_sha1=np.array(sha), T=np.asarray(recs['f64']['T'], np.int32),
               n_valid=np.asarray(recs['f64']['n_valid'], np.int32),
               eval_top5_f64=np.concatenate([np.asarray(t, np.int32).reshape(-1, 5) for t in recs['f64']['eval_top5']]),
               eval_T=np.asarray([np.asarray(t).shape[1] for t in recs['f64']['eval_top5']], np.int32),
               hitrate5_f64=np.float64(recs['f64']['hitrate5']), mrr5_f64=np.float64(recs['f64']['mrr5']))
    names = [a for a in recs if a != 'f64']
    out['f32_arms'] = np.array(names)
    out['loss_f32'] = np.stack([np.asarray(recs[a]['total'], np.float64) for a in names])
    out['hitrate5_f32'] = np.asarray([recs[a]['hitrate5'] for a in names]); out['mrr5_f32'] = np.asarray([recs[a]['mrr5'] for a in names])
    np.savez_compressed(GOLD, **out)
    dev = np.abs(out['loss_f32'] - f64[None])
    held = [int(next((i for i, d in enumerate(r) if d >= 1e-3), len(r))) for r in dev]
    print("wrote %s (%d bytes)" % (GOLD, os.path.getsize(GOLD)))
    for a, r, h in zip(names, dev, held):
        print("%-7s |loss - f64|: max %.2e mean %.2e, 1e-3 held for %d steps" % (a, r.max(), r.mean(), h))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1] == "run":
        torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "4")))
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else None, family=sys.argv[4] if len(sys.argv) > 4 else "A")
    else:
        merge(sys.argv[2] if len(sys.argv) > 2 else "A")
