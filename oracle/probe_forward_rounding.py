#!/usr/bin/env python
"""Why the scorer's FORWARD layer-1 GEMM keeps exact operands (profiles/r05_notes.md section 9) - the question asked of the CPU oracle alone, no
HIP kernel involved: the fp32 oracle with the two operands of matching_dense_layer_1 (reference nar_model.py:447-451, 478-495) rounded the way
cham_gemm_f32x2h rounds them - h = fp16(x s), l = fp16(x s - h), x' = (h + l) / s; s = 2^14 for cand (.) pred, the max-row-norm scale for the
weight - in the FORWARD only (straight-through: gradients flow as if x' = x), trained for the first steps of the 200-step loss-curve setup and
compared with the float64 trajectory of tests/golden/loss_curve_200.npz next to the twelve unrounded fp32 realisations.  Result (round 5): 1.1-1.4 x
the drift of sixteen unrounded arms for the scorer's layer (twelve arms; nothing measurable with only one of its two operands rounded, four arms
each), 0.8-1.0 x for the CAR layer-2 matmul (four arms).

TEST INFRASTRUCTURE (see oracle/__init__.py); runs in the build container:
  python oracle/probe_forward_rounding.py [steps=45] [perm seeds ...]   -> gpurun_out/forward_rounding_probe.json
  PROBE_WHERE=car (the same question asked of the CAR layer-2 matmul's operands, which the DEFAULT arithmetic does round), PROBE_CONTROLS=0
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def two_plane(x, scale):
    """x -> (fp16(x s) + fp16(x s - fp16(x s))) / s with a straight-through gradient."""
    xs = x.detach() * scale
    h = xs.half()
    l = (xs - h.float()).half()
    q = (h.float() + l.float()) / scale          # 11 + 11 bits: exact in fp32
    return x + (q - x.detach())


def pow2_scale(bound):
    return 2.0 ** (15 - math.frexp(bound)[1]) if bound > 0 else 1.0


def run(steps, perm, rounded, where="scorer"):
    """where = "scorer": the operands of matching_dense_layer_1; "car": those of the CAR layer-2 matmul over the candidate rows (what
    cham_gemm_h2 rounds in the DEFAULT arithmetic: the PreCAR output under the bound max|U| + max|V| <= 2 max|Z1|, W2 under max|W2|)."""
    from oracle.nar_oracle import NAROracle
    from tests import helpers as H

    class Rounded(NAROracle):
        def _car(self, x, candidate_rows=False):
            if not (rounded and where == "car" and candidate_rows):
                return super()._car(x, candidate_rows)
            w = self.w
            pre = self._leaky_site('Z1', self._mm(x, w['PreCAR/kernel']) + w['PreCAR/bias'])
            self._tap('Z1', pre)
            k = w['CAR/kernel']
            sp, sk = pow2_scale(2.0 * float(pre.detach().abs().max())), pow2_scale(float(k.detach().abs().max()))
            return self._store(torch.tanh(self._mm(two_plane(pre, sp), two_plane(k, sk)) + w['CAR/bias']))

        def _scorer(self, m):
            if not (rounded and where.startswith("scorer")):
                return super()._scorer(m)
            w = self.w
            k = w['match1/kernel']
            bound = float(k.detach().double().pow(2).sum(1).sqrt().max()) * 1.0009765625          # k_h2_scale_rownorm
            sw = 2.0 ** (15 - math.frexp(bound)[1])
            ma = m if where == "scorer_w" else two_plane(m, 2.0 ** 14)          # scorer_a / scorer_w: only one of the two operands rounded
            kw = k if where == "scorer_a" else two_plane(k, sw)
            s1 = self._leaky_site('S1', self._mm(ma, kw) + w['match1/bias'])
            s2 = self._leaky_site('S2', self._mm(s1, w['match2/kernel']) + w['match2/bias'])
            s3 = self._store(self._leaky_site('S3', self._mm(s2, w['match3/kernel']) + w['match3/bias']))
            self._tap('S1', s1); self._tap('S2', s2); self._tap('S3', s3)
            return self._mmf(s3, w['match4/kernel']) + w['match4/bias']

    p, batches, st, w = H.loss_curve_setup()
    orc = Rounded(p, weights=w, sum_perm_seed=perm)
    out = []
    for i, (f, l) in enumerate(batches[2:2 + steps]):
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        out.append(float(orc.train_step(f, l, buf, pop)['total_loss'].double()))
        H.update_state(st, f, l)
    return out


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "4")))
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    where = os.environ.get("PROBE_WHERE", "scorer")          # scorer | car | scorer_a | scorer_w
    controls = os.environ.get("PROBE_CONTROLS", "1") == "1"
    perms = [None if a == "none" else int(a) for a in sys.argv[2:]] or [None, 1, 2]
    fx = np.load(os.path.join(ROOT, "tests", "golden", "loss_curve_200.npz"))
    f64 = fx['loss_f64'][:steps]
    res = dict(steps=steps, arms={})
    for perm in perms:
        for rounded in ((True, False) if controls else (True,)):
            label = {"scorer": "forward operands on two fp16 planes", "car": "CAR layer-2 forward operands on two fp16 planes",
                     "scorer_a": "only cand (.) pred on two fp16 planes", "scorer_w": "only Ws1 on two fp16 planes"}[where]
            name = "%s, sum_perm_seed=%s" % (label if rounded else "exact operands (control)", perm)
            dev = np.abs(np.asarray(run(steps, perm, rounded, where)) - f64)
            res['arms'][name] = [float(x) for x in dev]
            print("%-70s |loss - f64| at steps 10 15 19 22 25 30 40: %s" % (name, " ".join("%.1e" % dev[s] for s in (10, 15, 19, 22, 25, 30, 40) if s < steps)), flush=True)
    res['fixture_fp32_arms_max_dev'] = [float(x) for x in np.abs(fx['loss_f32'] - fx['loss_f64'][None])[:, :steps].max(0)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "forward_rounding_probe%s.json" % ("" if where == "scorer" else "_" + where)), "w") as fh:
        json.dump(res, fh)
