"""The criteria of tests/test_g1shape_parity_gpu.py::test_loss_curve_g1_shape_and_hitrate evaluated OFFLINE: the curves that test dumped
(gpurun_out/loss_curve_200_<family>.json -> profiles/r06_loss_curve_200_<family>.json: |loss - f64| per step of the HIP arms) against the
committed fixtures tests/golden/loss_curve_200[_B|_C].npz (float64 trajectory + the fp32 realisations of the oracle), with the envelope
factor fitted on the OTHER families (tests/helpers.py).  python scripts/loss_curve_check.py [family [curve.json]]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def evaluate(family="A", curve_path=None):
    from tests import helpers as H
    fx = np.load(H.loss_curve_fixture_path(family))
    d = json.load(open(curve_path or os.path.join(ROOT, "profiles", "r06_loss_curve_200_%s.json" % family)))
    f64, f32 = fx['loss_f64'], fx['loss_f32'][:4]
    n = int(d['steps'])
    assert np.allclose(np.asarray(d['loss_f64']), f64[:n], rtol=0, atol=0), "the curve was measured against another float64 trajectory"
    K, n_plain = H.loss_curve_envelope_factor(family), H.loss_curve_plain_steps(family)
    dev32 = np.abs(f32 - f64[None])[:, :n]
    env = np.maximum.accumulate(dev32.max(0))
    held = lambda x: int(next((i for i, v in enumerate(x) if v >= 1e-3), len(x)))
    out = dict(family=family, envelope_factor=K, plain_steps=n_plain, fp32_arms=int(f32.shape[0]), oracle_held=[held(r) for r in dev32],
               oracle_worst=[float(r.max()) for r in dev32], oracle_mean=[float(r.mean()) for r in dev32], arms={})
    for name, key in (("default", "abs_dev_default"), ("native", "abs_dev_native"), ("s1_forward_h2", "abs_dev_s1_forward_h2")):
        if key not in d:
            continue
        x = np.asarray(d[key])
        out['arms'][name] = dict(held_1e3=held(x), worst=float(x.max()), mean=float(x.mean()),
                                 max_fraction_of_bound=float((x / (K * env + 1e-4)).max()),
                                 max_ratio_to_envelope_after_step_10=float(((x - 1e-4) / np.maximum(env, 1e-12))[10:].max()),
                                 violations=int((x > K * env + 1e-4).sum() + (x[:n_plain] >= 1e-3).sum()),
                                 mean_ok=bool(x.mean() <= 2.0 * dev32.mean(1).max() + 1e-4))
    return out


if __name__ == "__main__":
    fams = [sys.argv[1]] if len(sys.argv) > 1 else ["A", "B", "C"]
    for f in fams:
        print(json.dumps(evaluate(f, sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
