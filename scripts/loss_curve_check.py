"""The criteria of tests/test_g1shape_parity_gpu.py::test_loss_curve_g1_shape_and_hitrate evaluated OFFLINE: a curve that test dumped
(gpurun_out/loss_curve_200.json -> profiles/r05_loss_curve_200.json: |loss - f64| per step of the two HIP arms) against the committed fixture
tests/golden/loss_curve_200.npz (float64 trajectory + the fp32 realisations of the oracle).  python scripts/loss_curve_check.py [curve.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def evaluate(curve_path=None):
    fx = np.load(os.path.join(ROOT, "tests", "golden", "loss_curve_200.npz"))
    d = json.load(open(curve_path or os.path.join(ROOT, "profiles", "r05_loss_curve_200.json")))
    f64, f32 = fx['loss_f64'], fx['loss_f32']
    n = int(d['steps'])
    assert np.allclose(np.asarray(d['loss_f64']), f64[:n], rtol=0, atol=0), "the curve was measured against another float64 trajectory"
    dev32 = np.abs(f32 - f64[None])[:, :n]
    env = np.maximum.accumulate(dev32.max(0))
    held = lambda x: int(next((i for i, v in enumerate(x) if v >= 1e-3), len(x)))
    out = dict(fp32_arms=int(f32.shape[0]), oracle_held=[held(r) for r in dev32], oracle_worst=[float(r.max()) for r in dev32],
               oracle_mean=[float(r.mean()) for r in dev32], arms={})
    loo = 0.0
    for i in range(dev32.shape[0]):
        e = np.maximum.accumulate(np.delete(dev32, i, 0).max(0))
        loo = max(loo, float((dev32[i] / np.maximum(e, 1e-12))[10:].max()))
    out['oracle_leave_one_out_worst_ratio'] = loo
    for name, key in (("default", "abs_dev_default"), ("native", "abs_dev_native")):
        x = np.asarray(d[key])
        out['arms'][name] = dict(held_1e3=held(x), worst=float(x.max()), mean=float(x.mean()), dev_at_step_19=float(x[19]),
                                 max_fraction_of_bound=float((x / (4.0 * env + 1e-4)).max()),
                                 max_ratio_to_envelope_after_step_25=float((x / env)[25:].max()),
                                 violations=int((x > 4.0 * env + 1e-4).sum() + (x[:25] >= 1e-3).sum()),
                                 mean_ok=bool(x.mean() <= 2.0 * dev32.mean(1).max() + 1e-4))
    return out


if __name__ == "__main__":
    print(json.dumps(evaluate(sys.argv[1] if len(sys.argv) > 1 else None), indent=1))
