"""GPU probe (manual tool, not a test): WHY does the scorer's layer-1 FORWARD matmul on two fp16 planes (CHAM_S1_H2=a) leave the float64 loss
curve faster than every other fp32 arm on two of three trajectory families (profiles/r06_notes.md section 4), although its per-launch error
against float64 is smaller than the six-product kernel's?

Arms, each a free-running 200-step HIP trajectory per family against the committed float64 curve (tests/golden/loss_curve_200*.npz):
  default   forward on six bf16 products, exact 24-bit operands (the shipped configuration)
  h2        forward on the two-plane kernel (cham_gemm_f32x2h: operands split while staged, three fp16 products, the a_l b_l term dropped)
  round     the SAME operand rounding - cand (.) pred and Ws1 rounded to h + l under the same scales - but EXACT products: the rounded
            operands are materialised (torch, in this script only) and multiplied by the six-product kernel
  round_a / round_b   only one of the two operands rounded
If `round` drifts like `h2`, the deterministic 22-bit operand rounding is the cause; if it drifts like `default`, the kernel's summation is.
usage: python scripts/probe_s1_forward_gpu.py [steps]   ->  gpurun_out/s1_forward_probe.json + a table on stdout"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tests import helpers as H


def round2h(x, scale):
    """x -> (h + l) / scale with h = fp16(x s), l = fp16(x s - h): csrc/common.h split2h (values, not bit patterns)."""
    xs = x * scale
    h = xs.half()
    l = (xs - h.float()).half()
    return (h.float() + l.float()) / scale


def wrap_forward(model, which):
    """Intercepts the scorer's layer-1 forward GEMM of `model` (rowscale = pred, N = 128, leaky epilogue) and feeds it rounded operands."""
    rt = model.rt
    orig = rt.gemm

    def gemm(A, B, C, M, N, K, lda, ldb, ldc, *a, **kw):
        rs = kw.get('rowscale')
        if rs is None or N != 128 or kw.get('transA', 0) or kw.get('act') != 1:
            return orig(A, B, C, M, N, K, lda, ldb, ldc, *a, **kw)
        nc = kw['rs_div']
        rows = torch.arange(M, device=A.device) // nc
        prod = A[:M, :K] * rs[rows, :K]                       # cand (.) pred in fp32, as the kernel's staging computes it
        Bm = B[:K, :N]
        if which in ("round", "round_a"):
            prod = round2h(prod, float(rt.sc_unit[0]))
        if which in ("round", "round_b"):
            Bm = round2h(Bm, float(rt.sc_ws1n[0]))
        kw2 = dict(kw, rowscale=None, ldrs=0, rs_div=1, h2scales=None)
        return orig(prod.contiguous(), Bm.contiguous(), C, M, N, K, K, N, ldc, *a, **kw2)
    rt.gemm = gemm


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = {}
    for family in "ABC":
        fx = np.load(H.loss_curve_fixture_path(family))
        f64, f32 = fx['loss_f64'], fx['loss_f32'][:4]
        env = np.maximum.accumulate(np.abs(f32 - f64[None]).max(0))
        cfg = dict(H.LOSS_CURVE, **H.LOSS_CURVE_FAMILIES[family])
        p, batches, st, w = H.loss_curve_setup(family=family)
        arms = {}
        for name in ("default", "h2", "round", "round_a", "round_b"):
            if name == "h2":
                os.environ["CHAM_S1_H2"] = "a"
            m, _ = H.make_pair(p, seed=cfg['weight_seed'])
            os.environ.pop("CHAM_S1_H2", None)
            if name.startswith("round"):
                wrap_forward(m, name)
            arms[name] = m
        dev = {k: [] for k in arms}
        for i, (f, l) in enumerate(batches[2:2 + steps]):
            buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
            for k, m in arms.items():
                m.feed_state(pop, buf)
                loss = m.train_step(m.upload_batch(f, l)).cpu().numpy()
                dev[k].append(abs(float(loss[0]) - float(f64[i])))
            H.update_state(st, f, l)
        held = lambda x: int(next((i for i, v in enumerate(x) if v >= 1e-3), len(x)))
        res = {}
        for k, x in dev.items():
            x = np.asarray(x)
            res[k] = dict(held_1e3=held(x), worst=float(x.max()), mean=float(x.mean()),
                          ratio_to_envelope_after_step_10=float(((x - 1e-4) / np.maximum(env[:steps], 1e-12))[10:].max()),
                          geo_mean_steps_10_29=float(np.exp(np.log(np.maximum(x[10:30], 1e-12)).mean())))
        out[family] = dict(oracle_held=[held(np.abs(r - f64)[:steps]) for r in f32], arms=res)
        print("family %s (oracle fp32 arms hold 1e-3 for %s steps)" % (family, out[family]['oracle_held']))
        for k, r in res.items():
            print("   %-8s 1e-3 held %3d steps, worst %.2e, mean %.2e, worst ratio to the envelope after step 10 %5.2f, geo-mean |dev| steps 10-29 %.2e"
                  % (k, r['held_1e3'], r['worst'], r['mean'], r['ratio_to_envelope_after_step_10'], r['geo_mean_steps_10_29']), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "s1_forward_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
