"""Scorer layer-1 dgrad fused with the cand (.) pred / CAR-tanh backward (csrc/dm_fused.hip) against a float64 reference of
   dM = dS1 Ws1^T,  dZ2 = dM * pred * (1 - Z2^2),  dpred = (sum_c dM * Z2) * (1 - pred^2),  b2part = sum_c dZ2   (per position)
and against the two kernels it replaces (cham_gemm_f32x3 + cham_mulpred_bwd_p3) on the same operands: 1 + N = 51 (G1), 101 (Adressa),
201 (config 5), the limits 32 and 256, positions that do not fill the last workgroup, several widths C; repeatability (race screen)."""
import pytest
import torch

from tests.test_gemm_p3_gpu import split3

pytestmark = pytest.mark.gpu


def _case(gpu, BT, N, C, seed=0, reps=1):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    NC = N + 1
    Rc = BT * NC
    g = torch.Generator(device=gpu).manual_seed(seed)
    dS1 = torch.randn(Rc, 128, device=gpu, generator=g)
    Ws1 = torch.randn(C, 128, device=gpu, generator=g) * 0.1
    Z2 = torch.tanh(torch.randn(Rc, C, device=gpu, generator=g))
    pred = torch.tanh(torch.randn(BT, C, device=gpu, generator=g))
    Wp = split3(Ws1)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(reps):
        planes = torch.full((3, Rc, C), float('nan'), dtype=torch.bfloat16, device=gpu)
        dpred = torch.full((BT, C), float('nan'), device=gpu)
        b2p = torch.full((BT, C), float('nan'), device=gpu)
        check(lib.cham_dm_mulpred_p3(ptr(dS1), 128, 128, ptr(Wp), C * 128, ptr(Z2), ptr(pred), C, BT, N, ptr(planes), Rc * C, ptr(dpred), ptr(b2p), st),
              "cham_dm_mulpred_p3")
        torch.cuda.synchronize()
        outs.append((planes, dpred, b2p))
    for o in outs[1:]:
        assert all(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
                   for a, b in zip(o, outs[0])), "fused kernel is not repeatable (race?)"
    planes, dpred, b2p = outs[0]
    dM = dS1.double() @ Ws1.double().t()
    pr = pred.double().repeat_interleave(NC, dim=0)
    o_ref = dM * pr * (1 - Z2.double() ** 2)
    dpred_ref = (dM * Z2.double()).view(BT, NC, C).sum(1) * (1 - pred.double() ** 2)
    b2_ref = o_ref.view(BT, NC, C).sum(1)
    got = planes[0].double() + planes[1].double() + planes[2].double()
    e_o = float((got - o_ref).abs().max()) / float(o_ref.abs().max())
    e_dp = float((dpred.double() - dpred_ref).abs().max()) / float(dpred_ref.abs().max())
    e_b2 = float((b2p.double() - b2_ref).abs().max()) / float(b2_ref.abs().max())
    # the pair it replaces, same operands
    dMf = torch.empty(Rc, C, device=gpu)
    check(lib.cham_gemm_f32x3(ptr(dS1), 128, 0, ptr(Ws1), 128, 1, ptr(dMf), C, Rc, C, 128, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), "x3")
    planes2 = torch.empty(3, Rc, C, dtype=torch.bfloat16, device=gpu)
    dpred2, b2p2 = torch.empty(BT, C, device=gpu), torch.empty(BT, C, device=gpu)
    check(lib.cham_mulpred_bwd_p3(ptr(dMf), ptr(Z2), ptr(pred), C, BT, N, ptr(dpred2), ptr(planes2), Rc * C, ptr(b2p2), st), "mulpred")
    torch.cuda.synchronize()
    got2 = planes2[0].double() + planes2[1].double() + planes2[2].double()
    e_o2 = float((got2 - o_ref).abs().max()) / float(o_ref.abs().max())
    e_dp2 = float((dpred2.double() - dpred_ref).abs().max()) / float(dpred_ref.abs().max())
    return (e_o, e_dp, e_b2), (e_o2, e_dp2)


@pytest.mark.parametrize("BT,N,C", [(40, 50, 1024), (7, 50, 256), (23, 100, 1024), (9, 200, 1024), (16, 31, 128), (5, 255, 64), (1, 50, 64), (33, 63, 512)])
def test_dm_fused_matches_float64_and_the_unfused_pair(gpu, BT, N, C):
    (e_o, e_dp, e_b2), (e_o2, e_dp2) = _case(gpu, BT, N, C)
    assert e_o < 2e-6 and e_dp < 5e-6 and e_b2 < 5e-6, (e_o, e_dp, e_b2)
    # fp32-grade: not measurably worse than the on-the-fly split GEMM + the separate elementwise kernel
    assert e_o < 1.5 * e_o2 + 2e-7 and e_dp < 1.5 * e_dp2 + 5e-7, (e_o, e_o2, e_dp, e_dp2)


def test_dm_fused_is_repeatable_at_the_g1_shape(gpu):
    (e_o, e_dp, e_b2), _ = _case(gpu, 256 * 19 // 4, 50, 1024, seed=3, reps=4)
    assert e_o < 2e-6 and e_dp < 5e-6 and e_b2 < 5e-6, (e_o, e_dp, e_b2)


def test_dm_fused_argument_errors(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import ptr
    lib = _lib.load()
    x = torch.zeros(64 * 51, 128, device=gpu); z = torch.zeros(64 * 51, 64, device=gpu); pr = torch.zeros(64, 64, device=gpu)
    w = torch.zeros(3, 64, 128, dtype=torch.bfloat16, device=gpu); o = torch.zeros(3, 64 * 51, 64, dtype=torch.bfloat16, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda K, C, N: lib.cham_dm_mulpred_p3(ptr(x), 128, K, ptr(w), 64 * 128, ptr(z), ptr(pr), C, 64, N, ptr(o), 64 * 51 * 64, ptr(pr), None, st)
    assert call(128, 64, 50) == 0
    assert call(64, 64, 50) < 0          # K != 128
    assert call(128, 96, 50) < 0         # C % 64
    assert call(128, 64, 10) < 0         # 1 + N < 32
    assert call(128, 64, 300) < 0        # 1 + N > 256


@pytest.mark.parametrize("BT,N,C", [(40, 50, 1024), (7, 50, 256), (23, 100, 1024), (9, 200, 1024), (16, 31, 128), (1, 50, 64)])
def test_dm_fused_bf16_twin_matches_the_unfused_pair(gpu, BT, N, C):
    """BASELINE configs[2] arithmetic: the all-bf16 form (cham_dm_mulpred_b16) against the two kernels it replaces on the same bf16 operands
    (cham_gemm_b16 NT with bf16 output + cham_mulpred_bwd_b16, then cham_colsum_b16 for b2): dM is rounded to bf16 exactly where the pair
    stores it, so the bf16 gradient matrix is BIT-identical; the per-position sums differ by fp32 summation order only."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    NC = N + 1
    Rc = BT * NC
    g = torch.Generator(device=gpu).manual_seed(BT + N)
    dS1 = (torch.randn(Rc, 128, device=gpu, generator=g)).bfloat16()
    Ws1 = (torch.randn(C, 128, device=gpu, generator=g) * 0.1).bfloat16()
    Z2 = torch.tanh(torch.randn(Rc, C, device=gpu, generator=g)).bfloat16()
    pred = torch.tanh(torch.randn(BT, C, device=gpu, generator=g))
    st = torch.cuda.current_stream().cuda_stream
    # unfused pair
    dZ = torch.empty(Rc, C, dtype=torch.bfloat16, device=gpu)
    check(lib.cham_gemm_b16(ptr(dS1), 128, 0, ptr(Ws1), 128, 1, ptr(dZ), C, 0, Rc, C, 128, None, 0, None, 0, 0, 0, None, 0, 1, st), "gemm_b16")
    dpred_ref = torch.empty(BT, C, device=gpu)
    check(lib.cham_mulpred_bwd_b16(ptr(dZ), ptr(Z2), ptr(pred), C, BT, N, ptr(dpred_ref), st), "mulpred_b16")
    torch.cuda.synchronize()
    b2_ref = dZ.float().view(BT, NC, C).double().sum(1)
    # fused
    outs = []
    for _ in range(2):
        o = torch.full((Rc, C), float('nan'), dtype=torch.bfloat16, device=gpu)
        dp = torch.full((BT, C), float('nan'), device=gpu); b2 = torch.full((BT, C), float('nan'), device=gpu)
        check(lib.cham_dm_mulpred_b16(ptr(dS1), 128, 128, ptr(Ws1), ptr(Z2), ptr(pred), C, BT, N, ptr(o), ptr(dp), ptr(b2), st), "dm_b16")
        torch.cuda.synchronize()
        outs.append((o, dp, b2))
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    o, dp, b2 = outs[0]
    same = (o.view(torch.int16) == dZ.view(torch.int16)).float().mean().item()
    # (accumulation order inside the K = 128 product differs between the two kernels: a sum that lands on the other side of a bf16
    # rounding boundary moves an element by one bf16 ulp - rare)
    assert same > 0.995, same
    assert float((o.float() - dZ.float()).abs().max()) <= 2.0 ** -7 * float(dZ.float().abs().max())
    assert float((dp.double() - dpred_ref.double()).abs().max()) < 2e-3 * float(dpred_ref.abs().max())
    assert float((b2.double() - b2_ref).abs().max()) < 2e-3 * float(b2_ref.abs().max()) + 1e-6


@pytest.mark.parametrize("BT,N,C,wide", [(40, 50, 1024, False), (23, 100, 1024, True), (9, 200, 1024, False), (16, 31, 128, True), (5, 255, 64, False),
                                         (256 * 19 // 4, 50, 1024, True)])
def test_dm_fused_two_fp16_plane_products_match_float64_and_the_six_product_form(gpu, BT, N, C, wide):
    """MODE 3 (cham_dm_mulpred_h2h, round 5): the kernel's own K = 128 products as THREE fp16-plane products (dS1 under the scale of its max row
    norm - cham_h2_scale_rownorm2's second record - Ws1's planes under its row-norm record) next to MODE 1 (six bf16 products) on the same
    operands and records: the two-plane outputs, dpred and the b2 partial sums against float64, not measurably worse than MODE 1; `wide`: the
    rows of dS1 spread over 20 binary orders of magnitude (one scale per matrix: the absolute-error contract)."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    NC = N + 1
    Rc = BT * NC
    g = torch.Generator(device=gpu).manual_seed(11)
    dS1 = torch.randn(Rc, 128, device=gpu, generator=g) * 1e-3
    if wide:
        dS1 = dS1 * torch.exp2(-20 * torch.rand(Rc, 1, device=gpu, generator=g))
    Ws1 = torch.randn(C, 128, device=gpu, generator=g) * 0.05
    Z2 = torch.tanh(torch.randn(Rc, C, device=gpu, generator=g))
    pred = torch.tanh(torch.randn(BT, C, device=gpu, generator=g))
    st = torch.cuda.current_stream().cuda_stream
    Wp = split3(Ws1)
    rw, rout, rds = (torch.zeros(8, device=gpu) for _ in range(3))
    check(lib.cham_h2_scale_rownorm(ptr(Ws1), C, 128, 128, None, ptr(rw), st), "rownorm")
    check(lib.cham_h2_scale_rownorm2(ptr(dS1), Rc, 128, 128, rw.data_ptr() + 8, ptr(rout), ptr(rds), st), "rownorm2")
    Wh = torch.zeros(2, C, 128, dtype=torch.float16, device=gpu)
    check(lib.cham_split2h(ptr(Ws1), C, 128, 128, ptr(Wh), C * 128, 128, None, 0, 0, ptr(rw), 0, st), "split2h")
    outs = []
    for mode in (1, 3, 3):
        D = torch.full((2, Rc, C), float('nan'), dtype=torch.float16, device=gpu)
        dp, b2 = torch.full((BT, C), float('nan'), device=gpu), torch.full((BT, C), float('nan'), device=gpu)
        if mode == 1:
            check(lib.cham_dm_mulpred_h2(ptr(dS1), 128, 128, ptr(Wp), C * 128, ptr(Z2), ptr(pred), C, BT, N, ptr(D), Rc * C, ptr(rout), ptr(dp), ptr(b2), st), "h2")
        else:
            check(lib.cham_dm_mulpred_h2h(ptr(dS1), 128, 128, ptr(Wh), C * 128, ptr(rds), ptr(rw), ptr(Z2), ptr(pred), C, BT, N, ptr(D), Rc * C, ptr(rout),
                                          ptr(dp), ptr(b2), st), "h2h")
        torch.cuda.synchronize()
        outs.append((D, dp, b2))
    assert all(torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a, b.view(torch.int16) if b.dtype == torch.float16 else b)
               for a, b in zip(outs[1], outs[2])), "not repeatable (race?)"
    dM = dS1.double() @ Ws1.double().t()
    pr = pred.double().repeat_interleave(NC, dim=0)
    o_ref = dM * pr * (1 - Z2.double() ** 2)
    dp_ref = (dM * Z2.double()).view(BT, NC, C).sum(1) * (1 - pred.double() ** 2)
    b2_ref = o_ref.view(BT, NC, C).sum(1)
    inv = float(rout[1])
    errs = []
    for D, dp, b2 in outs[:2]:
        assert torch.isfinite(D.float()).all() and float(D[0].float().abs().max()) <= 2.0 ** 15
        got = (D[0].double() + D[1].double()) * inv
        errs.append((float((got - o_ref).abs().max()) / float(o_ref.abs().max()), float((dp.double() - dp_ref).abs().max()) / float(dp_ref.abs().max()),
                     float((b2.double() - b2_ref).abs().max()) / float(b2_ref.abs().max())))
    (e1_o, e1_dp, e1_b2), (e3_o, e3_dp, e3_b2) = errs
    assert e3_o < 2e-6 and e3_dp < 5e-6 and e3_b2 < 5e-6, errs
    assert e3_o < 1.5 * e1_o + 3e-7 and e3_dp < 1.5 * e1_dp + 5e-7 and e3_b2 < 1.5 * e1_b2 + 5e-7, errs
    # the sign of the h plane is the sign of the value (the CAR dgrad reads leaky' from a plane like it; here: sanity of the plane format)
    assert torch.equal(outs[1][0][0].view(torch.int16) > 0, ((outs[1][0][0].double() + outs[1][0][1].double()) > 0))
    assert lib.cham_dm_mulpred_h2h(ptr(dS1), 128, 128, ptr(Wh), C * 128, None, ptr(rw), ptr(Z2), ptr(pred), C, BT, N, ptr(outs[0][0]), Rc * C, ptr(rout),
                                   ptr(outs[0][1]), ptr(outs[0][2]), st) < 0
