"""Two-fp16-plane GEMM (csrc/gemm_h2.hip: three plane products per fp32 product, LDS-DMA staging, 256x256 tile, four-stage ring) against a
float64 reference: the device-side scales, both layouts, every epilogue, ragged rows / columns / K tails (descriptor range checks), one-
to four-step reductions (pipeline prologue), split-K placements, repeatability (race screen), operands with wide dynamic range and fp16
subnormal low planes, and - on the step's own shapes - its error next to the native fp32 MFMA's on the same operands: the bar the
two-plane arithmetic must clear to be the default (<= 1.5 x native, as tests/test_gemm_p3_gpu.py demands of the six-product bf16 split)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = 248064          # B * T * (1 + N) at the G1 shape


def _lib_():
    from chameleon_recsys_amd import _lib
    return _lib.load()


def new_rec(gpu):
    return torch.zeros(8, dtype=torch.float32, device=gpu)


def split2h_dev(x, rec=None, transposed=False):
    """fp32 [R, C] -> ([2, R, C] fp16 planes (and [2, C, R] of the transpose), record) through cham_split2h (scale from max |x|)."""
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib_()
    R, C = x.shape
    rec = new_rec(x.device) if rec is None else rec
    P = torch.zeros(2, R, C, dtype=torch.float16, device=x.device)
    PT = torch.zeros(2, C, R, dtype=torch.float16, device=x.device) if transposed else None
    check(lib.cham_split2h(ptr(x), R, C, C, ptr(P), R * C, C, ptr(PT), C * R, R, ptr(rec), 1, torch.cuda.current_stream().cuda_stream), "cham_split2h")
    return (P, PT, rec) if transposed else (P, rec)


def split2h_ref(x, scale):
    xs = x * scale
    h = xs.to(torch.float16)
    l = (xs - h.float()).to(torch.float16)
    hb = h.view(torch.int16).clone()
    hb[(x > 0) & (hb == 0)] = 1
    return torch.stack([hb.view(torch.float16), l])


def _counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_h2_launch_counts(out, 1 if reset else 0)
    return list(out)


def _gemm(lib, Ap, a_ps, lda, ra, Bp, b_ps, ldb, rb, tn, C, ldc, M, N, K, bias=None, act=0, dref=None, ldr=0, dact=0, accumulate=0, ws=None, splits=1):
    from chameleon_recsys_amd._lib import check, ptr
    check(lib.cham_gemm_h2(ptr(Ap), a_ps, lda, ptr(ra), ptr(Bp), b_ps, ldb, ptr(rb), tn, ptr(C), ldc, M, N, K, ptr(bias), act, ptr(dref), ldr, dact,
                           accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, torch.cuda.current_stream().cuda_stream), "cham_gemm_h2")


def _nt(gpu, M, N, K, bias=False, act=0, dref=False, seed=0, scale=1.0, check_ref=True, reps=1, row_spread=0.0, wide=None):
    """wide: None = the library's current NT kernel choice; 0 / 1 = force the 32-byte-piece (round 4) / 64-byte-piece (round 5) kernel."""
    lib = _lib_()
    if wide is not None:
        was = lib.cham_gemm_h2_set_nt_wide(int(wide))
        try:
            return _nt(gpu, M, N, K, bias, act, dref, seed, scale, check_ref, reps, row_spread)
        finally:
            lib.cham_gemm_h2_set_nt_wide(was)
    g = torch.Generator(device=gpu).manual_seed(seed)
    A = torch.randn(M, K, device=gpu, generator=g) * scale
    if row_spread:
        A = A * torch.exp2(-row_spread * torch.rand(M, 1, device=gpu, generator=g))
    B = torch.randn(N, K, device=gpu, generator=g) * (K ** -0.5)
    bias_t = torch.randn(N, device=gpu, generator=g) * scale if bias else None
    Y = torch.randn(M, N, device=gpu, generator=g) if dref else None
    (Ap, ra), (Bp, rb) = split2h_dev(A), split2h_dev(B)
    Yh = split2h_dev(Y)[0][0].contiguous() if dref else None
    C = torch.full((M, N), float('nan'), device=gpu)
    outs = []
    for _ in range(reps):
        C.fill_(float('nan'))
        _gemm(lib, Ap, M * K, K, ra, Bp, N * K, K, rb, 0, C, N, M, N, K, bias=bias_t, act=act, dref=Yh, ldr=N, dact=1 if dref else 0)
        torch.cuda.synchronize()
        outs.append(C.clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "plane GEMM is not repeatable (race?)"
    if not check_ref:
        return outs[0]
    R = A.double() @ B.double().t()
    if bias:
        R = R + bias_t.double()
    if act == 2:
        R = torch.tanh(R)
    if dref:
        R = R * torch.where(Y.double() > 0, 1.0, 0.2)
    return float((outs[0].double() - R).abs().max()) / max(scale if not row_spread else 0.0, float(R.abs().max()))


def _tn(gpu, M, N, K, splits=1, accumulate=0, seed=0, reps=1):
    lib = _lib_()
    g = torch.Generator(device=gpu).manual_seed(seed)
    A = torch.randn(K, M, device=gpu, generator=g)
    B = torch.randn(K, N, device=gpu, generator=g)
    C0 = torch.randn(M, N, device=gpu, generator=g)
    (Ap, ra), (Bp, rb) = split2h_dev(A), split2h_dev(B)
    ws = torch.empty(32 << 20, dtype=torch.float32, device=gpu) if splits != 1 else None
    outs = []
    for _ in range(reps):
        C = C0.clone() if accumulate else torch.full((M, N), float('nan'), device=gpu)
        _gemm(lib, Ap, K * M, M, ra, Bp, K * N, N, rb, 1, C, N, M, N, K, accumulate=accumulate, ws=ws, splits=splits)
        torch.cuda.synchronize()
        outs.append(C)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "plane GEMM is not repeatable (race?)"
    R = A.double().t() @ B.double()
    if accumulate:
        R = R + C0.double()
    return float((outs[0].double() - R).abs().max()) / float(R.abs().max())


def test_scale_records_and_split_kernel(gpu):
    """cham_h2_scale_absmax (one and two arrays), cham_h2_scale_rownorm (K = 128 and generic, with a factor), cham_split2h (planes + transposed
    planes bit-exact against the host formula, sign kept on underflow), and that a record can be re-used (the kernels clear their scratch)."""
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib_()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=gpu).manual_seed(5)
    X = torch.randn(300, 264, device=gpu, generator=g) * torch.exp2(-30 * torch.rand(300, 1, device=gpu, generator=g))
    Y = torch.randn(1000, 128, device=gpu, generator=g) * 3.0
    rec = new_rec(gpu)
    for _ in range(3):          # re-use of the record
        check(lib.cham_h2_scale_absmax(ptr(X), X.numel(), None, 0, ptr(rec), st), "absmax")
        torch.cuda.synchronize()
        r = rec.cpu().numpy()
        b = float(X.abs().max())
        assert r[2] == np.float32(b) and 2.0 ** 14 <= b * r[0] < 2.0 ** 15 and r[0] * r[1] == 1.0 and not rec.view(torch.int32)[4:7].any()
    check(lib.cham_h2_scale_absmax(ptr(X), X.numel(), ptr(Y), Y.numel(), ptr(rec), st), "absmax2")
    torch.cuda.synchronize()
    assert rec.cpu().numpy()[2] == np.float32(float(X.abs().max()) + float(Y.abs().max()))
    rn, rn2 = new_rec(gpu), new_rec(gpu)
    check(lib.cham_h2_scale_rownorm(ptr(Y), 1000, 128, 128, None, ptr(rn), st), "rownorm")
    check(lib.cham_h2_scale_rownorm(ptr(X), 300, 264, 264, rn.data_ptr() + 8, ptr(rn2), st), "rownorm2")
    torch.cuda.synchronize()
    ny, nx = float(Y.double().norm(dim=1).max()), float(X.double().norm(dim=1).max())
    assert ny <= rn.cpu().numpy()[2] <= ny * 1.002
    assert nx * ny <= rn2.cpu().numpy()[2] <= nx * ny * 1.004
    P, PT, r = split2h_dev(X, transposed=True)
    torch.cuda.synchronize()
    ref = split2h_ref(X, float(r[0]))
    assert torch.equal(P.view(torch.int16), ref.view(torch.int16))
    assert torch.equal(PT.view(torch.int16), ref.transpose(1, 2).contiguous().view(torch.int16))
    assert torch.equal(P[0].view(torch.int16) > 0, X > 0)          # sign of the h plane == sign of x everywhere (leaky' is read from it)
    back = (P[0].double() + P[1].double()) / float(r[0])
    big = X.abs() * float(r[0]) >= 2.0 ** -3
    assert float(((back - X.double()).abs()[big] / X.double().abs()[big]).max()) <= 2.0 ** -22
    assert float((back - X.double()).abs()[~big].max()) <= 2.0 ** -24 / float(r[0])


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (256, 256, 32), (256, 256, 48), (256, 256, 64), (256, 256, 80), (512, 512, 64), (300, 260, 96),
                                   (1000, 1024, 1024), (77, 520, 416), (1, 4, 16), (513, 256, 1024)])
def test_h2_nt(gpu, M, N, K):
    assert _nt(gpu, M, N, K) < 5e-5
    assert _nt(gpu, M, N, K, bias=True, act=2) < 5e-5
    assert _nt(gpu, M, N, K, bias=True) < 5e-5
    assert _nt(gpu, M, N, K, dref=True) < 5e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (256, 256, 64), (256, 256, 96), (512, 512, 128), (300, 260, 96), (1000, 1024, 1024), (77, 520, 416),
                                   (1, 4, 32), (513, 256, 1024), (4096, 1024, 1024)])
def test_h2_nt_wide_pieces_bit_identical_to_narrow(gpu, M, N, K):
    """gemm_h2w_kernel (64-byte source pieces: two 16-k chunks of a row per LDS-DMA request, two 32-k buffers, one barrier per two steps)
    against gemm_h2_kernel<false> (32-byte pieces, four 16-k stages): the same three plane products per chunk in the same order, so the
    outputs must be BIT-identical - every epilogue, ragged rows / columns (descriptor range checks of the new piece map), one to 32
    double chunks (prologue / branchy tail of the ring), repeated launches (race screen: a fragment read overtaking its DMA is noise)."""
    lib = _lib_()
    for kw in (dict(), dict(bias=True, act=2), dict(bias=True), dict(dref=True)):
        _counts(lib, reset=True)
        a = _nt(gpu, M, N, K, check_ref=False, wide=0, seed=11, **kw)
        assert _counts(lib)[2] == 0
        b = _nt(gpu, M, N, K, check_ref=False, wide=1, seed=11, reps=3, **kw)
        assert _counts(lib)[2] == 3, _counts(lib)
        assert torch.equal(a, b), (kw, float((a - b).abs().max()))
        assert _nt(gpu, M, N, K, wide=1, seed=11, **kw) < 5e-5


def test_h2_nt_wide_is_the_default_and_needs_k_mod_32(gpu):
    lib = _lib_()
    assert lib.cham_gemm_h2_set_nt_wide(1) in (0, 1)
    _counts(lib, reset=True)
    _nt(gpu, 256, 256, 64, check_ref=False)
    _nt(gpu, 256, 256, 48, check_ref=False)          # K % 32 != 0: the 32-byte-piece kernel keeps it
    c = _counts(lib)
    assert c[0] == 2 and c[2] == 1, c


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (256, 256, 40), (256, 256, 70), (256, 512, 777), (512, 256, 3001), (1024, 1024, 5000), (256, 256, 1)])
def test_h2_tn_wgrad_splitk(gpu, M, N, K):
    assert _tn(gpu, M, N, K) < 1e-4
    assert _tn(gpu, M, N, K, splits=0) < 1e-4
    assert _tn(gpu, M, N, K, splits=7) < 1e-4
    assert _tn(gpu, M, N, K, splits=8) < 1e-4          # multiples of 8: one-K-split-per-XCD placement
    assert _tn(gpu, M, N, K, splits=0, accumulate=1) < 1e-4


def test_h2_exact_on_small_integers_and_layout(gpu):
    """Operands that ARE fp16 numbers have empty low planes and small-integer products are exact; A = I against an asymmetric B catches row /
    column swaps of the fragment maps, the swizzles and the C/D map, in both layouts."""
    lib = _lib_()
    n = 512
    I = torch.eye(n, device=gpu)
    B = (torch.arange(n * n, dtype=torch.float32, device=gpu).reshape(n, n) % 97 - 31.0)
    (Ip, ri), (Bp, rb) = split2h_dev(I), split2h_dev(B)
    assert not Ip[1].any() and not Bp[1].any()
    C = torch.zeros(n, n, device=gpu)
    _gemm(lib, Ip, n * n, n, ri, Bp, n * n, n, rb, 0, C, n, n, n, n)
    torch.cuda.synchronize()
    assert torch.equal(C, B.t())              # C = I B^T
    C.zero_()
    _gemm(lib, Ip, n * n, n, ri, Bp, n * n, n, rb, 1, C, n, n, n, n)
    torch.cuda.synchronize()
    assert torch.equal(C, B)                  # C = I^T B
    C.zero_()
    _gemm(lib, Bp, n * n, n, rb, Ip, n * n, n, ri, 1, C, n, n, n, n)
    torch.cuda.synchronize()
    assert torch.equal(C, B.t())              # C = B^T I


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e12, 1e30])
def test_h2_dynamic_range_of_the_scale(gpu, scale):
    """The whole operand far from fp16's range: the power-of-two scale brings it back, results as accurate as at unit scale."""
    assert _nt(gpu, 300, 260, 96, scale=scale) < 5e-5


@pytest.mark.parametrize("spread", [12.0, 24.0, 36.0])
def test_h2_rows_of_very_different_magnitude(gpu, spread):
    """Rows spread over `spread` binary orders of magnitude under ONE per-tensor scale (what the gradient at the CAR tanh looks like): the
    error is bounded relative to the LARGEST output (absolute error of the small rows <= 2^-40 of the bound per element), which is what a
    sum over rows - the W2 weight gradient - and the max-normalised gradient checks of the parity tests see."""
    assert _nt(gpu, 1000, 512, 256, row_spread=spread, seed=3) < 5e-5


def test_h2_row_relative_error_follows_the_documented_contract(gpu):
    """The accuracy contract of a two-plane operand is relative to the matrix's BOUND, not to each row ("accuracy of the
    default fp32 arithmetic"): with the bound scaled to [2^14, 2^15), an element keeps 22 bits while |x s| >= 2^-3 and an ABSOLUTE error of
    2^-25 / s below that (fp16 subnormal low plane).  For an output row whose A row sits 2^-d below the bound that is a relative error of
    max(~2^-21, ~2^(d - 40)) - fp32 grade down to d = 18, 1e-3 at d = 30 (the round-4 verdict's example) - which this test MEASURES row by
    row against float64 (rows spread over 2^40) and holds to 4 x that model, next to the native fp32 MFMA on the same operands (row-relative
    ~1e-6 at every magnitude).  What it means for the step: the two-plane operands are Z1 (PreCAR output, rows within a few binary orders
    of each other) and dZ2 (gradient at the CAR tanh): a dZ2 row 2^-30 below the largest one carries 1e-3 relative error into its dZ1 row
    and its share of the W2 gradient - entries 2^-30 below the largest gradient entry, which TF-Adam's epsilon (1e-8 against sqrt(v))
    flattens anyway.  A per-row scale would factor out of the NT product but not out of the W2 weight gradient (the contraction runs over
    the rows), so the planes would have to be written twice: rejected on cost, contract stated instead."""
    lib = _lib_()
    M, N, K = 2048, 256, 512
    g = torch.Generator(device=gpu).manual_seed(21)
    d = torch.linspace(0.0, 40.0, M, device=gpu).unsqueeze(1)                    # row r sits 2^-d(r) below the largest row
    A = torch.randn(M, K, device=gpu, generator=g) * torch.exp2(-d)
    B = torch.randn(N, K, device=gpu, generator=g) * (K ** -0.5)
    (Ap, ra), (Bp, rb) = split2h_dev(A), split2h_dev(B)
    C = torch.empty(M, N, device=gpu)
    _gemm(lib, Ap, M * K, K, ra, Bp, N * K, K, rb, 0, C, N, M, N, K)
    Cn = torch.empty(M, N, device=gpu)
    _native(lib, A, K, 0, B, K, 1, Cn, M, N, K, None, 0, None, 0, None, 1)
    torch.cuda.synchronize()
    R = A.double() @ B.double().t()
    row_max = R.abs().amax(1)
    e_h2 = ((C.double() - R).abs().amax(1) / row_max).cpu().numpy()
    e_nat = ((Cn.double() - R).abs().amax(1) / row_max).cpu().numpy()
    dd = d.flatten().cpu().numpy()
    # the matrix bound is max |A| (row 0 region, ~4 sigma): rows sit d + ~2 binary orders below it
    model = np.maximum(2.0 ** -21, 2.0 ** (dd + 2.5 - 40.0))
    print("row-relative error, two-plane vs native fp32 MFMA, at d = 0 / 10 / 18 / 24 / 30 / 36: %s vs %s" % (
        ["%.1e" % e_h2[int(x / 40.0 * (M - 1))] for x in (0, 10, 18, 24, 30, 36)], ["%.1e" % e_nat[int(x / 40.0 * (M - 1))] for x in (0, 10, 18, 24, 30, 36)]))
    assert (e_h2 <= 4.0 * model).all(), (float((e_h2 / model).max()), int((e_h2 / model).argmax()))
    assert e_h2[dd <= 16.0].max() < 3e-6 and e_nat.max() < 1e-5               # fp32 grade while rows are within 2^-16 of the bound; native: everywhere
    assert e_h2[dd >= 34.0].min() > 1e-4                                      # ... and the degradation is real: the contract is not vacuous


def test_h2_is_repeatable_under_load(gpu):
    """Race screen: the same launch five times, bit-identical (a fragment read that overtakes its DMA shows up as run-to-run noise)."""
    _nt(gpu, 4096, 1024, 1024, bias=True, act=2, check_ref=False, reps=5)
    assert _tn(gpu, 1024, 1024, 40000, splits=0, reps=5) < 1e-4


def test_h2_argument_errors(gpu):
    from chameleon_recsys_amd._lib import ptr
    lib = _lib_()
    P = torch.zeros(2, 256, 256, dtype=torch.float16, device=gpu)
    r = new_rec(gpu); r[0] = 1.0; r[1] = 1.0
    C = torch.zeros(256, 256, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    ok = lambda *a: lib.cham_gemm_h2(*a, st)
    assert ok(ptr(P), 65536, 256, ptr(r), ptr(P), 65536, 256, ptr(r), 0, ptr(C), 256, 256, 256, 24, None, 0, None, 0, 0, 0, None, 0, 1) < 0      # NT: K % 16
    assert ok(ptr(P), 65536, 256, ptr(r), ptr(P), 65536, 256, ptr(r), 1, ptr(C), 256, 250, 256, 256, None, 0, None, 0, 0, 0, None, 0, 1) < 0     # TN: M % 256
    assert ok(ptr(P), 65536, 250, ptr(r), ptr(P), 65536, 256, ptr(r), 0, ptr(C), 256, 256, 256, 16, None, 0, None, 0, 0, 0, None, 0, 1) < 0      # lda % 8
    assert ok(None, 65536, 256, ptr(r), ptr(P), 65536, 256, ptr(r), 0, ptr(C), 256, 256, 256, 16, None, 0, None, 0, 0, 0, None, 0, 1) < 0
    assert ok(ptr(P), 65536, 256, None, ptr(P), 65536, 256, ptr(r), 0, ptr(C), 256, 256, 256, 16, None, 0, None, 0, 0, 0, None, 0, 1) < 0        # no scale record
    assert ok(ptr(P), 65536, 256, ptr(r), ptr(P), 65536, 256, ptr(r), 0, ptr(C), 256, 256, 256, 16, None, 1, None, 0, 0, 0, None, 0, 1) < 0      # act without bias


# ---- the benchmarked step's own shapes: error next to the native fp32 MFMA on the same operands ----------------------------------------
def _native(lib, A, lda, tA, B, ldb, tB, C, M, N, K, bias, act, dref, dact, ws, splits):
    from chameleon_recsys_amd._lib import check, ptr
    check(lib.cham_gemm_f32(ptr(A), lda, tA, ptr(B), ldb, tB, ptr(C), N, M, N, K, ptr(bias), act, ptr(dref), N, dact, None, 0, 1, 0, ptr(ws),
                            ws.numel() * 4 if ws is not None else 0, splits, torch.cuda.current_stream().cuda_stream), "native")
    torch.cuda.synchronize()


def test_h2_big_car_forward_and_dgrad(gpu):
    lib = _lib_()
    R, Cw = ROWS, 1024
    g = torch.Generator(device=gpu).manual_seed(1)
    A = torch.randn(R, Cw, device=gpu, generator=g)
    W = torch.randn(Cw, Cw, device=gpu, generator=g) * 0.03
    bias = torch.randn(Cw, device=gpu, generator=g)
    Y = torch.randn(R, Cw, device=gpu, generator=g)
    Ap, ra = split2h_dev(A)
    Wp, WTp, rw = split2h_dev(W, transposed=True)
    Yh = split2h_dev(Y)[0][0].contiguous()
    rows = torch.arange(0, R, R // 2048, device=gpu)[:2048]
    out = torch.empty(R, Cw, device=gpu)
    _counts(lib, reset=True)
    # forward: tanh(A W + b), B operand = planes of W^T
    _gemm(lib, Ap, R * Cw, Cw, ra, WTp, Cw * Cw, Cw, rw, 0, out, Cw, R, Cw, Cw, bias=bias, act=2)
    torch.cuda.synchronize()
    ref = torch.tanh(A[rows].double() @ W.double() + bias.double())
    e_h2 = float((out[rows].double() - ref).abs().max())
    _native(lib, A, Cw, 0, W, Cw, 0, out, R, Cw, Cw, bias, 2, None, 0, None, 1)
    e_nat = float((out[rows].double() - ref).abs().max())
    print("CAR forward: two-plane fp16 %.2e, native fp32 MFMA %.2e (abs, tanh outputs)" % (e_h2, e_nat))
    assert e_h2 < 3e-4 and e_h2 < 1.5 * e_nat + 1e-7, (e_h2, e_nat)
    # dgrad: (A W^T) * leaky'(Y), B operand = planes of W as stored
    _gemm(lib, Ap, R * Cw, Cw, ra, Wp, Cw * Cw, Cw, rw, 0, out, Cw, R, Cw, Cw, dref=Yh, ldr=Cw, dact=1)
    torch.cuda.synchronize()
    ref = (A[rows].double() @ W.double().t()) * torch.where(Y[rows].double() > 0, 1.0, 0.2)
    scale = float(ref.abs().max())
    e_h2 = float((out[rows].double() - ref).abs().max()) / scale
    _native(lib, A, Cw, 0, W, Cw, 1, out, R, Cw, Cw, None, 0, Y, 1, None, 1)
    e_nat = float((out[rows].double() - ref).abs().max()) / scale
    print("CAR dgrad: two-plane fp16 %.2e, native fp32 MFMA %.2e (of max)" % (e_h2, e_nat))
    assert e_h2 < 5e-5 and e_h2 < 1.5 * e_nat + 1e-7, (e_h2, e_nat)
    c = _counts(lib)
    assert c[0] == 2 and c[1] == 0 and c[2] == 2, c          # both NT launches on the 64-byte-piece kernel (K = 1024)


def test_h2_big_w2_wgrad_splitk(gpu):
    lib = _lib_()
    R, Cw = ROWS, 1024
    g = torch.Generator(device=gpu).manual_seed(4)
    A = torch.randn(R, Cw, device=gpu, generator=g)
    # a gradient-like second operand: rows spread over 30 binary orders of magnitude
    D = torch.randn(R, Cw, device=gpu, generator=g) * torch.exp2(-30 * torch.rand(R, 1, device=gpu, generator=g))
    (Ap, ra), (Dp, rd) = split2h_dev(A), split2h_dev(D)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=gpu)
    out = torch.empty(Cw, Cw, device=gpu)
    ref = A.double().t() @ D.double()
    scale = float(ref.abs().max())
    _native(lib, A, Cw, 1, D, Cw, 0, out, Cw, Cw, R, None, 0, None, 0, ws, 0)
    e_nat = float((out.double() - ref).abs().max()) / scale
    for splits in (0, 8, 16, 32):
        out.fill_(float('nan'))
        _gemm(lib, Ap, R * Cw, Cw, ra, Dp, R * Cw, Cw, rd, 1, out, Cw, Cw, Cw, R, ws=ws, splits=splits)
        torch.cuda.synchronize()
        e_h2 = float((out.double() - ref).abs().max()) / scale
        print("W2 wgrad (%d splits): two-plane fp16 %.2e, native fp32 MFMA %.2e (of max)" % (splits, e_h2, e_nat))
        assert e_h2 < 1e-4 and e_h2 < 1.5 * e_nat + 1e-7, (splits, e_h2, e_nat)


def test_h2_producers_match_the_three_plane_ones(gpu):
    """cham_combine_fwd_h2 / cham_mulpred_bwd_h2 / cham_dm_mulpred_h2 against their three-bf16-plane twins on the same inputs: the planes of
    either format sum back to the same fp32 values (to the formats' own resolution), dpred / b2 partial sums identical."""
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib_()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=gpu).manual_seed(9)
    C, BT, N, pmax = 256, 37, 50, 1000
    NC, RV = N + 1, 2 * 37 + 1000 + 1
    Rc = BT * NC
    U = torch.randn(BT, C, device=gpu, generator=g)
    V = torch.randn(RV, C, device=gpu, generator=g)
    neg_slot = torch.randint(0, pmax, (BT, N), device=gpu, generator=g, dtype=torch.int32)
    Z3 = torch.zeros(3, Rc, C, dtype=torch.bfloat16, device=gpu)
    Z2h = torch.zeros(2, Rc, C, dtype=torch.float16, device=gpu)
    rec = new_rec(gpu)
    check(lib.cham_combine_fwd_p3(ptr(U), ptr(V), C, BT, N, pmax, ptr(neg_slot), ptr(Z3), Rc * C, st), "p3")
    check(lib.cham_h2_scale_absmax(ptr(U), U.numel(), ptr(V), V.numel(), ptr(rec), st), "scale")
    check(lib.cham_combine_fwd_h2(ptr(U), ptr(V), C, BT, N, pmax, ptr(neg_slot), ptr(Z2h), Rc * C, ptr(rec), st), "h2")
    torch.cuda.synchronize()
    z3 = Z3.double().sum(0)
    z2 = Z2h.double().sum(0) * float(rec[1])
    assert float(Z2h[0].float().abs().max()) <= 2.0 ** 15
    assert float((z3 - z2).abs().max()) <= 2.0 ** -21 * float(z3.abs().max())
    assert torch.equal(Z2h[0].view(torch.int16) > 0, z3 > 0)
    # gradient at the CAR tanh: unfused and fused forms
    dS1 = torch.randn(Rc, 128, device=gpu, generator=g) * torch.exp2(-20 * torch.rand(Rc, 1, device=gpu, generator=g))
    Ws1 = torch.randn(C, 128, device=gpu, generator=g) * 0.05
    Z2c = torch.tanh(torch.randn(Rc, C, device=gpu, generator=g))
    pred = torch.tanh(torch.randn(BT, C, device=gpu, generator=g))
    Wp = torch.zeros(3, C, 128, dtype=torch.bfloat16, device=gpu)
    check(lib.cham_split3(ptr(Ws1), C, 128, 128, ptr(Wp), C * 128, 128, None, 0, 0, st), "split3")
    rn, rd = new_rec(gpu), new_rec(gpu)
    check(lib.cham_h2_scale_rownorm(ptr(Ws1), C, 128, 128, None, ptr(rn), st), "rownorm")
    check(lib.cham_h2_scale_rownorm(ptr(dS1), Rc, 128, 128, rn.data_ptr() + 8, ptr(rd), st), "rownorm")
    D3 = torch.zeros(3, Rc, C, dtype=torch.bfloat16, device=gpu)
    D2 = torch.zeros(2, Rc, C, dtype=torch.float16, device=gpu)
    dp3, dp2, b3, b2 = (torch.zeros(BT, C, device=gpu) for _ in range(4))
    check(lib.cham_dm_mulpred_p3(ptr(dS1), 128, 128, ptr(Wp), C * 128, ptr(Z2c), ptr(pred), C, BT, N, ptr(D3), Rc * C, ptr(dp3), ptr(b3), st), "dm p3")
    check(lib.cham_dm_mulpred_h2(ptr(dS1), 128, 128, ptr(Wp), C * 128, ptr(Z2c), ptr(pred), C, BT, N, ptr(D2), Rc * C, ptr(rd), ptr(dp2), ptr(b2), st), "dm h2")
    torch.cuda.synchronize()
    d3 = D3.double().sum(0)
    d2 = D2.double().sum(0) * float(rd[1])
    dM = dS1.double() @ Ws1.double().t()
    assert float(dM.abs().max()) <= float(rd[2]), "the Cauchy-Schwarz bound must hold"
    assert float(D2[0].float().abs().max()) <= 2.0 ** 15 and torch.isfinite(D2.float()).all()
    assert float((d3 - d2).abs().max()) <= 2.0 ** -21 * float(d3.abs().max()) + 2.0 ** -24 * float(rd[1])
    assert torch.equal(dp3, dp2) and torch.equal(b3, b2)
    # unfused form on the fp32 dM
    dMf = dM.float().contiguous()
    D2u = torch.zeros(2, Rc, C, dtype=torch.float16, device=gpu)
    dpu, bu = torch.zeros(BT, C, device=gpu), torch.zeros(BT, C, device=gpu)
    check(lib.cham_mulpred_bwd_h2(ptr(dMf), ptr(Z2c), ptr(pred), C, BT, N, ptr(dpu), ptr(D2u), Rc * C, ptr(bu), ptr(rd), st), "mulpred h2")
    torch.cuda.synchronize()
    d2u = D2u.double().sum(0) * float(rd[1])
    assert float((d2u - d3).abs().max()) <= 1e-5 * float(d3.abs().max())


# ---- TILE-BLOCKED operands (round 6; common.h h2b_index, include/chameleon_nar.h cham_gemm_h2b): the 16-byte pieces the kernels move are the
# ---- same element sets in both layouts - same LDS images, same products in the same order: BIT-IDENTICAL to the row-major operands
def _gemm_b(lib, Ap, a_ps, lda, ra, Bp, b_ps, ldb, rb, tn, C, ldc, M, N, K, bias=None, act=0, dref=None, ldr=0, dact=0, accumulate=0, ws=None, splits=1,
            a_tiles=0, b_tiles=0, dref_blocked=0, expect=0):
    from chameleon_recsys_amd._lib import ptr
    rc = lib.cham_gemm_h2b(ptr(Ap), a_ps, lda, ptr(ra), ptr(Bp), b_ps, ldb, ptr(rb), tn, ptr(C), ldc, M, N, K, ptr(bias), act, ptr(dref), ldr, dact,
                           accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, a_tiles, b_tiles, dref_blocked,
                           torch.cuda.current_stream().cuda_stream)
    assert rc == expect, rc


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (700, 320, 64), (1024, 1024, 1024), (5 * 256 + 51, 1024, 1024), (255, 96, 128), (12 * 51, 256, 512)])
@pytest.mark.parametrize("epi", ["plain", "bias", "tanh", "dgrad"])
def test_h2_nt_blocked_a_bit_identical_to_row_major(gpu, M, N, K, epi):
    from chameleon_recsys_amd.nar.nar_model import planes_from_blocked, planes_to_blocked
    lib = _lib_()
    g = torch.Generator(device=gpu).manual_seed(M + N + K)
    A = torch.randn(M, K, device=gpu, generator=g)
    B = torch.randn(N, K, device=gpu, generator=g) * (K ** -0.5)
    bias_t = torch.randn(N, device=gpu, generator=g) if epi in ("bias", "tanh") else None
    (Ap, ra), (Bp, rb) = split2h_dev(A), split2h_dev(B)
    Ab, tiles = planes_to_blocked(Ap)
    assert torch.equal(planes_from_blocked(Ab, tiles, K)[:, :M], Ap)
    # garbage in the rows beyond M of the last tile must not matter (NT: rows are independent, those outputs are never stored)
    junk = planes_from_blocked(Ab, tiles, K).clone(); junk[:, M:] = float('nan'); Ab = planes_to_blocked(junk)[0]
    a_ps = Ab[0].numel()
    Yh = Yb = None
    if epi == "dgrad":      # the saved activation's h plane, [M, N]: row-major and tile-blocked (N % 32 == 0)
        Yh = split2h_dev(torch.randn(M, N, device=gpu, generator=g))[0][0].contiguous()
        Yb = planes_to_blocked(Yh[None])[0][0].contiguous()
    ref, got = (torch.full((M, N), float('nan'), device=gpu) for _ in range(2))
    kw = dict(bias=bias_t, act=2 if epi == "tanh" else 0, ldr=N, dact=1 if epi == "dgrad" else 0)
    _gemm_b(lib, Ap, M * K, K, ra, Bp, N * K, K, rb, 0, ref, N, M, N, K, dref=Yh, **kw)
    c0 = _counts(lib)
    _gemm_b(lib, Ab, a_ps, K, ra, Bp, N * K, K, rb, 0, got, N, M, N, K, dref=Yb, a_tiles=tiles, dref_blocked=1 if epi == "dgrad" else 0, **kw)
    c1 = _counts(lib)
    torch.cuda.synchronize()
    assert c1[3] == c0[3] + 1 and c1[2] == c0[2] + 1            # the blocked-A instance of the 64-byte-piece kernel ran
    assert not torch.isnan(ref).any() and torch.equal(ref, got)
    if epi == "dgrad":      # blocked A with a row-major activation plane, and the other way round
        got2, got3 = (torch.full((M, N), float('nan'), device=gpu) for _ in range(2))
        _gemm_b(lib, Ab, a_ps, K, ra, Bp, N * K, K, rb, 0, got2, N, M, N, K, dref=Yh, a_tiles=tiles, **kw)
        _gemm_b(lib, Ap, M * K, K, ra, Bp, N * K, K, rb, 0, got3, N, M, N, K, dref=Yb, dref_blocked=1, **kw)
        assert torch.equal(ref, got2) and torch.equal(ref, got3)


@pytest.mark.parametrize("M,N,K,splits", [(256, 256, 16, 1), (256, 512, 700, 1), (512, 256, 5000, 0), (1024, 1024, 9 * 256 * 4 + 51 * 3, 32),
                                          (256, 256, 1000, 3), (1024, 1024, 255 * 51, 7)])
@pytest.mark.parametrize("which", ["ab", "a", "b"])
def test_h2_tn_blocked_operands_bit_identical_to_row_major(gpu, M, N, K, splits, which):
    """TN contracts over the ROWS: split boundaries inside a row tile (k-chunks that are no multiple of 256), a reduction length that
    is no multiple of 16, and STALE non-zero data in the tile's rows beyond K (an earlier, longer step) - masked by the kernel."""
    from chameleon_recsys_amd.nar.nar_model import planes_from_blocked, planes_to_blocked
    lib = _lib_()
    g = torch.Generator(device=gpu).manual_seed(K + splits)
    A = torch.randn(K, M, device=gpu, generator=g)
    B = torch.randn(K, N, device=gpu, generator=g)
    (Ap, ra), (Bp, rb) = split2h_dev(A), split2h_dev(B)
    def blocked(P, ld):
        Pb, tiles = planes_to_blocked(P)
        rm = planes_from_blocked(Pb, tiles, ld).clone()
        rm[:, K:] = torch.randn(rm[:, K:].shape, device=gpu, generator=g).half() * 100      # stale rows: finite, non-zero
        return planes_to_blocked(rm)[0], tiles
    Ab, ta = blocked(Ap, M)
    Bb, tb = blocked(Bp, N)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=gpu) if splits != 1 else None
    ref, got = (torch.full((M, N), float('nan'), device=gpu) for _ in range(2))
    _gemm_b(lib, Ap, K * M, M, ra, Bp, K * N, N, rb, 1, ref, N, M, N, K, ws=ws, splits=splits)
    ua, ub = "a" in which, "b" in which
    c0 = _counts(lib)
    _gemm_b(lib, Ab if ua else Ap, Ab[0].numel() if ua else K * M, M, ra, Bb if ub else Bp, Bb[0].numel() if ub else K * N, N, rb, 1, got, N, M, N, K, ws=ws,
            splits=splits, a_tiles=ta if ua else 0, b_tiles=tb if ub else 0)
    torch.cuda.synchronize()
    assert _counts(lib)[4] == c0[4] + 1
    assert not torch.isnan(ref).any() and torch.equal(ref, got)
    err = float((got.double() - A.double().t() @ B.double()).abs().max()) / float((A.double().t() @ B.double()).abs().max())
    assert err < 2e-6, err


def test_h2_blocked_argument_checks(gpu):
    lib = _lib_()
    blk = lib.cham_h2b_block_elements()
    assert blk >= 8192 and blk % 8 == 0
    ps = 2 * 2 * blk                                  # two row tiles x two column blocks (ld = 64)
    P = torch.zeros(2, ps, dtype=torch.float16, device=gpu)
    W = torch.zeros(2, 256, 64, dtype=torch.float16, device=gpu)
    r = new_rec(gpu)
    C = torch.zeros(300, 256, device=gpu)
    ok = dict(expect=0)
    _gemm_b(lib, P, ps, 64, r, W, 256 * 64, 64, r, 0, C, 256, 300, 256, 64, a_tiles=2, **ok)
    bad = dict(expect=-22)
    _gemm_b(lib, P, ps, 64, r, W, 256 * 64, 64, r, 0, C, 256, 300, 256, 64, a_tiles=1, **bad)            # too few row tiles for M = 300
    _gemm_b(lib, P, ps // 2, 64, r, W, 256 * 64, 64, r, 0, C, 256, 300, 256, 64, a_tiles=2, **bad)       # planes overlap
    _gemm_b(lib, P, ps, 64, r, W, 256 * 64, 64, r, 0, C, 256, 300, 256, 64, b_tiles=1, **bad)            # NT: B (weights) is row-major
    _gemm_b(lib, P, ps, 64, r, W, 256 * 64, 64, r, 0, C, 256, 300, 256, 48, a_tiles=2, **bad)            # ld != K / K % 32
    was = lib.cham_gemm_h2_set_nt_wide(0)
    try:
        _gemm_b(lib, P, ps, 64, r, W, 256 * 64, 64, r, 0, C, 256, 300, 256, 64, a_tiles=2, **bad)        # needs the 64-byte-piece kernel
    finally:
        lib.cham_gemm_h2_set_nt_wide(was)


@pytest.mark.parametrize("BT,N,C", [(7, 50, 1024), (64, 50, 256), (33, 31, 128), (5, 100, 512)])
def test_blocked_plane_producers_write_the_row_major_planes_bit_for_bit(gpu, BT, N, C):
    """cham_combine_fwd_h2b (PreCAR output) and cham_dm_mulpred_h2_blk (gradient at the CAR tanh, both arithmetics) in the tile-blocked
    layout against their row-major forms: same bits after un-blocking, rows beyond BT (1 + N) untouched (zero), every other output equal."""
    from chameleon_recsys_amd._lib import check, ptr
    from chameleon_recsys_amd.nar.nar_model import blocked_plane_elements, planes_from_blocked
    lib = _lib_()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=gpu).manual_seed(BT * N + C)
    NC, pmax = N + 1, 40
    R = BT * NC
    tiles = -(-R // 256) + 1
    U = torch.randn(BT, C, device=gpu, generator=g)
    V = torch.randn(2 * BT + pmax + 1, C, device=gpu, generator=g)
    neg_slot = torch.randint(0, pmax + 1, (BT, N), device=gpu, generator=g, dtype=torch.int32)
    rec = new_rec(gpu)
    check(lib.cham_h2_scale_absmax(ptr(U), U.numel(), ptr(V), V.numel(), ptr(rec), st), "absmax")
    Zr = torch.zeros(2, R, C, dtype=torch.float16, device=gpu)
    bps = blocked_plane_elements(tiles, C)
    Zb = torch.zeros(2, bps, dtype=torch.float16, device=gpu)
    check(lib.cham_combine_fwd_h2b(ptr(U), ptr(V), C, BT, N, pmax, ptr(neg_slot), ptr(Zr), R * C, ptr(rec), 0, st), "row-major")
    check(lib.cham_combine_fwd_h2b(ptr(U), ptr(V), C, BT, N, pmax, ptr(neg_slot), ptr(Zb), bps, ptr(rec), 1, st), "blocked")
    un = planes_from_blocked(Zb, tiles, C)
    assert torch.equal(un[:, :R].view(torch.int16), Zr.view(torch.int16)) and not un[:, R:].any()
    assert lib.cham_combine_fwd_h2b(ptr(U), ptr(V), C, BT, N, pmax, ptr(neg_slot), ptr(Zb), R * C - 8, ptr(rec), 1, st) == -22      # plane stride < whole tiles
    if not (32 <= NC <= 256 and C % 64 == 0):
        return
    # ---- the fused scorer dgrad
    dS1 = torch.randn(R, 128, device=gpu, generator=g) * 1e-3
    Ws1 = torch.randn(C, 128, device=gpu, generator=g) * 0.05
    Z2c = torch.tanh(torch.randn(R, C, device=gpu, generator=g))
    pred = torch.tanh(torch.randn(BT, C, device=gpu, generator=g))
    sc_ws1n, sc_out, sc_ds1 = new_rec(gpu), new_rec(gpu), new_rec(gpu)
    check(lib.cham_h2_scale_rownorm(ptr(Ws1), C, 128, 128, None, ptr(sc_ws1n), st), "rownorm w")
    check(lib.cham_h2_scale_rownorm2(ptr(dS1), R, 128, 128, ptr(sc_ws1n[2:]), ptr(sc_out), ptr(sc_ds1), st), "rownorm ds1")
    Wh = torch.zeros(2, C, 128, dtype=torch.float16, device=gpu)
    check(lib.cham_split2h(ptr(Ws1), C, 128, 128, ptr(Wh), C * 128, 128, None, 0, 0, ptr(sc_ws1n), 0, st), "split2h")
    W3 = torch.zeros(3, C, 128, dtype=torch.bfloat16, device=gpu)
    check(lib.cham_split3(ptr(Ws1), C, 128, 128, ptr(W3), C * 128, 128, None, 0, 0, st), "split3")
    for f16 in (True, False):
        outs = []
        for blk in (0, 1):
            D = torch.zeros(2, bps, dtype=torch.float16, device=gpu) if blk else torch.zeros(2, R, C, dtype=torch.float16, device=gpu)
            dpred, b2 = torch.zeros(BT, C, device=gpu), torch.zeros(BT, C, device=gpu)
            ps = D[0].numel()
            if blk:
                check(lib.cham_dm_mulpred_h2_blk(ptr(dS1), 128, 128, ptr(Wh if f16 else W3), C * 128, ptr(sc_ds1) if f16 else None, ptr(sc_ws1n) if f16 else None,
                                                 ptr(Z2c), ptr(pred), C, BT, N, ptr(D), ps, ptr(sc_out), ptr(dpred), ptr(b2), st), "blk")
            elif f16:
                check(lib.cham_dm_mulpred_h2h(ptr(dS1), 128, 128, ptr(Wh), C * 128, ptr(sc_ds1), ptr(sc_ws1n), ptr(Z2c), ptr(pred), C, BT, N, ptr(D), ps,
                                              ptr(sc_out), ptr(dpred), ptr(b2), st), "h2h")
            else:
                check(lib.cham_dm_mulpred_h2(ptr(dS1), 128, 128, ptr(W3), C * 128, ptr(Z2c), ptr(pred), C, BT, N, ptr(D), ps, ptr(sc_out), ptr(dpred), ptr(b2), st), "h2")
            outs.append((D, dpred, b2))
        (Dr, dpr, b2r), (Db, dpb, b2b) = outs
        un = planes_from_blocked(Db, tiles, C)
        assert Dr.abs().max() > 0
        assert torch.equal(un[:, :R].view(torch.int16), Dr.view(torch.int16)) and not un[:, R:].any(), f16
        assert torch.equal(dpr, dpb) and torch.equal(b2r, b2b)


@pytest.mark.parametrize("BT,N,C,K", [(7, 50, 256, 64), (40, 50, 1024, 128), (64, 31, 256, 96), (5, 100, 512, 64), (3, 127, 256, 32), (2, 300, 256, 64),
                                      (97, 50, 260, 64), (1, 50, 256, 64)])
@pytest.mark.parametrize("wide", [1, 0])
def test_h2_dgrad_group_sums(gpu, BT, N, C, K, wide):
    """Round 6: the CAR dgrad's epilogue leaves the column sums of every click's N + 1 candidate rows (cham_gemm_h2_dgrad_gs), and
    cham_combine_bwd_gs builds dU from those pieces instead of reading the rows again.  The dgrad's own output must be bit-identical to the
    plain launch; dU must agree with float64 sums of THAT output to fp32 summation accuracy; dV (rows copied / slot sums) bit-identical to
    cham_combine_bwd's.  Shapes: groups of 32 ... 301 rows against the 128-row chunks and 256-row tiles, a ragged last tile, ragged columns
    (260), one click, both NT kernels."""
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib_()
    st = torch.cuda.current_stream().cuda_stream
    G, M = N + 1, BT * (N + 1)
    g = torch.Generator(device='cpu').manual_seed(BT * 1000 + N)
    A = torch.randn(M, K, generator=g).to(gpu)
    W = torch.randn(C, K, generator=g).to(gpu) * 0.1
    Y = torch.randn(M, C, generator=g).to(gpu)
    Ap, ra = split2h_dev(A)
    Wp, rw = split2h_dev(W)
    Yp, _ = split2h_dev(Y)
    was = lib.cham_gemm_h2_set_nt_wide(wide)
    try:
        dpre0 = torch.zeros(BT + M, C, device=gpu)
        dpre1 = torch.zeros(BT + M, C, device=gpu)
        dpre_in = torch.randn(BT, C, generator=g).to(gpu)
        dpre0[:BT] = dpre_in; dpre1[:BT] = dpre_in
        _gemm(lib, Ap, M * K, K, ra, Wp, C * K, K, rw, 0, dpre0[BT:], C, M, C, K, dref=Yp, ldr=C, dact=1)
        gs = torch.full((int(lib.cham_gemm_h2_groupsum_bytes(M, C, G)) // 4,), float('nan'), device=gpu)
        check(lib.cham_gemm_h2_dgrad_gs(ptr(Ap), M * K, K, ptr(ra), ptr(Wp), C * K, K, ptr(rw), ptr(dpre1[BT:]), C, M, C, K, ptr(Yp), C, 0, 0, G,
                                        ptr(gs), gs.numel() * 4, st), "cham_gemm_h2_dgrad_gs")
    finally:
        lib.cham_gemm_h2_set_nt_wide(was)
    assert torch.equal(dpre0, dpre1)
    pmax = 11
    gi = torch.Generator(device='cpu').manual_seed(5)
    neg_slot = torch.randint(0, pmax + 1, (BT, N), generator=gi, dtype=torch.int32)
    for b in range(BT):       # a click holds a pool slot at most once (the sampler's contract); the rest is the zero-padding slot
        seen = set()
        for n in range(N):
            s = int(neg_slot[b, n])
            if s < pmax and s in seen:
                neg_slot[b, n] = pmax
            seen.add(s)
    neg_slot = neg_slot.to(gpu)
    RV = 2 * BT + pmax + 1
    ws = torch.zeros(int(lib.cham_combine_bwd_workspace_bytes(C, BT, N, pmax)) // 4 + 64, device=gpu)
    dU0, dV0 = torch.zeros(BT, C, device=gpu), torch.zeros(RV, C, device=gpu)
    dU1, dV1 = torch.zeros(BT, C, device=gpu), torch.zeros(RV, C, device=gpu)
    if C % 4 == 0:
        check(lib.cham_combine_bwd(ptr(dpre0), C, BT, N, pmax, ptr(neg_slot), ptr(dU0), ptr(dV0), ptr(ws), ws.numel() * 4, st), "cham_combine_bwd")
        check(lib.cham_combine_bwd_gs(ptr(dpre1), C, BT, N, pmax, ptr(neg_slot), ptr(dU1), ptr(dV1), ptr(ws), ws.numel() * 4, ptr(gs), gs.numel() * 4, st),
              "cham_combine_bwd_gs")
        assert torch.equal(dV0, dV1)
        ref = dpre_in.double() + dpre1[BT:].double().view(BT, G, C).sum(1)
        mag = dpre_in.double().abs() + dpre1[BT:].double().abs().view(BT, G, C).sum(1)
        err = ((dU1.double() - ref).abs() / mag).max().item()
        err0 = ((dU0.double() - ref).abs() / mag).max().item()
        assert err < 4e-7 and err <= 4 * max(err0, 1e-8), (err, err0)
        # every piece the consumer read was written (no NaN left in the sums), and the run is repeatable bit for bit
        assert torch.isfinite(dU1).all()
        gs2 = torch.full_like(gs, float('nan'))
        check(lib.cham_gemm_h2_dgrad_gs(ptr(Ap), M * K, K, ptr(ra), ptr(Wp), C * K, K, ptr(rw), ptr(dpre1[BT:]), C, M, C, K, ptr(Yp), C, 0, 0, G,
                                        ptr(gs2), gs2.numel() * 4, st), "cham_gemm_h2_dgrad_gs")
        assert torch.equal(torch.nan_to_num(gs, nan=-1.0), torch.nan_to_num(gs2, nan=-1.0))
    # argument checks: groups under 32 rows, a short buffer, the TN form
    assert lib.cham_gemm_h2_dgrad_gs(ptr(Ap), M * K, K, ptr(ra), ptr(Wp), C * K, K, ptr(rw), ptr(dpre1[BT:]), C, M, C, K, ptr(Yp), C, 0, 0, 31,
                                     ptr(gs), gs.numel() * 4, st) < 0
    assert lib.cham_gemm_h2_dgrad_gs(ptr(Ap), M * K, K, ptr(ra), ptr(Wp), C * K, K, ptr(rw), ptr(dpre1[BT:]), C, M, C, K, ptr(Yp), C, 0, 0, G,
                                     ptr(gs), gs.numel() * 4 - 4, st) < 0
