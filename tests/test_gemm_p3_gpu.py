"""Plane-product GEMM over pre-split bf16 planes (csrc/gemm_p3.hip: LDS-DMA staging, 256x256 tile, three-stage ring) against a float64
reference: both layouts, every epilogue, ragged rows / columns / K tails (descriptor range checks), one- and two-step reductions
(pipeline prologue), split-K placements, repeatability (race screen), and - on the step's own shapes - its error next to the native
fp32 MFMA's on the same operands (fp32-GRADE error, as tests/test_gemm_x3_gpu.py demands of the on-the-fly split)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = 248064          # B * T * (1 + N) at the G1 shape


def split3(x):
    """fp32 -> three bf16 planes [3, ...] (round to nearest even; the two subtractions are exact): the device's split3."""
    h = x.to(torch.bfloat16)
    r = x - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return torch.stack([h, m, l]).contiguous()


def _counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_p3_launch_counts(out, 1 if reset else 0)
    return list(out)


def _nt(gpu, M, N, K, bias=False, act=0, dref=False, seed=0, scale=1.0, check_ref=True, reps=1):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(seed)
    A = torch.randn(M, K, device=gpu, generator=g) * scale
    B = torch.randn(N, K, device=gpu, generator=g) * (K ** -0.5)          # pre-activations of unit scale (tanh not saturated everywhere)
    bias_t = torch.randn(N, device=gpu, generator=g) * scale if bias else None
    Y = torch.randn(M, N, device=gpu, generator=g) if dref else None
    Ap, Bp = split3(A), split3(B)
    Yh = Y.to(torch.bfloat16).contiguous() if dref else None
    C = torch.full((M, N), float('nan'), device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(reps):
        C.fill_(float('nan'))
        check(lib.cham_gemm_p3(ptr(Ap), M * K, K, ptr(Bp), N * K, K, 0, ptr(C), N, M, N, K, ptr(bias_t), act, ptr(Yh), N, 1 if dref else 0, 0,
                               None, 0, 1, st), "cham_gemm_p3")
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
        R = R * torch.where(Yh.double() > 0, 1.0, 0.2)
    return float((outs[0].double() - R).abs().max()) / max(scale, float(R.abs().max()))


def _tn(gpu, M, N, K, splits=1, accumulate=0, seed=0, reps=1):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(seed)
    A = torch.randn(K, M, device=gpu, generator=g)
    B = torch.randn(K, N, device=gpu, generator=g)
    C0 = torch.randn(M, N, device=gpu, generator=g)
    Ap, Bp = split3(A), split3(B)
    ws = torch.empty(32 << 20, dtype=torch.float32, device=gpu) if splits != 1 else None
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(reps):
        C = C0.clone() if accumulate else torch.full((M, N), float('nan'), device=gpu)
        check(lib.cham_gemm_p3(ptr(Ap), K * M, M, ptr(Bp), K * N, N, 1, ptr(C), N, M, N, K, None, 0, None, 0, 0, accumulate, ptr(ws),
                               ws.numel() * 4 if ws is not None else 0, splits, st), "cham_gemm_p3")
        torch.cuda.synchronize()
        outs.append(C)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "plane GEMM is not repeatable (race?)"
    R = A.double().t() @ B.double()
    if accumulate:
        R = R + C0.double()
    return float((outs[0].double() - R).abs().max()) / float(R.abs().max())


def test_split3_kernel_is_bit_exact_and_sums_back(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(7)
    X = torch.randn(300, 264, device=gpu, generator=g) * torch.logspace(-20, 20, 264, device=gpu)
    P = torch.zeros(3, 300, 264, dtype=torch.bfloat16, device=gpu)
    PT = torch.zeros(3, 264, 304, dtype=torch.bfloat16, device=gpu)
    check(lib.cham_split3(ptr(X), 300, 264, 264, ptr(P), 300 * 264, 264, ptr(PT), 264 * 304, 304, torch.cuda.current_stream().cuda_stream), "cham_split3")
    torch.cuda.synchronize()
    ref = split3(X)
    assert torch.equal(P.view(torch.int16), ref.view(torch.int16))
    assert torch.equal(PT[:, :, :300].view(torch.int16), ref.transpose(1, 2).contiguous().view(torch.int16))
    assert torch.equal(P[0].float() + P[1].float() + P[2].float(), X)          # the three planes carry all 24 significand bits


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (256, 256, 32), (256, 256, 48), (512, 512, 64), (300, 260, 96), (1000, 1024, 1024), (77, 520, 416),
                                   (1, 4, 16), (513, 256, 1024)])
def test_p3_nt(gpu, M, N, K):
    assert _nt(gpu, M, N, K) < 5e-5
    assert _nt(gpu, M, N, K, bias=True, act=2) < 5e-5
    assert _nt(gpu, M, N, K, bias=True) < 5e-5
    assert _nt(gpu, M, N, K, dref=True) < 5e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (256, 256, 40), (256, 512, 777), (512, 256, 3001), (1024, 1024, 5000), (256, 256, 1)])
def test_p3_tn_wgrad_splitk(gpu, M, N, K):
    assert _tn(gpu, M, N, K) < 1e-4
    assert _tn(gpu, M, N, K, splits=0) < 1e-4
    assert _tn(gpu, M, N, K, splits=7) < 1e-4
    assert _tn(gpu, M, N, K, splits=8) < 1e-4          # multiples of 8: one-K-split-per-XCD placement
    assert _tn(gpu, M, N, K, splits=0, accumulate=1) < 1e-4


def test_p3_exact_on_bf16_operands_and_layout(gpu):
    """Operands that ARE bf16 numbers have empty middle / low planes and integer-valued products are exact; A = I against an asymmetric
    B catches row / column swaps of the fragment maps, the swizzles and the C/D map, in both layouts."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    n = 512
    I = torch.eye(n, device=gpu)
    B = (torch.arange(n * n, dtype=torch.float32, device=gpu).reshape(n, n) % 97 - 31.0)
    Ip, Bp = split3(I), split3(B)
    st = torch.cuda.current_stream().cuda_stream
    C = torch.zeros(n, n, device=gpu)
    check(lib.cham_gemm_p3(ptr(Ip), n * n, n, ptr(Bp), n * n, n, 0, ptr(C), n, n, n, n, None, 0, None, 0, 0, 0, None, 0, 1, st), "nt")
    torch.cuda.synchronize()
    assert torch.equal(C, B.t())              # C = I B^T
    C.zero_()
    check(lib.cham_gemm_p3(ptr(Ip), n * n, n, ptr(Bp), n * n, n, 1, ptr(C), n, n, n, n, None, 0, None, 0, 0, 0, None, 0, 1, st), "tn")
    torch.cuda.synchronize()
    assert torch.equal(C, B)                  # C = I^T B
    C.zero_()
    check(lib.cham_gemm_p3(ptr(Bp), n * n, n, ptr(Ip), n * n, n, 1, ptr(C), n, n, n, n, None, 0, None, 0, 0, 0, None, 0, 1, st), "tn")
    torch.cuda.synchronize()
    assert torch.equal(C, B.t())              # C = B^T I


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e12, 1e30])
def test_p3_dynamic_range(gpu, scale):
    assert _nt(gpu, 300, 260, 96, scale=scale) < 5e-5


def test_p3_is_repeatable_under_load(gpu):
    """Race screen: the same launch five times, bit-identical (a fragment read that overtakes its DMA shows up as run-to-run noise)."""
    _nt(gpu, 4096, 1024, 1024, bias=True, act=2, check_ref=False, reps=5)
    assert _tn(gpu, 1024, 1024, 40000, splits=0, reps=5) < 1e-4


def test_p3_argument_errors(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import ptr
    lib = _lib.load()
    P = torch.zeros(3, 256, 256, dtype=torch.bfloat16, device=gpu)
    C = torch.zeros(256, 256, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    ok = lambda *a: lib.cham_gemm_p3(*a, st)
    assert ok(ptr(P), 65536, 256, ptr(P), 65536, 256, 0, ptr(C), 256, 256, 256, 24, None, 0, None, 0, 0, 0, None, 0, 1) < 0      # NT: K % 16
    assert ok(ptr(P), 65536, 256, ptr(P), 65536, 256, 1, ptr(C), 256, 250, 256, 256, None, 0, None, 0, 0, 0, None, 0, 1) < 0     # TN: M % 256
    assert ok(ptr(P), 65536, 250, ptr(P), 65536, 256, 0, ptr(C), 256, 256, 256, 16, None, 0, None, 0, 0, 0, None, 0, 1) < 0      # lda % 8
    assert ok(None, 65536, 256, ptr(P), 65536, 256, 0, ptr(C), 256, 256, 256, 16, None, 0, None, 0, 0, 0, None, 0, 1) < 0
    assert ok(ptr(P), 65536, 256, ptr(P), 65536, 256, 0, ptr(C), 256, 256, 256, 16, None, 1, None, 0, 0, 0, None, 0, 1) < 0      # act without bias


# ---- the benchmarked step's own shapes: error next to the native fp32 MFMA on the same operands ----------------------------------------
def _native(lib, A, lda, tA, B, ldb, tB, C, M, N, K, bias, act, dref, dact, ws, splits):
    from chameleon_recsys_amd._lib import check, ptr
    check(lib.cham_gemm_f32(ptr(A), lda, tA, ptr(B), ldb, tB, ptr(C), N, M, N, K, ptr(bias), act, ptr(dref), N, dact, None, 0, 1, 0, ptr(ws),
                            ws.numel() * 4 if ws is not None else 0, splits, torch.cuda.current_stream().cuda_stream), "native")
    torch.cuda.synchronize()


def test_p3_big_car_forward_and_dgrad(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    R, Cw = ROWS, 1024
    g = torch.Generator(device=gpu).manual_seed(1)
    A = torch.randn(R, Cw, device=gpu, generator=g)
    W = torch.randn(Cw, Cw, device=gpu, generator=g) * 0.03
    bias = torch.randn(Cw, device=gpu, generator=g)
    Y = torch.randn(R, Cw, device=gpu, generator=g)
    Ap = split3(A); Wp = split3(W); WTp = split3(W.t().contiguous()); Yh = Y.to(torch.bfloat16)
    rows = torch.arange(0, R, R // 2048, device=gpu)[:2048]
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(R, Cw, device=gpu)
    _counts(lib, reset=True)
    # forward: tanh(A W + b), B operand = planes of W^T
    check(lib.cham_gemm_p3(ptr(Ap), R * Cw, Cw, ptr(WTp), Cw * Cw, Cw, 0, ptr(out), Cw, R, Cw, Cw, ptr(bias), 2, None, 0, 0, 0, None, 0, 1, st), "fwd")
    torch.cuda.synchronize()
    ref = torch.tanh(A[rows].double() @ W.double() + bias.double())
    e_p3 = float((out[rows].double() - ref).abs().max())
    _native(lib, A, Cw, 0, W, Cw, 0, out, R, Cw, Cw, bias, 2, None, 0, None, 1)
    e_nat = float((out[rows].double() - ref).abs().max())
    assert e_p3 < 3e-4 and e_p3 < 1.5 * e_nat + 1e-7, (e_p3, e_nat)
    # dgrad: (A W^T) * leaky'(Y), B operand = planes of W as stored
    check(lib.cham_gemm_p3(ptr(Ap), R * Cw, Cw, ptr(Wp), Cw * Cw, Cw, 0, ptr(out), Cw, R, Cw, Cw, None, 0, ptr(Yh), Cw, 1, 0, None, 0, 1, st), "dgrad")
    torch.cuda.synchronize()
    ref = (A[rows].double() @ W.double().t()) * torch.where(Yh[rows].double() > 0, 1.0, 0.2)
    scale = float(ref.abs().max())
    e_p3 = float((out[rows].double() - ref).abs().max()) / scale
    Yf = Yh.float()
    _native(lib, A, Cw, 0, W, Cw, 1, out, R, Cw, Cw, None, 0, Yf, 1, None, 1)
    e_nat = float((out[rows].double() - ref).abs().max()) / scale
    assert e_p3 < 5e-5 and e_p3 < 1.5 * e_nat + 1e-7, (e_p3, e_nat)
    c = _counts(lib)
    assert c[0] == 2 and c[1] == 0, c


def test_p3_big_w2_wgrad_splitk(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    R, Cw = ROWS, 1024
    g = torch.Generator(device=gpu).manual_seed(4)
    A = torch.randn(R, Cw, device=gpu, generator=g)
    D = torch.randn(R, Cw, device=gpu, generator=g)
    Ap, Dp = split3(A), split3(D)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(Cw, Cw, device=gpu)
    ref = A.double().t() @ D.double()
    scale = float(ref.abs().max())
    _native(lib, A, Cw, 1, D, Cw, 0, out, Cw, Cw, R, None, 0, None, 0, ws, 0)
    e_nat = float((out.double() - ref).abs().max()) / scale
    for splits in (0, 8, 16, 12):
        out.fill_(float('nan'))
        check(lib.cham_gemm_p3(ptr(Ap), R * Cw, Cw, ptr(Dp), R * Cw, Cw, 1, ptr(out), Cw, Cw, Cw, R, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4,
                               splits, st), "wgrad")
        torch.cuda.synchronize()
        e_p3 = float((out.double() - ref).abs().max()) / scale
        assert e_p3 < 1e-4 and e_p3 < 1.5 * e_nat + 1e-7, (splits, e_p3, e_nat)


def test_infinite_operand_nan_in_plane_arithmetic_inf_in_native(gpu):
    """The documented semantic difference of the plane-product arithmetic (csrc/gemm_x3.hip header, --gemm_dtype help): splitting an
    infinite operand computes inf - inf, so the affected outputs are NaN where the native fp32 MFMA yields +-inf (or NaN only where it
    meets a zero).  Pinned for all three GEMM paths, and one level up: behind a tanh epilogue (the CAR layer) the native path SATURATES
    to a finite +-1 while both plane paths return NaN - a diverging run shows up as NaN in the default arithmetic, possibly as a finite
    loss in f32_native."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    M = N = K = 256
    g = torch.Generator(device=gpu).manual_seed(3)
    A = torch.rand(M, K, device=gpu, generator=g) + 0.5          # strictly positive: inf * b keeps a sign
    B = torch.rand(K, N, device=gpu, generator=g) + 0.5
    A[7, 3] = float('inf')
    bias = torch.zeros(N, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for name, fn in (("native", lib.cham_gemm_f32), ("x3", lib.cham_gemm_f32x3)):
        for act in (0, 2):
            C = torch.zeros(M, N, device=gpu)
            check(fn(ptr(A), K, 0, ptr(B), N, 0, ptr(C), N, M, N, K, ptr(bias) if act else None, act, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), name)
            outs[(name, act)] = C
    Ap, BTp = split3(A), split3(B.t().contiguous())
    for act in (0, 2):
        C = torch.zeros(M, N, device=gpu)
        check(lib.cham_gemm_p3(ptr(Ap), M * K, K, ptr(BTp), N * K, K, 0, ptr(C), N, M, N, K, ptr(bias) if act else None, act, None, 0, 0, 0, None, 0, 1, st), "p3")
        outs[("p3", act)] = C
    torch.cuda.synchronize()
    other = torch.ones(M, dtype=torch.bool, device=gpu); other[7] = False
    for k, C in outs.items():
        assert torch.isfinite(C[other]).all(), k                 # only the row that holds the inf is affected
    assert torch.isinf(outs[("native", 0)][7]).all() and (outs[("native", 0)][7] > 0).all()
    assert (outs[("native", 2)][7] == 1.0).all()                 # tanh(+inf) = 1: finite
    for name in ("x3", "p3"):
        assert torch.isnan(outs[(name, 0)][7]).all() and torch.isnan(outs[(name, 2)][7]).all(), name
