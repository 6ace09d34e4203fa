"""The bf16 configuration's CAR GEMMs on the LDS-DMA core (csrc/gemm_p3.hip, gemm_b1_kernel: one bf16 plane per operand, three
consecutive 16-k chunks per stage) against a float64 reference of the SAME bf16 operands (products exact, fp32 accumulate, bf16
output rounding) and against cham_gemm_b16 on the step's shapes: both layouts, every epilogue, ragged rows / columns, K tails that
end inside a 48-k stage (zero-sized chunk descriptors), one- to three-stage reductions (pipeline prologue), split-K placements,
repeatability."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = 248064          # B * T * (1 + N) at the G1 shape


def _lib():
    from chameleon_recsys_amd import _lib
    return _lib.load()


def _counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_p3_launch_counts(out, 1 if reset else 0)
    return list(out)


def _nt(gpu, M, N, K, bias=False, act=0, dref=False, seed=0, reps=1, ragged_ld=0):
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib()
    g = torch.Generator(device=gpu).manual_seed(seed)
    lda = K + ragged_ld
    Af = torch.randn(M, lda, device=gpu, generator=g)
    Bf = torch.randn(N, lda, device=gpu, generator=g) * (K ** -0.5)
    A, B = Af.bfloat16().contiguous(), Bf.bfloat16().contiguous()
    bias_t = torch.randn(N, device=gpu, generator=g) if bias else None
    Y = (torch.randn(M, N, device=gpu, generator=g)).bfloat16().contiguous() if dref else None
    C = torch.full((M, N), float('nan'), device=gpu, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    c0 = _counts(lib)
    for _ in range(reps):
        C.fill_(float('nan'))
        check(lib.cham_gemm_b16_dma(ptr(A), lda, ptr(B), lda, 0, ptr(C), N, M, N, K, ptr(bias_t), act, ptr(Y), N, 1 if dref else 0, 0,
                                    None, 0, 1, st), "cham_gemm_b16_dma")
        torch.cuda.synchronize()
        outs.append(C.clone())
    assert _counts(lib)[2] == c0[2] + reps
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "not repeatable (race?)"
    R = A[:, :K].double() @ B[:, :K].double().t()
    if bias:
        R = R + bias_t.double()
    if act == 2:
        R = torch.tanh(R)
    if dref:
        R = R * torch.where(Y.double() > 0, 1.0, 0.2)
    got = outs[0].double()
    assert torch.isfinite(got).all()
    # bf16 output: half an ulp of the result (2^-9 relative) + the fp32 accumulation error
    err = (got - R).abs()
    tol = R.abs() * 2.0 ** -8 + 1e-5 * max(1.0, float(R.abs().max()))
    assert bool((err <= tol).all()), (float(err.max()), float((err / tol).max()))
    return outs[0]


@pytest.mark.parametrize("M,N,K", [(256, 256, 48), (256, 256, 16), (256, 256, 32), (256, 256, 96), (256, 256, 144), (512, 256, 1024),
                                   (300, 260, 64), (1, 4, 16), (777, 1024, 1040), (257, 252, 208)])
def test_nt_plain_shapes(gpu, M, N, K):
    _nt(gpu, M, N, K, seed=M + K)


def test_nt_epilogues_and_repeatability(gpu):
    _nt(gpu, 1000, 512, 1024, bias=True, act=2, seed=1, reps=3)
    _nt(gpu, 1000, 512, 1024, dref=True, seed=2, reps=3)
    _nt(gpu, 515, 1024, 400, bias=True, act=2, seed=3)
    _nt(gpu, 515, 1024, 400, dref=True, seed=4)


def test_nt_leading_dimension_larger_than_k(gpu):
    """K ends inside a row (lda > K): the chunks past K must not pick up the row's own tail columns."""
    _nt(gpu, 300, 256, 80, seed=5, ragged_ld=48)
    _nt(gpu, 300, 256, 1024, seed=6, ragged_ld=8)


def _tn(gpu, M, N, K, splits=0, accumulate=False, seed=0, reps=1):
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib()
    g = torch.Generator(device=gpu).manual_seed(seed)
    A = (torch.randn(K, M, device=gpu, generator=g)).bfloat16().contiguous()
    B = (torch.randn(K, N, device=gpu, generator=g) * (K ** -0.5)).bfloat16().contiguous()
    C0 = torch.randn(M, N, device=gpu, generator=g)
    ws = torch.empty(64 * M * N, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(reps):
        C = C0.clone() if accumulate else torch.full((M, N), float('nan'), device=gpu)
        check(lib.cham_gemm_b16_dma(ptr(A), M, ptr(B), N, 1, ptr(C), N, M, N, K, None, 0, None, 0, 0, 1 if accumulate else 0, ptr(ws),
                                    ws.numel() * 4, splits, st), "cham_gemm_b16_dma")
        torch.cuda.synchronize()
        outs.append(C.clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "not repeatable (race?)"
    R = A.double().t() @ B.double()
    if accumulate:
        R = R + C0.double()
    err = float((outs[0].double() - R).abs().max())
    assert err < 2e-5 * max(1.0, float(R.abs().max())), err
    return _counts(lib)


@pytest.mark.parametrize("K,splits", [(48, 1), (50, 1), (16, 1), (100, 1), (5000, 1), (5000, 0), (20011, 0), (20000, 8), (20000, 5), (3100, 2)])
def test_tn_shapes_and_splits(gpu, K, splits):
    c = _tn(gpu, 256, 512, K, splits=splits, seed=K)
    if splits > 1:
        assert c[7] <= splits


def test_tn_accumulate_and_repeatability(gpu):
    _tn(gpu, 512, 256, 7000, splits=0, accumulate=True, seed=3, reps=3)
    _tn(gpu, 256, 256, 300, splits=1, accumulate=True, seed=4)


def test_rejects_what_it_does_not_take(gpu):
    from chameleon_recsys_amd._lib import ptr
    lib = _lib()
    A = torch.zeros(256, 256, device=gpu, dtype=torch.bfloat16)
    C = torch.zeros(256, 256, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.cham_gemm_b16_dma(ptr(A), 256, ptr(A), 256, 0, ptr(C), 256, 256, 256, 24, None, 0, None, 0, 0, 0, None, 0, 1, st) < 0      # K % 16
    assert lib.cham_gemm_b16_dma(ptr(A), 256, ptr(A), 256, 1, ptr(C), 256, 128, 256, 256, None, 0, None, 0, 0, 0, None, 0, 1, st) < 0     # TN M % 256
    assert lib.cham_gemm_b16_dma(ptr(A), 256, ptr(A), 256, 0, ptr(C), 256, 256, 256, 256, None, 1, None, 0, 0, 0, None, 0, 1, st) < 0     # leaky fwd


@pytest.mark.parametrize("which", ["fwd", "dgrad", "wgrad"])
def test_car_shapes_match_the_register_staged_kernels(gpu, which):
    """At the G1 CAR shapes: same bf16 operands through cham_gemm_b16 (register-staged) and through the LDS-DMA core.  Both accumulate
    exact bf16 products in fp32 (different association): fp32 outputs agree to accumulation rounding, bf16 outputs to one bf16 ulp
    on the few elements whose fp32 value sits at a rounding boundary."""
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib()
    g = torch.Generator(device=gpu).manual_seed(9)
    st = torch.cuda.current_stream().cuda_stream
    C_ = 1024
    W = (torch.randn(C_, C_, device=gpu, generator=g) * C_ ** -0.5).bfloat16().contiguous()
    X = torch.randn(ROWS, C_, device=gpu, generator=g).bfloat16().contiguous()
    ws = torch.empty(64 << 20, device=gpu)
    if which == "wgrad":
        D = (torch.randn(ROWS, C_, device=gpu, generator=g) * ROWS ** -0.5).bfloat16().contiguous()
        a, b = torch.empty(C_, C_, device=gpu), torch.empty(C_, C_, device=gpu)
        check(lib.cham_gemm_b16(ptr(X), C_, 1, ptr(D), C_, 0, ptr(a), C_, 1, C_, C_, ROWS, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4, 0, st), "b16")
        check(lib.cham_gemm_b16_dma(ptr(X), C_, ptr(D), C_, 1, ptr(b), C_, C_, C_, ROWS, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4, 0, st), "dma")
        torch.cuda.synchronize()
        assert float((a - b).abs().max()) < 2e-5 * float(a.abs().max())
        return
    bias = torch.randn(C_, device=gpu, generator=g) if which == "fwd" else None
    Y = torch.randn(ROWS, C_, device=gpu, generator=g).bfloat16().contiguous() if which == "dgrad" else None
    a = torch.empty(ROWS, C_, device=gpu, dtype=torch.bfloat16)
    b = torch.empty(ROWS, C_, device=gpu, dtype=torch.bfloat16)
    act, dact = (2, 0) if which == "fwd" else (0, 1)
    check(lib.cham_gemm_b16(ptr(X), C_, 0, ptr(W), C_, 1, ptr(a), C_, 0, ROWS, C_, C_, ptr(bias), act, ptr(Y), C_, dact, 0, None, 0, 1, st), "b16")
    check(lib.cham_gemm_b16_dma(ptr(X), C_, ptr(W), C_, 0, ptr(b), C_, ROWS, C_, C_, ptr(bias), act, ptr(Y), C_, dact, 0, None, 0, 1, st), "dma")
    torch.cuda.synchronize()
    d = (a.float() - b.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(a.float().abs().max())          # at most one bf16 ulp apart
    assert float((d > 0).float().mean()) < 2e-3                                # and only on boundary cases


@pytest.mark.parametrize("M,N,K,kw", [(512, 256, 1024, {}), (1000, 512, 1024, dict(bias=True, act=2)), (1000, 512, 1024, dict(dref=True)), (300, 260, 64, {}),
                                      (777, 1024, 128, dict(bias=True, act=2)), (256, 256, 192, dict(dref=True)), (2048, 1024, 1024, dict(ragged_ld=64))])
def test_nt_wide_pieces_kernel_against_the_narrow_one(gpu, M, N, K, kw):
    """gemm_b1w_kernel (round 6: 64-byte source pieces, K % 64 == 0) and gemm_b1_kernel (32-byte pieces, 48-k stages) on the same operands: each
    within the float64 tolerance of _nt(), and within one bf16 ulp of each other (another grouping of the K loop: fp32 sums differ in the last
    bits, the bf16 rounding of the output may then differ by one ulp); launch counter [4] proves which kernel ran."""
    lib = _lib()
    assert lib.cham_gemm_b16_dma_set_nt_wide(1) in (0, 1)
    try:
        c0 = _counts(lib)
        wide = _nt(gpu, M, N, K, seed=K + M, **kw)
        assert _counts(lib)[4] == c0[4] + 1
        lib.cham_gemm_b16_dma_set_nt_wide(0)
        c0 = _counts(lib)
        narrow = _nt(gpu, M, N, K, seed=K + M, **kw)
        assert _counts(lib)[4] == c0[4]
    finally:
        lib.cham_gemm_b16_dma_set_nt_wide(1)
    d = (wide.float() - narrow.float()).abs()
    assert bool((d <= narrow.float().abs() * 2.0 ** -7 + 1e-6).all()), float(d.max())
    assert float((d > 0).float().mean()) < 0.05          # ... and on few entries


def test_nt_wide_kernel_is_not_taken_when_k_is_no_multiple_of_64(gpu):
    lib = _lib()
    c0 = _counts(lib)
    _nt(gpu, 256, 256, 1040, seed=3)
    _nt(gpu, 256, 256, 48, seed=4)
    assert _counts(lib)[4] == c0[4]
