"""bf16-resident MFMA GEMMs (csrc/gemm_b16.hip) against float64 on the bf16 operand values: NT (forward / dgrad, bias + activation
or x activation'(saved bf16 output), bf16 or fp32 output) and TN (wgrad: both operands free-index-contiguous, fragments through the LDS
transpose read; split-K) on every tile instance, ragged M, asymmetric data (a swapped C^T accumulator or a mis-gathered transpose
read is an O(1) error)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(gpu, M, N, K, mode, bias=False, act=0, dref=False, dact=0, out_f32=False, accumulate=0, splits=1, variant=-1, seed=0):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    nt = mode == 'nt'
    A = torch.randn((M, K) if nt else (K, M), generator=g).bfloat16()
    B = torch.randn((N, K) if nt else (K, N), generator=g).bfloat16()
    bias_t = torch.randn(N, generator=g) if bias else None
    ref_t = torch.tanh(torch.randn(M, N, generator=g)).bfloat16() if dref else None
    C0 = torch.randn(M, N, generator=g)
    R = (A.double() @ B.double().t()) if nt else (A.double().t() @ B.double())
    if bias:
        R = R + bias_t.double()
    if act == 1:
        R = torch.where(R > 0, R, 0.2 * R)
    elif act == 2:
        R = torch.tanh(R)
    if dref:
        y = ref_t.double()
        R = R * (torch.where(y > 0, 1.0, 0.2) if dact == 1 else (1 - y * y))
    if accumulate:
        R = R + C0.double()
    dA, dB = A.to(gpu), B.to(gpu)
    dC = (C0.clone() if out_f32 else torch.full((M, N), float('nan')).bfloat16()).to(gpu)
    dbias = bias_t.to(gpu) if bias else None
    dref_t = ref_t.to(gpu) if dref else None
    ws = torch.empty(32 << 20, dtype=torch.float32, device=gpu) if splits != 1 else None
    lib.cham_gemm_b16_set_variant(variant)
    try:
        rc = lib.cham_gemm_b16(ptr(dA), A.shape[1], 0 if nt else 1, ptr(dB), B.shape[1], 1 if nt else 0, ptr(dC), N, int(out_f32),
                               M, N, K, ptr(dbias), act, ptr(dref_t), N, dact, accumulate, ptr(ws),
                               (32 << 20) * 4 if ws is not None else 0, splits, torch.cuda.current_stream().cuda_stream)
    finally:
        lib.cham_gemm_b16_set_variant(-1)
    check(rc, "cham_gemm_b16")
    torch.cuda.synchronize()
    out = dC.cpu().double()
    scale = max(1.0, float(R.abs().max()))
    err = float((out - R).abs().max()) / scale
    return err


BF16_OUT = 2.0 ** -8          # one bf16 rounding of the output (relative to the largest entry)


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (1000, 1024, 128), (77, 64, 128), (513, 32, 64), (259, 256, 1024), (64, 128, 32)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_nt_forward_and_dgrad(gpu, M, N, K, variant):
    assert _run(gpu, M, N, K, 'nt', out_f32=True, variant=variant) < 5e-5
    assert _run(gpu, M, N, K, 'nt', variant=variant) < BF16_OUT
    assert _run(gpu, M, N, K, 'nt', bias=True, act=1, variant=variant) < BF16_OUT
    assert _run(gpu, M, N, K, 'nt', bias=True, act=2, variant=variant) < BF16_OUT
    assert _run(gpu, M, N, K, 'nt', dref=True, dact=1, variant=variant) < BF16_OUT


def test_tn_wgrad_many_splits_small_output(gpu):
    """The scorer's weight-gradient shape class: a small output with tens of K-splits goes through the wide fixed-order reduction
    (16 threads per column group), with and without accumulate, repeatably."""
    for M, N, K, splits in ((1024, 128, 40000, 0), (1024, 128, 40000, 64), (128, 64, 60000, 0), (64, 32, 33000, 40)):
        a = _run(gpu, M, N, K, 'tn', out_f32=True, splits=splits)
        b = _run(gpu, M, N, K, 'tn', out_f32=True, splits=splits)
        assert a < 1e-4 and a == b, (M, N, K, splits, a, b)
        assert _run(gpu, M, N, K, 'tn', out_f32=True, splits=splits, accumulate=1) < 1e-4


@pytest.mark.parametrize("M,N,K", [(128, 128, 5000), (1024, 128, 3001), (64, 32, 20000), (1024, 1024, 4099), (128, 64, 777)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_tn_wgrad_splitk(gpu, M, N, K, variant):
    assert _run(gpu, M, N, K, 'tn', out_f32=True, variant=variant) < 1e-4
    assert _run(gpu, M, N, K, 'tn', out_f32=True, splits=0, variant=variant) < 1e-4
    assert _run(gpu, M, N, K, 'tn', out_f32=True, splits=16, accumulate=1, variant=variant) < 1e-4
    assert _run(gpu, M, N, K, 'tn', out_f32=True, splits=5, variant=variant) < 1e-4


def test_asymmetric_identity(gpu):
    """A = I against an asymmetric B: catches a transposed accumulator map or a wrong transpose-read gather exactly."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    n = 160
    I = torch.eye(n).bfloat16().to(gpu)
    Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97 - 31.0).bfloat16()      # small integers: exact in bf16
    dB = Bm.to(gpu)
    s = torch.cuda.current_stream().cuda_stream
    C = torch.zeros(n, n, device=gpu)
    check(lib.cham_gemm_b16(ptr(I), n, 0, ptr(dB), n, 1, ptr(C), n, 1, n, n, n, None, 0, None, 0, 0, 0, None, 0, 1, s), "nt")   # I B^T
    torch.cuda.synchronize()
    assert torch.equal(C.cpu(), Bm.float().t())
    check(lib.cham_gemm_b16(ptr(I), n, 1, ptr(dB), n, 0, ptr(C), n, 1, n, n, n, None, 0, None, 0, 0, 0, None, 0, 1, s), "tn")   # I^T B
    torch.cuda.synchronize()
    assert torch.equal(C.cpu(), Bm.float())
    check(lib.cham_gemm_b16(ptr(dB), n, 1, ptr(I), n, 0, ptr(C), n, 1, n, n, n, None, 0, None, 0, 0, 0, None, 0, 1, s), "tn")   # B^T I
    torch.cuda.synchronize()
    assert torch.equal(C.cpu(), Bm.float().t())


ROWS = 72 * 19 * 51


@pytest.mark.parametrize("variant", [1, 2])
def test_car_shapes(gpu, variant):
    """The three CAR layer-2 GEMMs and the scorer layer-1 trio at the step's own shapes (72-session G1 batch)."""
    assert _run(gpu, ROWS, 1024, 1024, 'nt', bias=True, act=2, variant=variant, seed=1) < BF16_OUT
    assert _run(gpu, ROWS, 1024, 1024, 'nt', dref=True, dact=1, variant=variant, seed=2) < BF16_OUT
    assert _run(gpu, 1024, 1024, ROWS, 'tn', out_f32=True, splits=0, variant=variant, seed=3) < 1e-4
    assert _run(gpu, ROWS, 128, 1024, 'nt', bias=True, act=1, variant=variant, seed=4) < BF16_OUT
    assert _run(gpu, ROWS, 1024, 128, 'nt', variant=variant, seed=5) < BF16_OUT
    assert _run(gpu, 1024, 128, ROWS, 'tn', out_f32=True, splits=0, variant=variant, seed=6) < 1e-4
