"""fp32 GEMM on the bf16 matrix cores (csrc/gemm_x3.hip: three bf16 planes per operand, six plane products, fp32 accumulate) against a
float64 reference - the same shapes, transpose modes, epilogues and tolerances as the native fp32 MFMA tests (test_gemm_gpu.py), and,
on the benchmarked step's own shapes, its error next to the native path's on the same operands: the claim is fp32-GRADE error, not a
reduced-precision mode, so the test fails if the split path is measurably worse than v_mfma_f32_32x32x2_f32."""
import ctypes

import pytest
import torch

from tests.test_gemm_gpu import ROWS, _BigCase, _run

pytestmark = pytest.mark.gpu


def _x3_counts(lib, reset=False):
    out = (ctypes.c_longlong * 8)()
    lib.cham_gemm_f32x3_launch_counts(out, 1 if reset else 0)
    return list(out)


@pytest.fixture(params=[-1, 0, 2], ids=["auto", "128x128", "256x128"])
def variant(request):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    lib.cham_gemm_f32x3_set_variant(request.param)
    yield request.param
    lib.cham_gemm_f32x3_set_variant(-1)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 260, 96), (77, 1024, 408), (1000, 64, 128), (513, 32, 64), (64, 72, 1024), (500, 128, 16),
                                   (130, 132, 4)])
def test_x3_nn(gpu, variant, M, N, K):
    assert _run(gpu, M, N, K, x3=True) < 5e-5
    assert _run(gpu, M, N, K, bias=True, act=1, x3=True) < 5e-5
    assert _run(gpu, M, N, K, bias=True, act=2, x3=True) < 5e-5
    assert _run(gpu, M, N, K, bias=True, x3=True) < 5e-5


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (259, 72, 1024), (1000, 408, 128), (123, 1024, 512), (400, 64, 32), (257, 136, 20)])
def test_x3_nt_dgrad(gpu, variant, M, N, K):
    assert _run(gpu, M, N, K, transB=1, x3=True) < 5e-5
    assert _run(gpu, M, N, K, transB=1, dref=True, dact=1, x3=True) < 5e-5
    assert _run(gpu, M, N, K, transB=1, dref=True, dact=2, accumulate=1, x3=True) < 5e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 5000), (72, 1024, 777), (408, 128, 3001), (64, 32, 20000), (1024, 128, 4099)])
def test_x3_tn_wgrad_splitk(gpu, variant, M, N, K):
    assert _run(gpu, M, N, K, transA=1, x3=True) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=0, x3=True) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=7, x3=True) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=8, x3=True) < 1e-4          # multiples of 8: one-K-split-per-XCD placement
    assert _run(gpu, M, N, K, transA=1, accumulate=1, x3=True) < 1e-4


def test_x3_rowscale(gpu, variant):
    assert _run(gpu, 51 * 40, 128, 256, rowscale=51, bias=True, act=1, x3=True) < 5e-5          # scorer layer 1
    assert _run(gpu, 256, 128, 51 * 40, transA=1, rowscale=51, splits=0, x3=True) < 5e-5         # its wgrad
    assert _run(gpu, 256, 128, 51 * 40, transA=1, rowscale=51, x3=True) < 5e-5                   # too short to split


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e12, 1e30])
def test_x3_dynamic_range(gpu, scale):
    """bf16 shares fp32's exponent range: the three planes neither overflow nor flush for operands far from 1 (the lowest plane of
    a 1e-30 operand is ~4e-36, still a normal number)."""
    assert _run(gpu, 300, 260, 96, x3=True, scale=scale) < 5e-5
    assert _run(gpu, 300, 128, 64, transB=1, x3=True, scale=scale) < 5e-5


def test_x3_exact_on_bf16_representable_operands_and_layout(gpu):
    """Operands that ARE bf16 numbers have empty middle / low planes: integer-valued products are exact.  A = I against an asymmetric
    B also catches row / column swaps of the fragment and C/D maps."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    n = 160
    A = torch.eye(n).to(gpu)
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97 - 31.0).to(gpu)
    for tA, tB in ((0, 0), (0, 1), (1, 0)):
        C = torch.zeros(n, n, device=gpu)
        check(lib.cham_gemm_f32x3(ptr(A), n, tA, ptr(B), n, tB, ptr(C), n, n, n, n, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1,
                                  torch.cuda.current_stream().cuda_stream), "gemm")
        torch.cuda.synchronize()
        assert torch.equal(C, B.t() if tB else B), (tA, tB)


def test_x3_narrow_outputs_delegate_to_native(gpu):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    _x3_counts(lib, reset=True)
    assert _run(gpu, 1000, 64, 128, x3=True) < 5e-5
    assert _run(gpu, 513, 32, 64, x3=True) < 5e-5
    c = _x3_counts(lib)
    assert c[3] == 2 and c[0] == c[1] == 0, c


def test_x3_argument_errors(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import ptr
    lib = _lib.load()
    A = torch.zeros(128, 128, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda *a: lib.cham_gemm_f32x3(*a, st)
    assert call(None, 128, 0, ptr(A), 128, 0, ptr(A), 128, 128, 128, 128, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1) < 0
    assert call(ptr(A), 126, 0, ptr(A), 128, 0, ptr(A), 128, 128, 128, 128, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1) < 0      # lda % 4
    assert call(ptr(A), 128, 1, ptr(A), 128, 1, ptr(A), 128, 128, 128, 128, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1) < 0      # TT
    assert call(ptr(A), 128, 0, ptr(A), 128, 0, ptr(A), 128, 128, 128, 128, None, 1, None, 0, 0, None, 0, 1, 0, None, 0, 1) < 0      # act without bias


# ---- the benchmarked step's own shapes: error next to the native fp32 MFMA path on the same operands -------------------------------------
def _both(lib, case, tol, splits=1):
    e_native = case.run(lib, -1, splits=splits)
    e_auto = case.run(lib, -1, splits=splits, x3=True)
    e_small = case.run(lib, 0, splits=splits, x3=True)
    assert e_auto < tol and e_small < tol, (e_native, e_auto, e_small)
    # fp32-grade: not measurably worse than the native fp32 matrix instruction (both are dominated by fp32 accumulation rounding)
    assert e_auto < 1.5 * e_native + 1e-7 and e_small < 1.5 * e_native + 1e-7, (e_native, e_auto, e_small)
    return e_native, e_auto


def test_x3_big_car_forward(gpu):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    _x3_counts(lib, reset=True)
    _both(lib, _BigCase(gpu, ROWS, 1024, 1024, bias=True, act=2, seed=1), 3e-4)
    c = _x3_counts(lib)
    assert c[1] == 1 and c[0] == 1, c          # automatic choice = 256x128


def test_x3_big_car_dgrad_and_scorer_dgrad(gpu):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    _both(lib, _BigCase(gpu, ROWS, 1024, 1024, transB=1, dref=True, dact=1, seed=2), 5e-5)
    _both(lib, _BigCase(gpu, ROWS, 1024, 128, transB=1, dref=True, dact=2, seed=3), 5e-5)


def test_x3_big_w2_wgrad_splitk(gpu):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    case = _BigCase(gpu, 1024, 1024, ROWS, transA=1, seed=4)
    for splits in (0, 8, 16, 12):
        _both(lib, case, 1e-4, splits=splits)


def test_x3_big_rowscale_scorer_layer1(gpu):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    _both(lib, _BigCase(gpu, ROWS, 128, 1024, bias=True, act=1, rowscale=51, seed=5), 5e-5)
    wg = _BigCase(gpu, 1024, 128, ROWS, transA=1, rowscale=51, seed=6)
    for splits in (0, 16):
        _both(lib, wg, 1e-4, splits=splits)
