"""fp32 and bf16 MFMA GEMM (csrc/gemm.hip) against a float64 reference: every transpose mode and epilogue at small ragged
shapes (128x128 / 256x64 / 256x32 tiles), and - at the end of the file - the benchmarked step's own shapes on the 256x128 / 256x256
tiles (automatic choice and forced through cham_gemm_set_variant)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(gpu, M, N, K, transA=0, transB=0, bias=False, act=0, dref=False, dact=0, rowscale=0, accumulate=0, splits=1, seed=0, bf16=False,
         x3=False, scale=1.0):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((K, M) if transA else (M, K), generator=g) * scale
    B = torch.randn((N, K) if transB else (K, N), generator=g)
    C0 = torch.randn(M, N, generator=g) * scale
    bias_t = torch.randn(N, generator=g) * scale if bias else None
    ref_t = torch.randn(M, N, generator=g) if dref else None
    rs_div = rowscale if rowscale else 1
    rs_rows = (A.shape[0] + rs_div - 1) // rs_div
    rs_t = torch.randn(rs_rows, A.shape[1], generator=g) if rowscale else None
    # fp64 reference
    Ae = A.double()
    if rowscale:
        Ae = Ae * rs_t.double()[torch.arange(A.shape[0]) // rs_div]
    Be = B.double()
    if bf16:      # operands rounded to bf16 (after the fp32 row-scale product), exact products, fp32-class accumulation
        Ae = Ae.float().bfloat16().double()
        Be = B.bfloat16().double()
    opA = Ae.t() if transA else Ae
    opB = Be.t() if transB else Be
    R = opA @ opB
    if bias:
        R = R + bias_t.double()
    if act == 1:
        R = torch.where(R > 0, R, 0.2 * R)
    elif act == 2:
        R = torch.tanh(R)
    if dref:
        y = ref_t.double()
        R = R * (torch.where(y > 0, 1.0, 0.2) if dact == 1 else (1 - y * y) if dact == 2 else 1.0)
    if accumulate:
        R = R + C0.double()
    dA, dB, dC = A.to(gpu), B.to(gpu), C0.clone().to(gpu)
    dbias = bias_t.to(gpu) if bias else None
    dref_t = ref_t.to(gpu) if dref else None
    drs = rs_t.to(gpu) if rowscale else None
    ws = torch.empty(32 << 20, dtype=torch.float32, device=gpu) if splits != 1 else None
    rc = (lib.cham_gemm_bf16 if bf16 else (lib.cham_gemm_f32x3 if x3 else lib.cham_gemm_f32))(ptr(dA), A.shape[1], transA, ptr(dB), B.shape[1], transB, ptr(dC), N, M, N, K, ptr(dbias), act,
                           ptr(dref_t), N, dact, ptr(drs), A.shape[1], rs_div, accumulate, ptr(ws),
                           (32 << 20) * 4 if ws is not None else 0, splits, torch.cuda.current_stream().cuda_stream)
    check(rc, "gemm")
    torch.cuda.synchronize()
    out = dC.cpu().double()
    err = float((out - R).abs().max()) / max(scale, float(R.abs().max()))
    return err


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 260, 96), (77, 1024, 408), (1000, 64, 128), (513, 32, 64), (64, 72, 1024)])
def test_nn(gpu, M, N, K):
    assert _run(gpu, M, N, K) < 5e-5
    assert _run(gpu, M, N, K, bias=True, act=1) < 5e-5
    assert _run(gpu, M, N, K, bias=True, act=2) < 5e-5


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (259, 72, 1024), (1000, 408, 128), (123, 1024, 512), (400, 64, 32)])
def test_nt_dgrad(gpu, M, N, K):
    assert _run(gpu, M, N, K, transB=1) < 5e-5
    assert _run(gpu, M, N, K, transB=1, dref=True, dact=1) < 5e-5
    assert _run(gpu, M, N, K, transB=1, dref=True, dact=2, accumulate=1) < 5e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 5000), (72, 1024, 777), (408, 128, 3001), (64, 32, 20000), (1024, 128, 4099)])
def test_tn_wgrad_splitk(gpu, M, N, K):
    assert _run(gpu, M, N, K, transA=1) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=0) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=7) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=8) < 1e-4          # multiples of 8: one-K-split-per-XCD placement
    assert _run(gpu, M, N, K, transA=1, splits=16) < 1e-4


def test_rowscale(gpu):
    assert _run(gpu, 51 * 40, 128, 256, rowscale=51, bias=True, act=1) < 5e-5          # scorer layer 1
    assert _run(gpu, 256, 128, 51 * 40, transA=1, rowscale=51, splits=0) < 5e-5         # its wgrad


def test_asymmetric_layout(gpu):
    """A = I against an asymmetric B catches row/col swaps of the MFMA C/D map."""
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    n = 160
    A = torch.eye(n).to(gpu)
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97 - 31.0).to(gpu)
    C = torch.zeros(n, n, device=gpu)
    check(lib.cham_gemm_f32(ptr(A), n, 0, ptr(B), n, 0, ptr(C), n, n, n, n, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1,
                            torch.cuda.current_stream().cuda_stream), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(C, B)


# ---- bf16-compute variant (BASELINE config 3): same contract, operands rounded to bf16 on the fly -------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 260, 96), (77, 1024, 408), (1000, 64, 128), (513, 32, 64), (64, 72, 1024)])
def test_bf16_nn(gpu, M, N, K):
    assert _run(gpu, M, N, K, bf16=True) < 5e-5
    assert _run(gpu, M, N, K, bias=True, act=1, bf16=True) < 5e-5
    assert _run(gpu, M, N, K, bias=True, act=2, bf16=True) < 5e-5
    assert _run(gpu, M, N, K, bias=True, bf16=True) < 5e-5


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (259, 72, 1024), (1000, 408, 128), (123, 1024, 512), (400, 64, 32)])
def test_bf16_nt_dgrad(gpu, M, N, K):
    assert _run(gpu, M, N, K, transB=1, bf16=True) < 5e-5
    assert _run(gpu, M, N, K, transB=1, dref=True, dact=1, bf16=True) < 5e-5
    assert _run(gpu, M, N, K, transB=1, dref=True, dact=2, accumulate=1, bf16=True) < 5e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 5000), (72, 1024, 777), (408, 128, 3001), (64, 32, 20000), (1024, 128, 4099)])
def test_bf16_tn_wgrad_splitk(gpu, M, N, K):
    assert _run(gpu, M, N, K, transA=1, bf16=True) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=0, bf16=True) < 1e-4
    assert _run(gpu, M, N, K, transA=1, splits=7, bf16=True) < 1e-4


def test_bf16_rowscale(gpu):
    assert _run(gpu, 51 * 40, 128, 256, rowscale=51, bias=True, act=1, bf16=True) < 5e-5
    assert _run(gpu, 256, 128, 51 * 40, transA=1, rowscale=51, splits=0, bf16=True) < 5e-5


# ---- the instances the benchmarked step actually runs on ---------------------------------------------------------
# launch_by_shape sends a GEMM to the 256x128 tile when M*N >= 2^20 and the grid has >= 256 workgroups, to 256x256 when
# additionally (NT or TN) K >= 512, M >= 1024, N >= 512.  The cases below are the G1-shape step's GEMMs at 72 sessions (69 768
# candidate rows x 1024) - each checked against float64 on the instance picked automatically AND with the tile forced through
# cham_gemm_set_variant (2 = 256x128, 4 = 256x256, 0 = 128x128), fp32 and bf16, with the launch counters proving which ran.
def _counts(lib, reset=False):
    import ctypes
    out = (ctypes.c_longlong * 16)()
    lib.cham_gemm_launch_counts(out, int(reset))
    return list(out)


class _BigCase:
    """One GEMM problem + its float64 references (fp32 operands / bf16-rounded operands), shared by all tile variants."""

    def __init__(self, gpu, M, N, K, transA=0, transB=0, bias=False, act=0, dref=False, dact=0, rowscale=0, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.M, self.N, self.K, self.tA, self.tB, self.act, self.dact, self.rs_div = M, N, K, transA, transB, act, dact, (rowscale or 1)
        self.A = torch.randn((K, M) if transA else (M, K), generator=g)
        self.B = torch.randn((N, K) if transB else (K, N), generator=g)
        self.bias = torch.randn(N, generator=g) if bias else None
        self.ref_t = torch.tanh(torch.randn(M, N, generator=g)) if dref else None
        self.rs = torch.randn((self.A.shape[0] + self.rs_div - 1) // self.rs_div, self.A.shape[1], generator=g) if rowscale else None
        self.dev = {k: (v.to(gpu) if v is not None else None) for k, v in
                    dict(A=self.A, B=self.B, bias=self.bias, ref=self.ref_t, rs=self.rs).items()}
        self.gpu = gpu
        self._refs = {}

    def reference(self, bf16):
        if bf16 in self._refs:
            return self._refs[bf16]
        Ae = self.A.double()
        if self.rs is not None:
            Ae = Ae * self.rs.double()[torch.arange(self.A.shape[0]) // self.rs_div]
        Be = self.B.double()
        if bf16:
            Ae, Be = Ae.float().bfloat16().double(), self.B.bfloat16().double()
        R = (Ae.t() if self.tA else Ae) @ (Be.t() if self.tB else Be)
        if self.bias is not None:
            R = R + self.bias.double()
        if self.act == 1:
            R = torch.where(R > 0, R, 0.2 * R)
        elif self.act == 2:
            R = torch.tanh(R)
        if self.ref_t is not None:
            y = self.ref_t.double()
            R = R * (torch.where(y > 0, 1.0, 0.2) if self.dact == 1 else (1 - y * y))
        self._refs[bf16] = R
        return R

    def run(self, lib, variant, bf16=False, splits=1, x3=False):
        from chameleon_recsys_amd._lib import check, ptr
        d = self.dev
        C = torch.full((self.M, self.N), float('nan'), device=self.gpu)
        ws = torch.empty(32 << 20, dtype=torch.float32, device=self.gpu) if splits != 1 else None
        (lib.cham_gemm_f32x3_set_variant if x3 else lib.cham_gemm_set_variant)(variant)
        try:
            rc = (lib.cham_gemm_bf16 if bf16 else (lib.cham_gemm_f32x3 if x3 else lib.cham_gemm_f32))(
                ptr(d['A']), self.A.shape[1], self.tA, ptr(d['B']), self.B.shape[1], self.tB, ptr(C), self.N, self.M, self.N, self.K,
                ptr(d['bias']), self.act, ptr(d['ref']), self.N, self.dact, ptr(d['rs']), self.A.shape[1], self.rs_div, 0, ptr(ws),
                (32 << 20) * 4 if ws is not None else 0, splits, torch.cuda.current_stream().cuda_stream)
        finally:
            (lib.cham_gemm_f32x3_set_variant if x3 else lib.cham_gemm_set_variant)(-1)
        check(rc, "gemm")
        torch.cuda.synchronize()
        R = self.reference(bf16)
        return float((C.cpu().double() - R).abs().max()) / max(1.0, float(R.abs().max()))


ROWS = 72 * 19 * 51          # candidate rows of a 72-session G1-shape batch: 273 row tiles of 256 (>= 256: the N = 128 scorer GEMM stays on 256x128)


@pytest.mark.parametrize("bf16", [False, True])
def test_big_tiles_nn_car_forward(gpu, bf16):
    """CAR layer 2 forward: [69 768, 1024] x [1024, 1024], bias + tanh (nar_model.py:384-403)."""
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    case = _BigCase(gpu, ROWS, 1024, 1024, bias=True, act=2, seed=1)
    _counts(lib, reset=True)
    # tolerance: the pre-activations are N(0, 32^2) sums of 1024 products; fp32 accumulation along k rounds at the running sum's
    # magnitude (ulp(32) = 3.8e-6, x sqrt(1024) steps), and the maximum is taken over 71 M outputs - measured 1.0e-4; a wrong
    # fragment / tile mapping is an O(1) error
    tol = 3e-4
    assert case.run(lib, -1, bf16) < tol
    c = _counts(lib)
    assert c[8 * bf16 + 1] == 1, "the automatic choice for this shape is the 256x128 tile: %r" % (c,)
    assert case.run(lib, 0, bf16) < tol            # 128x128 (the instance the small tests cover) agrees on the same data
    if not bf16:
        assert case.run(lib, 4, bf16) < tol        # 256x256
        assert _counts(lib)[2] == 1


@pytest.mark.parametrize("bf16", [False, True])
def test_big_tiles_nt_car_dgrad(gpu, bf16):
    """CAR layer 2 dgrad: dZ2 [69 768, 1024] x W2^T, x leaky'(Z1) (backward of nar_model.py:375-403)."""
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    case = _BigCase(gpu, ROWS, 1024, 1024, transB=1, dref=True, dact=1, seed=2)
    _counts(lib, reset=True)
    assert case.run(lib, -1, bf16) < 5e-5
    c = _counts(lib)
    assert (c[8 + 1] == 1) if bf16 else (c[2] == 1), "automatic choice: 256x256 (fp32) / 256x128 (bf16): %r" % (c,)
    assert case.run(lib, 2, bf16) < 5e-5
    assert case.run(lib, 0, bf16) < 5e-5
    case_t = _BigCase(gpu, ROWS, 1024, 128, transB=1, dref=True, dact=2, seed=3)       # scorer layer-1 dgrad (K = 128) x tanh'
    assert case_t.run(lib, -1, bf16) < 5e-5
    assert case_t.run(lib, 4, bf16) < 5e-5


@pytest.mark.parametrize("bf16", [False, True])
def test_big_tiles_tn_w2_wgrad_splitk(gpu, bf16):
    """W2 weight gradient: Z1^T [1024, 62 016] x dZ2 [69 768, 1024], split-K; multiples of 8 splits take the one-K-split-per-XCD
    placement."""
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    case = _BigCase(gpu, 1024, 1024, ROWS, transA=1, seed=4)
    _counts(lib, reset=True)
    for splits in (0, 8, 16, 12):
        assert case.run(lib, -1, bf16, splits=splits) < 1e-4, splits
    c = _counts(lib)      # fp32: 16 tiles of 256x256 x {16, 16} splits cover the chip, x {8, 12} do not -> demoted to 256x128
    assert (c[8 + 1] == 4) if bf16 else (c[2] == 2 and c[1] == 2), c
    for variant in (2, 0):
        assert case.run(lib, variant, bf16, splits=16) < 1e-4, variant
        assert case.run(lib, variant, bf16, splits=0) < 1e-4, variant


@pytest.mark.parametrize("bf16", [False, True])
def test_big_tiles_rowscale_scorer_layer1(gpu, bf16):
    """Scorer layer 1 with the `cand (.) pred` product fused as a row-broadcast scale of A (nar_model.py:478-495), forward NN
    [69 768, 1024] x [1024, 128] + bias + leaky and its weight gradient (TN, split-K)."""
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    fwd = _BigCase(gpu, ROWS, 128, 1024, bias=True, act=1, rowscale=51, seed=5)
    _counts(lib, reset=True)
    assert fwd.run(lib, -1, bf16) < 5e-5
    c = _counts(lib)
    assert c[8 * bf16 + 1] == 1, c
    assert fwd.run(lib, 0, bf16) < 5e-5
    wg = _BigCase(gpu, 1024, 128, ROWS, transA=1, rowscale=51, seed=6)
    for splits in (0, 16):
        assert wg.run(lib, -1, bf16, splits=splits) < 1e-4
        assert wg.run(lib, 2, bf16, splits=splits) < 1e-4


@pytest.mark.parametrize("N", [6, 2, 10])
@pytest.mark.parametrize("x3,bf16", [(False, False), (True, False), (False, True)])
def test_nt_narrow_output_with_split_hint(gpu, N, x3, bf16):
    """transB with an output width that is not a multiple of four and automatic split-K asked for: the vectorised
    split-K reductions walk float4 groups of a row, so such a shape must take the unsplit kernel - through every public entry point."""
    tol = 2e-2 if bf16 else 1e-4
    assert _run(gpu, 96, N, 4096, transB=1, splits=0, x3=x3, bf16=bf16) < tol
    assert _run(gpu, 96, N, 4096, transB=1, splits=5, x3=x3, bf16=bf16) < tol
    assert _run(gpu, 96, N, 4096, transB=1, dref=True, dact=1, splits=0, x3=x3, bf16=bf16) < tol


# ---- small-output TN weight gradients on the VALU kernel (csrc/gemm.hip gemm_tn_small_kernel, round 6): M <= 128, long reduction
@pytest.mark.parametrize("M,N,K,splits", [(64, 32, 20000, 0), (128, 64, 20000, 0), (72, 1024, 4864, 0), (64, 32, 600, 1), (128, 128, 512, 1), (4, 4, 5000, 0),
                                          (60, 36, 3001, 0), (100, 200, 777, 0), (128, 72, 2049, 3)])
@pytest.mark.parametrize("x3", [False, True])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_small_output_tn_kernel(gpu, M, N, K, splits, x3, accumulate):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    before = _counts(lib)
    err = _run(gpu, M, N, K, transA=1, accumulate=accumulate, splits=splits, seed=M + N, x3=x3)
    after = _counts(lib)
    assert after[5] == before[5] + 1, (before, after)              # the small-output kernel took it (through either entry point)
    assert err < 2e-6, err                                          # fp32 products and accumulation: a K-term fp32 sum against float64


def test_small_output_tn_kernel_is_repeatable_and_leaves_other_shapes_alone(gpu):
    from chameleon_recsys_amd import _lib
    lib = _lib.load()
    a = [_run(gpu, 64, 32, 50000, transA=1, splits=0, seed=9) for _ in range(2)]
    assert a[0] == a[1]                                             # fixed summation order (no atomics)
    before = _counts(lib)
    _run(gpu, 256, 64, 5000, transA=1, splits=0)                   # M > 128: the MFMA tile kernels
    _run(gpu, 64, 32, 256, transA=1)                               # short reduction
    _run(gpu, 64, 32, 5000, transA=1, bias=False, rowscale=3, splits=0)      # row-broadcast scale on A: the MFMA kernels' staging
    assert _counts(lib)[5] == before[5]
