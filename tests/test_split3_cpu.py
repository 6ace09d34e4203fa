"""The arithmetic claim behind csrc/gemm_x3.hip, checked on the CPU with a bit-level emulation of round-to-nearest-even bf16
(no GPU, no library call): an fp32 number splits EXACTLY into three bf16 numbers, and the six plane products kept by the kernel
reproduce the fp32 product to ~2^-24 of |a||b| - the size of one fp32 rounding, i.e. what the native fp32 MFMA path also carries per
accumulation step."""
import numpy as np
import pytest


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite inputs)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = u + 0x7FFF + ((u >> 16) & 1)
    return ((r >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(a):
    a = np.asarray(a, np.float32)
    h = bf16_rne(a)
    r = (a - h).astype(np.float32)          # exact in fp32 (Sterbenz-like: |r| <= 2^-8 |a|, fits 16 bits of significand)
    m = bf16_rne(r)
    r2 = (r - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def test_three_bf16_planes_sum_to_the_fp32_value_exactly():
    rng = np.random.default_rng(0)
    mant = rng.standard_normal(1 << 20).astype(np.float32)
    expo = rng.integers(-100, 100, mant.size)
    a = (mant * np.exp2(expo.astype(np.float64))).astype(np.float32)
    a = np.concatenate([a, np.float32([0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.17549435e-38, 1e-30, 255.99998, 1 + 2 ** -23, 1 - 2 ** -24])])
    h, m, l = split3(a)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)            # each plane is a bf16 number
    s = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    big = np.abs(a) >= 2.0 ** -100          # below that the lowest plane (~2^-16 |a|) falls into bf16's subnormal range (spacing 2^-133)
    assert np.array_equal(s[big | (a == 0)], a.astype(np.float64)[big | (a == 0)])          # exact, bit for bit
    assert np.all(np.abs(s - a.astype(np.float64))[~big] <= 2.0 ** -133)                    # tiny operands: absolute error of one subnormal step
    nz = a != 0
    assert np.all(np.abs(m[nz]) <= np.abs(a[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(a[nz]) * 2.0 ** -16)


def test_six_plane_products_match_the_fp32_product_to_one_fp32_rounding():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(1 << 18).astype(np.float32) * np.exp2(rng.integers(-20, 20, 1 << 18)).astype(np.float32)
    b = rng.standard_normal(1 << 18).astype(np.float32) * np.exp2(rng.integers(-20, 20, 1 << 18)).astype(np.float32)
    ah, am, al = (x.astype(np.float64) for x in split3(a))
    bh, bm, bl = (x.astype(np.float64) for x in split3(b))
    kept = ah * bh + (ah * bm + am * bh) + (ah * bl + al * bh + am * bm)      # every bf16 x bf16 product is exact in fp32 (16-bit significand)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert float(rel.max()) < 2.0 ** -23 and float(np.median(rel)) < 2.0 ** -27, (float(rel.max()), float(np.median(rel)))
    # without the three 2^-16 terms the error is 256x larger: they are what makes the split fp32-grade
    rel3 = np.abs(ah * bh + ah * bm + am * bh - exact) / np.abs(exact)
    assert float(rel3.max()) > 2.0 ** -17


X3_ORDER = ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))       # gemm_x3.hip: smallest terms first
P3_ORDER = ((2, 0), (1, 0), (0, 0), (0, 2), (1, 1), (0, 1))       # gemm_p3.hip: the order that lets one fragment set serve the pipeline


@pytest.mark.parametrize("order", [X3_ORDER, P3_ORDER], ids=["x3", "p3"])
def test_plane_product_dot_product_error_is_fp32_class(order):
    """A K = 1024 dot product: six-plane accumulation in fp32 (as the MFMA does per 16-k block, in either kernel's pass order - the
    accumulator already holds the sum over the previous k blocks, so the order inside a block does not matter) vs plain fp32
    accumulation, both against float64."""
    rng = np.random.default_rng(2)
    K, n = 1024, 2000
    a = rng.standard_normal((n, K)).astype(np.float32)
    b = rng.standard_normal((n, K)).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    A, B = split3(a), split3(b)
    acc = np.zeros(n, np.float32)
    for k0 in range(0, K, 16):
        for pa, pb in order:
            blk = (A[pa][:, k0:k0 + 16].astype(np.float64) * B[pb][:, k0:k0 + 16].astype(np.float64)).sum(1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)          # one fp32 rounding per MFMA accumulation
    plain = np.zeros(n, np.float32)
    for k0 in range(0, K, 2):                                                 # v_mfma_f32_32x32x2_f32: two k per accumulation
        plain = (plain.astype(np.float64) + (a[:, k0:k0 + 2].astype(np.float64) * b[:, k0:k0 + 2].astype(np.float64)).sum(1)).astype(np.float32)
    scale = np.abs(exact).max()
    e_planes, e_plain = np.abs(acc - exact).max() / scale, np.abs(plain - exact).max() / scale
    assert e_planes < 2e-6 and e_planes < 1.5 * e_plain + 1e-7, (e_planes, e_plain)
