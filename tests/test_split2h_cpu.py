"""The arithmetic claim behind csrc/gemm_h2.hip, checked on the CPU with numpy's IEEE float16 (round to nearest even, subnormals kept -
what v_cvt_f16_f32 / v_fma_mixlo_f16 do under float_denorm_mode_16_64 = 3): an fp32 number times a power-of-two scale splits into two
fp16 numbers that carry 22 of its 24 significand bits, and the THREE plane products the kernel keeps (h h + h l + l h) give a dot
product whose error against float64 is of the native fp32 MFMA's class - half the matrix instructions of the six-product bf16 split
(tests/test_split3_cpu.py).  Also the scale rule of k_h2_scale_* and the two range limits of fp16 the scale exists for."""
import numpy as np
import pytest


def h2_scale(bound):
    """h2_finish_scale: bound = m 2^e (0.5 <= m < 1) -> scale = 2^(15 - e), so that bound * scale lies in [2^14, 2^15)."""
    if not (bound > 0 and np.isfinite(bound)):
        return 1.0
    _, e = np.frexp(np.float32(bound))
    k = int(np.clip(15 - int(e), -110, 110))
    return float(np.ldexp(1.0, k))


def split2h(x, scale):
    xs = (np.asarray(x, np.float32) * np.float32(scale)).astype(np.float32)          # exact (power of two) unless it leaves the fp32 range
    with np.errstate(over='ignore'):
        h = xs.astype(np.float16)
    r = (xs - h.astype(np.float32)).astype(np.float32)                              # exact in fp32
    l = r.astype(np.float16)
    hb = h.view(np.uint16).copy()
    hb[(np.asarray(x) > 0) & (hb == 0)] = 1                                         # h2_keep_sign: a positive value never stores +0
    return hb.view(np.float16), l


def test_scale_rule_puts_the_bound_below_2_15():
    rng = np.random.default_rng(0)
    b = np.exp2(rng.uniform(-60, 60, 2000)).astype(np.float32)
    b = np.concatenate([b, np.float32([1.0, 2.0, 0.5, 32768.0, 65504.0, 1e-30, 3e38])])
    for x in b:
        s = h2_scale(x)
        assert 2.0 ** 14 <= float(x) * s < 2.0 ** 15 or s in (2.0 ** -110, 2.0 ** 110), (x, s)          # (clamped: scale and 1 / scale stay normal)
        assert np.ldexp(1.0, int(np.log2(s))) == s          # an exact power of two
    assert h2_scale(0.0) == 1.0 and h2_scale(float('inf')) == 1.0 and h2_scale(float('nan')) == 1.0


def test_two_fp16_planes_carry_22_bits_and_keep_the_sign():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(1 << 20) * np.exp2(rng.uniform(-12, 0, 1 << 20))).astype(np.float32)
    s = h2_scale(float(np.abs(x).max()))
    h, l = split2h(x, s)
    assert np.all(np.isfinite(h.astype(np.float32))) and float(np.abs(h.astype(np.float32)).max()) <= 2.0 ** 15
    back = (h.astype(np.float64) + l.astype(np.float64)) / s
    rel = np.abs(back - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    normal_l = np.abs(x.astype(np.float64) * s) >= 2.0 ** -3          # the l plane is a normal fp16 number: 11 + 11 bits
    assert float(rel[normal_l].max()) <= 2.0 ** -22
    assert float(np.abs(back - x)[~normal_l].max() if (~normal_l).any() else 0.0) <= 2.0 ** -25 / s
    # sign of the h plane as a signed 16-bit pattern == sign of x, for every element (leaky' is read from it)
    assert np.array_equal(h.view(np.int16) > 0, x > 0)


def test_wide_dynamic_range_rows_and_subnormals():
    """A gradient-like matrix: rows spread over 40 binary orders of magnitude under ONE per-tensor scale.  Rows within 2^-18 of the bound keep
    22 bits; below that the ABSOLUTE error stays <= 2^-25 / scale = 2^-40 of the bound (fp16 subnormal spacing 2^-24, half of it), values
    below that flush to (signed) zero - none of it visible in a sum that also contains the large rows."""
    rng = np.random.default_rng(2)
    R, C = 4096, 64
    mag = np.exp2(-rng.uniform(0, 40, R))[:, None]
    x = (rng.standard_normal((R, C)) * mag).astype(np.float32)
    bound = float(np.abs(x).max()) * 3.7                                # a bound, not the maximum (Cauchy-Schwarz is loose by a few x)
    s = h2_scale(bound)
    h, l = split2h(x, s)
    back = (h.astype(np.float64) + l.astype(np.float64)) / s
    err = np.abs(back - x.astype(np.float64))
    assert float(err.max()) <= max(2.0 ** -22 * float(np.abs(x).max()), 2.0 ** -24 / s)
    big = np.abs(x) * s >= 2.0 ** -3
    assert float((err[big] / np.abs(x[big])).max()) <= 2.0 ** -22
    assert float(err[~big].max()) <= 2.0 ** -24 / s                      # (2^-25 / s, doubled where h2_keep_sign lifts a +0 to the smallest subnormal)
    assert 2.0 ** -24 / s <= 2.0 ** -38 * bound
    # a column sum over all rows (what the W2 weight gradient does with such a matrix): error far below fp32's own rounding of the result
    w = rng.standard_normal(R)
    exact = (x.astype(np.float64) * w[:, None]).sum(0)
    got = (back * w[:, None]).sum(0)
    assert float(np.abs(got - exact).max()) < 2.0 ** -22 * float(np.abs(exact).max())


def test_three_plane_products_match_the_fp32_product():
    rng = np.random.default_rng(3)
    n = 1 << 18
    a = (rng.standard_normal(n) * np.exp2(rng.uniform(-6, 0, n))).astype(np.float32)
    b = (rng.standard_normal(n) * np.exp2(rng.uniform(-6, 0, n))).astype(np.float32)
    sa, sb = h2_scale(float(np.abs(a).max())), h2_scale(float(np.abs(b).max()))
    ah, al = (p.astype(np.float64) for p in split2h(a, sa))
    bh, bl = (p.astype(np.float64) for p in split2h(b, sb))
    # every fp16 x fp16 product has a 22-bit significand: exact in fp32 (the MFMA multiplies exactly and accumulates in fp32)
    for p in (ah * bh, ah * bl, al * bh):
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)
    kept = (ah * bh + ah * bl + al * bh) / (sa * sb)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    full = (np.abs(a) * sa >= 2.0 ** -3) & (np.abs(b) * sb >= 2.0 ** -3)          # both l planes are normal fp16 numbers (all but ~5e-4 of the draws)
    assert full.mean() > 0.999
    assert float(rel[full].max()) < 2.0 ** -20.5 and float(np.sqrt((rel[full] ** 2).mean())) < 2.0 ** -22.5, (float(rel[full].max()), float(np.sqrt((rel[full] ** 2).mean())))
    assert float(np.abs(kept - exact)[~full].max() if (~full).any() else 0.0) < 2.0 ** -22 * float(np.abs(exact).max())
    # without the two cross terms the error is 2^11 larger: they are what makes the split fp32-grade
    rel1 = np.abs(ah * bh / (sa * sb) - exact) / np.abs(exact)
    assert float(np.median(rel1)) > 2.0 ** -13


H2_ORDER = ((1, 0), (0, 0), (0, 1))          # gemm_h2.hip's pass order: A_l x B_h, A_h x B_h, A_h x B_l


@pytest.mark.parametrize("K", [1024, 4096])
def test_plane_product_dot_product_error_is_fp32_class(K):
    """A K-long dot product of unit-scale operands: three-plane accumulation in fp32 (one rounding per 16-k MFMA block and pass) vs the
    native fp32 MFMA's accumulation (v_mfma_f32_32x32x2_f32: two k per accumulation), both against float64 - the bar
    tests/test_gemm_h2_gpu.py holds the kernel to on the GPU (<= 1.5 x native)."""
    rng = np.random.default_rng(4)
    n = 1500
    a = rng.standard_normal((n, K)).astype(np.float32)
    b = (rng.standard_normal((n, K)) * K ** -0.5).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    sa, sb = h2_scale(float(np.abs(a).max())), h2_scale(float(np.abs(b).max()))
    A = [p.astype(np.float64) for p in split2h(a, sa)]
    B = [p.astype(np.float64) for p in split2h(b, sb)]
    acc = np.zeros(n, np.float32)
    for k0 in range(0, K, 16):
        for pa, pb in H2_ORDER:
            blk = (A[pa][:, k0:k0 + 16] * B[pb][:, k0:k0 + 16]).sum(1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    got = (acc.astype(np.float64) / sa) / sb
    plain = np.zeros(n, np.float32)
    for k0 in range(0, K, 2):
        plain = (plain.astype(np.float64) + (a[:, k0:k0 + 2].astype(np.float64) * b[:, k0:k0 + 2].astype(np.float64)).sum(1)).astype(np.float32)
    scale = np.abs(exact).max()
    e_planes, e_plain = np.abs(got - exact).max() / scale, np.abs(plain - exact).max() / scale
    r_planes, r_plain = np.sqrt(((got - exact) ** 2).mean()) / scale, np.sqrt(((plain - exact) ** 2).mean()) / scale
    print("K=%d: max err %.2e (native fp32 %.2e), rms %.2e (%.2e)" % (K, e_planes, e_plain, r_planes, r_plain))
    assert e_planes < 1.5 * e_plain + 1e-7 and r_planes < 1.5 * r_plain + 2e-8, (e_planes, e_plain, r_planes, r_plain)


def test_constant_record_of_a_product_of_two_tanh_outputs():
    """cham_gemm_f32x2h's A operand in the scorer's first layer (nar_model.py:478-495 of the reference): cand (.) pred, both tanh outputs, takes
    the CONSTANT record {2^14, 2^-14} instead of a measured bound - |x| 2^14 <= 2^14 stays a factor of four inside fp16's range and the three
    plane products of a K = 1024 dot product against a weight matrix scaled by its max ROW NORM (>= max |w|: what k_h2_scale_rownorm derives
    without a second pass) err by less than one plain fp32 summation of the same dot product does."""
    rng = np.random.default_rng(5)
    M, K, N = 512, 1024, 128
    cand = np.tanh(rng.standard_normal((M, K)) * 0.7).astype(np.float32)
    pred = np.tanh(rng.standard_normal((M // 32, K)) * 0.7).astype(np.float32)
    A = (cand * np.repeat(pred, 32, 0)).astype(np.float32)                  # the fp32 product the kernel forms while staging
    W = (rng.standard_normal((K, N)) * 0.03).astype(np.float32)
    sa = 2.0 ** 14
    sw = h2_scale(float(np.sqrt((W.astype(np.float64) ** 2).sum(1)).max()) * 1.0009765625)
    assert float(np.abs(A).max()) * sa <= 2.0 ** 14 and float(np.abs(W).max()) * sw < 2.0 ** 15
    ah, al = (p.astype(np.float64) for p in split2h(A, sa))
    wh, wl = (p.astype(np.float64) for p in split2h(W, sw))
    ref = A.astype(np.float64) @ W.astype(np.float64)
    planes = (ah @ wh + ah @ wl + al @ wh) / (sa * sw)                      # the three products, exactly accumulated: representation error only
    plain = (A @ W).astype(np.float64)                                      # an fp32 summation of the exact operands
    scale = float(np.abs(ref).max())
    e_planes, e_plain = float(np.abs(planes - ref).max()) / scale, float(np.abs(plain - ref).max()) / scale
    assert e_planes < 2.0 ** -21 and e_planes < e_plain, (e_planes, e_plain)
    assert abs(float((planes - ref).mean())) / scale < 1e-9               # and no bias: round-to-nearest planes
