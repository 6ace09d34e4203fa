"""Negative sampler: the oracle against the reference's own property tests
(nar_module/nar/benchmarks/candidate_sampling_tests.py:10-99), and the HIP sampler bit-exact against the oracle."""
import numpy as np
import pytest

from oracle import philox, sampler as S

BUF = np.array([1, 2, 3, 1, 2, 3, 4, 4, 4, 5, 5, 5, 6, 6, 7, 7, 8, 9, 10] + [0] * 13, dtype=np.int64)


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    f = lambda *a: [int(w) for w in philox.philox4x32_10(*a)]
    assert f(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert f(*[0xffffffff] * 6) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert f(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# ---- reference property tests, candidate_sampling_tests.py ------------------------------------------------
def test_get_sample_from_recently_clicked_items_buffer():          # :16-20
    sample = S.sample_from_recent_buffer(BUF, 5, 42, 0)
    assert sample.shape == (5,) and 0 not in sample
    assert np.isin(sample, BUF).all()


def _click(valid_list, n, j=0):
    pool = np.array(valid_list, dtype=np.int64)
    return S.neg_items_click(pool, S.canonical_slots(pool), np.ones(len(pool), bool), n, 3, j, 42, 7)[0]


def test_get_neg_items_click():                                     # :22-26
    sample = _click([1, 2, 2, 4, 4, 5, 4, 3, 2, 16, 4, 8, 6], 5)
    assert sample.shape == (5,) and np.unique(sample).shape == (5,)


def test_get_neg_items_click_padding():                             # :28-33
    sample = _click([1, 2, 2], 10)
    assert sample.shape == (10,) and np.count_nonzero(sample) == 2 and not sample[2:].any()


def _session(session_items, cands, n):
    pool = np.array(cands, dtype=np.int64)
    canon = S.canonical_slots(pool)
    valid = ~np.isin(pool, session_items)
    return np.vstack([S.neg_items_click(pool, canon, valid, n, 0, j, 42, 0)[0] if c != 0 else np.zeros(n, np.int64)
                      for j, c in enumerate(session_items)])


def test_get_neg_items_session():                                   # :36-45
    samples = _session([1, 2, 3], [1, 3, 5, 7, 9, 11, 13, 15, 18, 20, 9, 11], 10)
    assert samples.shape == (3, 10) and np.count_nonzero(samples == 0) == 6 and not samples[:, -2:].any()
    for i in [1, 2, 3]:
        assert i not in samples


def test_get_negative_samples_padded_sessions():                    # :61-71
    aci = np.array([[1, 2, 3], [4, 0, 0]], dtype=np.int64)
    out = np.stack([_session(list(r), [1, 3, 5, 7, 9, 11, 13, 15, 18, 20, 9, 11], 10) for r in aci])
    assert out.shape == (2, 3, 10) and np.count_nonzero(out == 0) == 2 * 10 + 2 * 3 and not out[1, -2:].any()
    for sess, neg in zip(aci, out):
        assert not (set(sess.ravel()) & set(neg.ravel())) - {0}


def test_get_batch_negative_samples():                              # :74-99
    aci = np.array([[1, 2, 3, 4, 5], [4, 5, 6, 7, 0]], dtype=np.int64)
    out = S.batch_negative_samples(aci, BUF, 4, 10, 42, 0)           # drops the last position (nar_model.py:275)
    assert out.shape == (2, 4, 4)
    for sess, neg in zip(aci, out):
        assert not (set(sess.ravel()) & set(neg.ravel())) - {0}
    for b in range(2):
        for j in range(4):
            nz = out[b, j][out[b, j] != 0]
            assert len(np.unique(nz)) == len(nz)                     # no repetition per click


def test_dp_rows_are_sharding_independent():
    rng = np.random.default_rng(0)
    aci = rng.integers(1, 200, size=(16, 6)).astype(np.int64)
    aci[3, 4:] = 0
    buf = np.concatenate([rng.integers(1, 200, size=300), np.zeros(100, np.int64)])
    full = S.batch_negative_samples(aci, buf, 5, 50, 42, 3)
    lo = S.batch_negative_samples(aci, buf, 5, 50, 42, 3, rows=range(0, 8))
    hi = S.batch_negative_samples(aci, buf, 5, 50, 42, 3, rows=range(8, 16))
    assert np.array_equal(full, np.concatenate([lo, hi]))


def test_popularity_weighting_matches_reference_clone_in_distribution():
    """The reference's numpy clone draws np.random.permutation; ours a keyed sort: same first-pick distribution."""
    pool = np.array([7] * 6 + [8] * 3 + [9], dtype=np.int64)
    canon = S.canonical_slots(pool)
    first = [S.neg_items_click(pool, canon, np.ones(10, bool), 1, b, 0, 42, s)[0][0] for s in range(40) for b in range(50)]
    frac = np.bincount(first, minlength=10)[7:] / len(first)
    assert np.allclose(frac, [0.6, 0.3, 0.1], atol=0.04)


# ---- HIP sampler == oracle, bit exact -----------------------------------------------------------------------
def _gpu_sample(gpu, aci, buf, N, n_buf, seed, step, row_begin=0, row_count=None):
    import torch
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    Bg, T1 = aci.shape
    row_count = Bg if row_count is None else row_count
    d_aci = torch.from_numpy(aci).to(gpu); d_buf = torch.from_numpy(buf).to(gpu)
    neg = torch.zeros(row_count, T1 - 1, N, dtype=torch.int64, device=gpu)
    slot = torch.zeros(row_count, T1 - 1, N, dtype=torch.int32, device=gpu)
    pool = torch.zeros(20 * N, dtype=torch.int64, device=gpu)
    canon = torch.zeros(20 * N, dtype=torch.int32, device=gpu)
    meta = torch.zeros(4, dtype=torch.int32, device=gpu)
    nb = lib.cham_neg_sample_workspace_bytes(Bg * T1, len(buf), n_buf)
    ws = torch.empty(nb, dtype=torch.uint8, device=gpu)
    check(lib.cham_neg_sample(ptr(d_aci), Bg, T1, ptr(d_buf), len(buf), seed, step, row_begin, row_count, N, n_buf, ptr(neg),
                              ptr(slot), ptr(pool), ptr(canon), ptr(meta), ptr(ws), nb, torch.cuda.current_stream().cuda_stream),
          "neg_sample")
    torch.cuda.synchronize()
    return neg.cpu().numpy(), slot.cpu().numpy(), pool.cpu().numpy(), canon.cpu().numpy(), meta.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T1,N,n_buf,buf_size,n_items,fill", [
    (64, 8, 10, 100, 2000, 1000, 1500),       # config 1 (tiny)
    (16, 5, 4, 10, 64, 50, 0),                # empty buffer, few candidates -> zero padding
    (256, 20, 50, 3000, 20000, 46000, 20000),  # config 2 (G1 shape), full buffer
    (32, 30, 100, 5000, 6000, 13000, 4000),    # Adressa shape
])
def test_hip_sampler_bit_exact(gpu, B, T1, N, n_buf, buf_size, n_items, fill):
    rng = np.random.default_rng(B * 1000 + N)
    aci = rng.integers(1, n_items, size=(B, T1)).astype(np.int64)
    lens = rng.integers(2, T1 + 1, size=B)
    for b in range(B):
        aci[b, lens[b]:] = 0
    buf = np.zeros(buf_size, np.int64)
    buf[:fill] = rng.integers(1, n_items, size=fill)
    for step in (0, 5):
        neg, slot, pool, canon, meta = _gpu_sample(gpu, aci, buf, N, n_buf, 42, step)
        ref, aux = S.batch_negative_samples(aci, buf, N, n_buf, 42, step, return_aux=True)
        P = len(aux['pool'])
        assert meta[3] == P and meta[1] == len(aux['buf_sample'])
        assert np.array_equal(pool[:P], aux['pool']) and not pool[P:].any()
        assert np.array_equal(canon[:P], aux['canon'])
        assert np.array_equal(neg, ref)
        # slots point at the sampled id (or the pad slot)
        ok = (slot >= 0) & (slot < 20 * N)
        assert np.array_equal(pool[np.where(ok, slot, 0)][ok], neg[ok])
        assert (neg[slot == 20 * N] == 0).all() and (neg[slot < 0] == 0).all()


@pytest.mark.gpu
def test_hip_sampler_row_shards(gpu):
    rng = np.random.default_rng(5)
    aci = rng.integers(1, 500, size=(32, 9)).astype(np.int64)
    buf = rng.integers(1, 500, size=800).astype(np.int64)
    full = _gpu_sample(gpu, aci, buf, 8, 200, 42, 11)[0]
    a = _gpu_sample(gpu, aci, buf, 8, 200, 42, 11, 0, 16)[0]
    b = _gpu_sample(gpu, aci, buf, 8, 200, 42, 11, 16, 16)[0]
    assert np.array_equal(full, np.concatenate([a, b]))


@pytest.mark.gpu
def test_hip_sampler_global_batch_of_8_gpus(gpu):
    """Bg = 2048 sessions (8 ranks x 256): 44k pool keys -> the threshold-prefiltered rank-select must stay bit exact;
    the oracle is evaluated for two row shards only (as rank 0 / rank 5 would)."""
    rng = np.random.default_rng(77)
    B, T1, N, n_buf = 2048, 20, 50, 3000
    aci = rng.integers(1, 46000, size=(B, T1)).astype(np.int64)
    lens = rng.integers(2, T1 + 1, size=B)
    for b in range(B):
        aci[b, lens[b]:] = 0
    buf = rng.integers(1, 46000, size=20000).astype(np.int64)
    for rb in (0, 5 * 256):
        neg, slot, pool, canon, meta = _gpu_sample(gpu, aci, buf, N, n_buf, 42, 9, rb, 32)
        ref, aux = S.batch_negative_samples(aci, buf, N, n_buf, 42, 9, rows=range(rb, rb + 32), return_aux=True)
        assert np.array_equal(pool, aux['pool']) and np.array_equal(canon, aux['canon'])
        assert meta[1] == len(aux['buf_sample']) == n_buf
        assert np.array_equal(neg, ref)


@pytest.mark.gpu
def test_hip_sampler_nearly_empty_buffer_falls_back_to_full_select(gpu):
    """Fewer valid keys than the selection size: the prefilter must not drop anything."""
    rng = np.random.default_rng(3)
    aci = rng.integers(1, 300, size=(8, 6)).astype(np.int64)
    buf = np.zeros(5000, np.int64); buf[:40] = rng.integers(1, 300, size=40)
    neg, slot, pool, canon, meta = _gpu_sample(gpu, aci, buf, 25, 500, 42, 2)
    ref, aux = S.batch_negative_samples(aci, buf, 25, 500, 42, 2, return_aux=True)
    P = len(aux['pool'])
    assert meta[3] == P and meta[1] == len(aux['buf_sample']) == 40
    assert np.array_equal(pool[:P], aux['pool']) and np.array_equal(neg, ref)
