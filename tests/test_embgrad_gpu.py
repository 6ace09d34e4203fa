"""Deterministic embedding-table gradients (csrc/features.hip: cham_emb_grad_scan, cham_group_rows, cham_emb_grad_grouped) against a
float64 index_add reference - the IndexedSlices TF builds for tf.nn.embedding_lookup (nar_model.py:741, 918) - with heavy
duplicate keys (Zipf ids: the most popular article sits in ~20 % of the rows), and bit-reproducibility (two runs identical)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _s():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("R,n_items,dim,F,c0", [(10729, 46000, 117, 408, 288), (3000, 500, 44, 112, 8), (70, 1000, 256, 260, 4),
                                                (23457, 5000000, 16, 96, 0), (4000, 3000, 378, 520, 140),
                                                (1300021, 5000000, 16, 24, 4),         # more rows than rounds 1-3 could group (2^20)
                                                (2048, 70000, 32, 40, 8), (1, 300, 8, 8, 0), (4097, 2, 12, 16, 4)])   # one full sort tile; one row; two ids
def test_grouped_item_embedding_gradient(gpu, R, n_items, dim, F, c0):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    rng = np.random.default_rng(R)
    key_bits = 0 if R == 70 else int(n_items - 1).bit_length()        # 0: all 32 bits of the ids are sorted
    ids = np.minimum(rng.zipf(1.2, size=R) * 7919 % n_items, n_items - 1).astype(np.int64)
    ids[rng.random(R) < 0.05] = 0                                   # the padding item is a real table row too
    dxs = rng.standard_normal((R, F)).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(F)).astype(np.float32)
    uniq, inv = np.unique(ids, return_inverse=True)
    ref = np.zeros((len(uniq), dim))
    np.add.at(ref, inv, dxs[:, c0:c0 + dim].astype(np.float64))
    ref *= gamma[c0:c0 + dim].astype(np.float64)
    d_ids, d_dxs, d_gamma = torch.from_numpy(ids).to(gpu), torch.from_numpy(dxs).to(gpu), torch.from_numpy(gamma).to(gpu)
    perm = torch.full((R,), -1, dtype=torch.int32, device=gpu)
    ws = torch.zeros(lib.cham_group_rows_workspace_bytes(R) // 4, dtype=torch.int32, device=gpu)
    seg = torch.full((int(lib.cham_group_rows_segments_len(R)),), -7, dtype=torch.int32, device=gpu)
    check(lib.cham_group_rows(ptr(d_ids), R, key_bits, ptr(perm), ptr(seg), ptr(ws), ws.numel() * 4, _s()), "cham_group_rows")
    torch.cuda.synchronize()
    pm = perm.cpu().numpy()
    assert np.array_equal(np.sort(pm), np.arange(R)), "perm is not a permutation"
    key = ids[pm] * (1 << 24) + pm
    assert (np.diff(key) > 0).all(), "rows are not sorted by (id, row)"
    # segment table: heads of the runs of equal ids in sorted order, the long ones listed
    sg = seg.cpu().numpy()
    sid = ids[pm]
    heads = np.flatnonzero(np.r_[True, sid[1:] != sid[:-1]])
    assert sg[0] == len(heads) and np.array_equal(sg[4:4 + len(heads)], heads) and sg[4 + len(heads)] == R
    lens = np.diff(np.r_[heads, R])
    longs = np.flatnonzero(lens > 32)
    assert sg[1] == len(longs) and np.array_equal(sg[R + 6:R + 6 + len(longs)], longs)
    chunks = (lens[longs] + 63) // 64                                # work items: 64-row chunks of the long segments
    wf = R + 6 + R // 33 + 2
    assert sg[2] == chunks.sum() and np.array_equal(sg[wf:wf + len(longs) + 1], np.r_[0, np.cumsum(chunks)])
    outs = []
    for _ in range(2):
        table = torch.zeros(n_items * dim if n_items <= 50000 else int(uniq.max() + 1) * dim, device=gpu)
        check(lib.cham_emb_grad_grouped(ptr(d_dxs), R, F, c0, dim, ptr(d_gamma), ptr(d_ids), ptr(perm), ptr(seg), ptr(table), _s()), "grouped")
        torch.cuda.synchronize()
        outs.append(table.cpu())
    assert torch.equal(outs[0], outs[1]), "not bit-reproducible"
    got = outs[0].view(-1, dim).double().numpy()
    assert np.abs(got[uniq] - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    untouched = np.ones(got.shape[0], bool); untouched[uniq] = False
    assert not got[untouched].any()


@pytest.mark.parametrize("R,card,dim,F,c0,via_ids", [(4864, 12, 14, 72, 10, False), (7424, 1022, 45, 124, 0, False),
                                                      (10729, 461, 37, 408, 0, True), (300, 29, 18, 72, 54, False), (5000, 300, 130, 200, 3, True)])
def test_scan_small_table_gradient(gpu, R, card, dim, F, c0, via_ids):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    rng = np.random.default_rng(R + card)
    dxs = rng.standard_normal((R, F)).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(F)).astype(np.float32)
    if via_ids:                       # article metadata: key(r) = meta[ids[r]]
        n_items = 20000
        meta = rng.integers(0, card, size=n_items, dtype=np.int64)
        ids = rng.integers(0, n_items, size=R, dtype=np.int64)
        keys, keysrc, d_ids = meta[ids], torch.from_numpy(meta).to(gpu), torch.from_numpy(ids).to(gpu)
    else:
        keys = np.minimum(rng.zipf(1.5, size=R) - 1, card - 1).astype(np.int64)
        keysrc, d_ids = torch.from_numpy(keys).to(gpu), None
    ref = np.zeros((card, dim))
    np.add.at(ref, keys, dxs[:, c0:c0 + dim].astype(np.float64))
    ref *= gamma[c0:c0 + dim].astype(np.float64)
    d_dxs, d_gamma = torch.from_numpy(dxs).to(gpu), torch.from_numpy(gamma).to(gpu)
    outs = []
    for _ in range(2):
        table = torch.full((card, dim), float('nan'), device=gpu)
        check(lib.cham_emb_grad_scan(ptr(d_dxs), R, F, c0, dim, ptr(d_gamma), ptr(keysrc), ptr(d_ids), card, ptr(table), _s()), "scan")
        torch.cuda.synchronize()
        outs.append(table.cpu())
    assert torch.equal(outs[0], outs[1]), "not bit-reproducible"
    assert np.abs(outs[0].double().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
