"""Item-row gather / assemble (nar_model.py:921-994 + scale/center :887-907): the LDS-tiled kernel (csrc/features.hip
k_item_assemble_lds: contiguous source segments staged per workgroup, 16-byte row stores) must produce exactly the rows of the
one-thread-per-element kernel it replaces - and both the rows of a numpy gather built from the layout's column descriptors - G1 and
Adressa feature schemas, ragged tile tails."""
import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import synthetic
from chameleon_recsys_amd.nar.layout import COL_ACE, COL_EMB, COL_ITEMEMB, COL_NOVELTY, COL_NUM, COL_OHE, COL_RECENCY

pytestmark = pytest.mark.gpu


def _numpy_item_rows(rt, ids, rec, nov, stats, g1, g2):
    """Item feature rows [R, Fi] from the resident tables, column by column as the layout describes them (float32 arithmetic)."""
    L = rt.layout
    desc = L.item_descriptors()
    flat, meta, ace = rt.flat.cpu().numpy(), rt.meta_cat.cpu().numpy(), rt.ace.cpu().numpy()
    R = ids.shape[0]
    grp = np.where(np.arange(R) < g1, 0, np.where(np.arange(R) < g2, 1, 2))
    f32 = np.float32

    def norm(x, st):          # normalize_values + min_max_normalization, nar_model.py:996-1039
        z = (x - st[:, 0]) / st[:, 1]
        scaled = (z - st[:, 2] + f32(1e-24)) / np.maximum(st[:, 3] - st[:, 2], f32(2e-24))
        return scaled * f32(2.0) - f32(1.0)
    out = np.zeros((R, L.Fi), np.float32)
    for c, (kind, feat, sub, dim, off) in enumerate(desc):
        if kind == COL_OHE:
            out[:, c] = (meta[feat, ids] == sub).astype(np.float32)
        elif kind == COL_EMB:
            out[:, c] = flat[off + meta[feat, ids] * dim + sub]
        elif kind == COL_NUM:
            out[:, c] = meta[feat, ids].astype(np.float32)
        elif kind == COL_ACE:
            out[:, c] = ace[ids, sub]
        elif kind == COL_ITEMEMB:
            out[:, c] = flat[off + ids * dim + sub]
        elif kind == COL_RECENCY:
            out[:, c] = norm(rec.astype(f32), stats[grp, 0:4].astype(f32))
        elif kind == COL_NOVELTY:
            out[:, c] = norm(nov.astype(f32), stats[grp, 4:8].astype(f32))
    return out


@pytest.mark.parametrize("dataset,n_items,D,R", [("gcom", 5000, 250, 1003), ("adressa", 3000, 64, 517), ("gcom", 2000, 128, 7)])
def test_item_assemble_lds_equals_elementwise(gpu, dataset, n_items, D, R):
    from chameleon_recsys_amd._lib import check, ptr
    from chameleon_recsys_amd.nar.nar_model import NARRuntime
    p = synthetic.default_params(n_items, D, C=128, H=64, dataset=dataset, buffer_size=500, for_norm=100)
    rt = NARRuntime(p, seed=3)
    L, lib = rt.layout, rt.lib
    rng = np.random.default_rng(R)
    w = rt.logical_weights()
    w['gamma'] = (1.0 + 0.1 * rng.standard_normal(w['gamma'].shape)).astype(np.float32)
    w['beta'] = (0.05 * rng.standard_normal(w['beta'].shape)).astype(np.float32)
    rt.load_logical_weights(w)
    ids = torch.from_numpy(rng.integers(0, n_items, size=R).astype(np.int64)).to(gpu)
    rec = torch.from_numpy(rng.random(R).astype(np.float32) * 5).to(gpu)
    nov = torch.from_numpy(rng.random(R).astype(np.float32) * 9).to(gpu)
    stats = torch.from_numpy(np.tile(np.array([2.0, 1.5, -1.2, 2.2, 4.0, 2.5, -1.5, 1.9], np.float32), (3, 1))).to(gpu)
    s = torch.cuda.current_stream().cuda_stream
    Fi, g1, g2 = L.Fi, R // 3, 2 * R // 3
    outs = []
    for lds in (False, True):
        xraw = torch.full((R, Fi), float('nan'), device=gpu)
        xs = torch.full((R, Fi), float('nan'), device=gpu)
        if lds:
            check(lib.cham_item_assemble_lds(ptr(ids), R, g1, g2, ptr(rt.meta_cat), rt.n_items, ptr(rt.ace), L.D, ptr(rec), ptr(nov), ptr(stats),
                                             ptr(rt.item_desc), Fi, ptr(rt.item_segs), rt.n_item_segs, ptr(rt.item_singles), rt.n_item_singles,
                                             ptr(rt.flat), ptr(rt.p('gamma_item')), ptr(rt.p('beta_item')), ptr(xraw), ptr(xs), s), "lds")
        else:
            check(lib.cham_item_assemble(ptr(ids), R, g1, g2, ptr(rt.meta_cat), rt.n_items, ptr(rt.ace), L.D, ptr(rec), ptr(nov), ptr(stats),
                                         ptr(rt.item_desc), Fi, ptr(rt.flat), ptr(rt.p('gamma_item')), ptr(rt.p('beta_item')), ptr(xraw),
                                         ptr(xs), s), "elementwise")
        torch.cuda.synchronize()
        outs.append((xraw.cpu(), xs.cpu()))
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]), "raw feature rows differ"
    assert torch.equal(outs[0][1], outs[1][1]), "scaled feature rows differ"
    # ... and both against a NUMPY gather built from the layout's column descriptors (nar_model.py:921-994 column order): table rows and
    # one-hot bits bit-exact, the two normalised columns and the gamma / beta image to fp32 rounding (the device contracts a*b+c to an FMA)
    ref_raw = _numpy_item_rows(rt, ids.cpu().numpy(), rec.cpu().numpy(), nov.cpu().numpy(), stats.cpu().numpy(), g1, g2)
    desc = L.item_descriptors()
    dyn = np.isin(desc[:, 0], (COL_RECENCY, COL_NOVELTY))
    got_raw, got_s = outs[1][0].numpy(), outs[1][1].numpy()
    assert np.array_equal(got_raw[:, ~dyn], ref_raw[:, ~dyn]), "gathered columns differ from the numpy gather"
    assert np.allclose(got_raw[:, dyn], ref_raw[:, dyn], rtol=2e-6, atol=2e-6)
    gam, bet = rt.p('gamma_item').cpu().numpy(), rt.p('beta_item').cpu().numpy()
    assert np.allclose(got_s, ref_raw * gam[None, :] + bet[None, :], rtol=2e-6, atol=2e-6)
    # the ACE block really is the article's content embedding row
    segs = rt.item_segs.cpu().numpy()
    ace_seg = [sg for sg in segs if sg[0] == 0][0]
    assert torch.equal(outs[1][0][:, ace_seg[1]:ace_seg[1] + ace_seg[2]], rt.ace.cpu()[ids.cpu()])


@pytest.mark.parametrize("R,F", [(10729, 408), (4864, 72), (37, 8), (1, 4), (5000, 1024), (300, 100)])
def test_feature_bwd_coalesced_column_sums(gpu, R, F):
    """cham_feature_bwd_ws (round 6: 64 columns x a row chunk per workgroup, coalesced row segments, partials added in a fixed order) against
    float64 and against cham_feature_bwd (one workgroup per column): dgamma = sum_r dxs * xraw, dbeta = sum_r dxs; repeatable bit for bit."""
    import torch
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(R + F)
    dxs = torch.randn(R, F, device=gpu, generator=g)
    xraw = torch.randn(R, F, device=gpu, generator=g)
    st = torch.cuda.current_stream().cuda_stream
    need = lib.cham_feature_bwd_workspace_bytes(F)
    ws = torch.empty(need // 4, dtype=torch.float32, device=gpu)
    outs = []
    for _ in range(2):
        dg, db = torch.full((F,), float('nan'), device=gpu), torch.full((F,), float('nan'), device=gpu)
        check(lib.cham_feature_bwd_ws(ptr(dxs), ptr(xraw), R, F, ptr(dg), ptr(db), ptr(ws), need, st), "cham_feature_bwd_ws")
        outs.append((dg.clone(), db.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dg0, db0 = torch.empty(F, device=gpu), torch.empty(F, device=gpu)
    check(lib.cham_feature_bwd(ptr(dxs), ptr(xraw), R, F, ptr(dg0), ptr(db0), st), "cham_feature_bwd")
    rg, rb = (dxs.double() * xraw.double()).sum(0), dxs.double().sum(0)
    tol = 1e-6 * max(1.0, R ** 0.5) * 4
    for got, ref, old in ((outs[0][0], rg, dg0), (outs[0][1], rb, db0)):
        assert float((got.double() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
        assert float((got - old).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    assert lib.cham_feature_bwd_ws(ptr(dxs), ptr(xraw), R, F, ptr(dg0), ptr(db0), ptr(ws), need - 4, st) == -22      # workspace too small
