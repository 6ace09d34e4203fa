"""Item-row gather / assemble (nar_model.py:921-994 + scale/center :887-907): the LDS-tiled kernel (csrc/features.hip
k_item_assemble_lds: contiguous source segments staged per workgroup, 16-byte row stores) must produce exactly the rows of the
one-thread-per-element kernel it replaces - G1 and Adressa feature schemas, ragged tile tails."""
import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import synthetic

pytestmark = pytest.mark.gpu


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
    # the ACE block really is the article's content embedding row
    segs = rt.item_segs.cpu().numpy()
    ace_seg = [sg for sg in segs if sg[0] == 0][0]
    assert torch.equal(outs[1][0][:, ace_seg[1]:ace_seg[1] + ace_seg[2]], rt.ace.cpu()[ids.cpu()])
