"""PreCAR combine backward (csrc/scorer.hip cham_combine_bwd): dU / dV against a float64 index_add reference.

Covers what the step-parity tests reach only by luck: hot pool slots referenced by (almost) every position (split over position
chunks), cold slots, the zero-padding slot held MANY times per click (fewer than N unique candidates: first batches, small
catalogs - nar_model.py:1252 zero-pads; the round-1 kernel truncated its match list there), masked clicks (slot -1),
and bit-reproducibility (no float atomics: two runs are bit-identical)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_slots(rng, BT, N, pmax, pad_frac, hot, masked_frac):
    """[BT, N] int32: per position N DISTINCT pool slots (popularity-skewed; slots < hot sit in every position), then a tail of
    ~pad_frac * N entries replaced by the zero-padding slot pmax; masked_frac of the positions are all -1."""
    slot = np.empty((BT, N), np.int32)
    w = 1.0 / np.arange(1, pmax + 1) ** 1.1
    w /= w.sum()
    for p in range(BT):
        s = rng.choice(pmax, size=N, replace=False, p=w)
        rest = s[~np.isin(s, np.arange(hot))]
        s = np.concatenate([np.arange(hot), rest])[:N].astype(np.int32)
        rng.shuffle(s)
        npad = int(round(pad_frac * N * rng.uniform(0.5, 1.5))) if pad_frac > 0 else 0
        if npad:
            s[N - min(npad, N):] = pmax
        slot[p] = s
    masked = rng.random(BT) < masked_frac
    slot[masked] = -1
    return slot, masked


@pytest.mark.parametrize("BT,N,pmax,C,pad_frac,hot,masked_frac", [
    (700, 50, 1000, 256, 0.0, 3, 0.0),          # hot slots in all 700 positions (> 512: chunked partials), no padding
    (1300, 50, 1000, 128, 0.3, 2, 0.1),         # 30 % zero-padded negatives, masked clicks, 3 position chunks
    (90, 9, 180, 128, 0.6, 1, 0.2),             # tiny: mostly padding
    (37, 200, 4000, 64, 0.05, 0, 0.0),          # config-5 negatives count
])
def test_combine_bwd_matches_index_add(gpu, BT, N, pmax, C, pad_frac, hot, masked_frac):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    rng = np.random.default_rng(BT + N)
    slot, masked = _make_slots(rng, BT, N, pmax, pad_frac, hot, masked_frac)
    NC = N + 1
    dpre = rng.standard_normal((BT + BT * NC, C)).astype(np.float32)
    cand = dpre[BT:].reshape(BT, NC, C)                        # (a view: the edits below land in dpre)
    cand[masked] = 0.0                                         # masked clicks carry exactly zero gradient
    dpre[:BT][masked] = 0.0
    # float64 reference
    dU = dpre[:BT].astype(np.float64) + cand.astype(np.float64).sum(1)
    dV = np.zeros((2 * BT + pmax + 1, C))
    dV[:BT] = dpre[:BT]
    dV[BT:2 * BT] = cand[:, 0]
    s = slot.reshape(-1)
    ok = s >= 0
    np.add.at(dV, 2 * BT + s[ok], cand[:, 1:].reshape(-1, C)[ok].astype(np.float64))
    if pad_frac >= 0.2:
        assert (s == pmax).sum() > 0.125 * ok.sum()            # the case ADVICE r01 asked for
    d_dpre, d_slot = torch.from_numpy(dpre).to(gpu), torch.from_numpy(slot).to(gpu)
    need = lib.cham_combine_bwd_workspace_bytes(C, BT, N, pmax)
    outs = []
    for _ in range(2):
        ws = torch.full(((need + 3) // 4,), float('nan'), dtype=torch.float32, device=gpu)
        o_dU = torch.full((BT, C), float('nan'), device=gpu)
        o_dV = torch.full((2 * BT + pmax + 1, C), float('nan'), device=gpu)
        check(lib.cham_combine_bwd(ptr(d_dpre), C, BT, N, pmax, ptr(d_slot), ptr(o_dU), ptr(o_dV), ptr(ws), ws.numel() * 4,
                                   torch.cuda.current_stream().cuda_stream), "cham_combine_bwd")
        torch.cuda.synchronize()
        outs.append((o_dU.cpu(), o_dV.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "not bit-reproducible"
    g_dU, g_dV = outs[0][0].double().numpy(), outs[0][1].double().numpy()
    assert np.isfinite(g_dU).all() and np.isfinite(g_dV).all()
    assert np.abs(g_dU - dU).max() < 2e-5 * max(1.0, np.abs(dU).max())
    assert np.abs(g_dV - dV).max() < 2e-5 * max(1.0, np.abs(dV).max())
    assert lib.cham_combine_bwd(ptr(d_dpre), C, BT, N, pmax, ptr(d_slot), ptr(o_dU), ptr(o_dV), ptr(ws), 64, None) == -22
