"""cham_scorer_tail_fused (csrc/scorer.hip, round 5): scorer layers 2-4 + softmax(/tau) + masked NLL + their backward down to the gradient at the
layer-1 output in one launch (nar_model.py:452-473, 511-517, 639-667 and the autodiff of those lines) against a float64 torch restatement of
the same lines - every output the rest of the step reads (S2, S3, logits, probs, nll, ds, dS3, dS2, dS1), 1 + N = 51 (G1), 101 (Adressa: two rows
per lane), 201 (configs[4]: four), positions that do not fill the last workgroup, masked (padded) positions, repeated launches bit-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(S1, W2, b2, W3, b3, w4, b4, BT, NC, tau, sum_mask, mask):
    S1 = S1.double().requires_grad_(True)
    leaky = lambda x: torch.nn.functional.leaky_relu(x, 0.2)
    S2 = leaky(S1 @ W2.double() + b2.double()); S2.retain_grad()
    S3 = leaky(S2 @ W3.double() + b3.double()); S3.retain_grad()
    logits = (S3 @ w4.double() + b4.double()).view(BT, NC); logits.retain_grad()
    logp = torch.log_softmax(logits / tau, dim=1)
    nll = -logp[:, 0] * mask.double()
    (nll.sum() / sum_mask).backward()
    return dict(S2=S2.detach(), S3=S3.detach(), logits=logits.detach(), probs=torch.softmax(logits.detach() / tau, 1), nll=nll.detach(),
                ds=logits.grad.reshape(-1), dS3=S3.grad, dS2=S2.grad, dS1=S1.grad)


@pytest.mark.parametrize("BT,N", [(37, 50), (64, 50), (9, 100), (6, 200), (1, 3)])
def test_scorer_tail_fused_matches_float64(gpu, BT, N):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(BT * 1000 + N)
    NC, R = N + 1, BT * (N + 1)
    rnd = lambda *s, sc=1.0: torch.randn(*s, device=gpu, generator=g) * sc
    S1 = torch.nn.functional.leaky_relu(rnd(R, 128), 0.2)              # a layer-1 OUTPUT (post-leaky: its sign is what leaky' reads)
    W2, b2, W3, b3, w4, b4 = rnd(128, 64, sc=0.12), rnd(64, sc=0.05), rnd(64, 32, sc=0.17), rnd(32, sc=0.05), rnd(32, sc=0.3), rnd(1, sc=0.05)
    mask = (torch.rand(BT, device=gpu, generator=g) > 0.25).to(torch.uint8)
    mask[0] = 1
    tau, sum_mask = 0.1, float(mask.sum()) + 3.0                        # (+ 3: the GLOBAL denominator of a data-parallel shard is not the local count)
    out = {k: torch.full(s, float('nan'), device=gpu) for k, s in dict(S2=(R, 64), S3=(R, 32), logits=(BT, NC), probs=(BT, NC), nll=(BT,), ds=(R,),
                                                                       dS3=(R, 32), dS2=(R, 64), dS1=(R, 128)).items()}
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: check(lib.cham_scorer_tail_fused(ptr(S1), 128, ptr(W2), ptr(b2), 64, ptr(W3), ptr(b3), 32, ptr(w4), ptr(b4), BT, N, tau, sum_mask,
                                                   ptr(mask), ptr(out['S2']), ptr(out['S3']), ptr(out['logits']), ptr(out['probs']), ptr(out['nll']),
                                                   ptr(out['ds']), ptr(out['dS3']), ptr(out['dS2']), ptr(out['dS1']), st), "cham_scorer_tail_fused")
    run(); torch.cuda.synchronize()
    first = {k: v.clone() for k, v in out.items()}
    run(); torch.cuda.synchronize()
    assert all(torch.equal(first[k], out[k]) for k in out), "not repeatable"
    ref = _ref(S1, W2, b2, W3, b3, w4, b4, BT, NC, tau, sum_mask, mask)
    for k, v in out.items():
        r = ref[k].reshape(v.shape)
        assert torch.isfinite(v).all(), k
        err = float((v.double() - r).abs().max()) / max(1e-30, float(r.abs().max()))
        assert err < (2e-5 if k in ("logits", "probs", "nll", "S2", "S3") else 5e-5), (k, err)
    assert float(out['ds'].view(BT, NC)[mask == 0].abs().max() if (mask == 0).any() else 0.0) == 0.0          # padded positions: exactly zero gradient
    assert float(out['dS1'].view(BT, NC * 128)[mask == 0].abs().max() if (mask == 0).any() else 0.0) == 0.0


def test_scorer_tail_fused_argument_errors(gpu):
    from chameleon_recsys_amd import _lib
    from chameleon_recsys_amd._lib import ptr
    lib = _lib.load()
    x = torch.zeros(1024, device=gpu)
    m = torch.ones(4, dtype=torch.uint8, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda K1, K2, K3, tau, sm: lib.cham_scorer_tail_fused(ptr(x), K1, ptr(x), ptr(x), K2, ptr(x), ptr(x), K3, ptr(x), ptr(x), 1, 3, tau, sm, ptr(m),
                                                                  ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), st)
    assert call(128, 64, 32, 0.1, 1.0) == 0
    assert call(128, 64, 16, 0.1, 1.0) < 0 and call(256, 64, 32, 0.1, 1.0) < 0 and call(128, 64, 32, 0.0, 1.0) < 0 and call(128, 64, 32, 0.1, 0.0) < 0
