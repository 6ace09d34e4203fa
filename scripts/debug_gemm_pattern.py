"""Debug aid: error pattern of one big GEMM case per tile variant (which rows / columns of the 256-row tile are wrong)."""
import sys
import torch
sys.path.insert(0, '.')
from tests.test_gemm_gpu import _BigCase, ROWS
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import check, ptr

lib = _lib.load()
gpu = torch.device('cuda:0')


def pattern(case, variant, bf16=False):
    d = case.dev
    C = torch.full((case.M, case.N), float('nan'), device=gpu)
    lib.cham_gemm_set_variant(variant)
    rc = (lib.cham_gemm_bf16 if bf16 else lib.cham_gemm_f32)(
        ptr(d['A']), case.A.shape[1], case.tA, ptr(d['B']), case.B.shape[1], case.tB, ptr(C), case.N, case.M, case.N, case.K,
        ptr(d['bias']), case.act, ptr(d['ref']), case.N, case.dact, None, case.A.shape[1], 1, 0, None, 0, 1,
        torch.cuda.current_stream().cuda_stream)
    lib.cham_gemm_set_variant(-1)
    check(rc, 'gemm')
    torch.cuda.synchronize()
    R = case.reference(bf16).float()
    E = (C.cpu() - R).abs()
    bad = E > 1e-2 * max(1.0, float(R.abs().max())) * 0.05
    nb = int(bad.sum())
    print("variant %d bf16 %d: max err %.3g, wrong %d of %d (%.4f)" % (variant, bf16, float(E.max()), nb, bad.numel(), nb / bad.numel()))
    if nb:
        r, c = bad.nonzero(as_tuple=True)
        print("  rows mod 256 hist (16 bins):", torch.bincount((r % 256) // 16, minlength=16).tolist())
        print("  cols mod 256 hist (16 bins):", torch.bincount((c % 256) // 16, minlength=16).tolist())
        print("  cols mod 32 hist:", torch.bincount(c % 32, minlength=32).tolist())
        print("  rows mod 32 hist:", torch.bincount(r % 32, minlength=32).tolist())
        print("  row tile hist (first 8 / last 8):", torch.bincount(r // 256, minlength=273)[:8].tolist(), torch.bincount(r // 256, minlength=273)[-8:].tolist())
        i = 0
        print("  first wrong: row %d col %d got %.4f want %.4f ratio %.4f" % (r[i], c[i], C[r[i], c[i]], R[r[i], c[i]], C[r[i], c[i]] / R[r[i], c[i]]))
        ratio = (C.cpu()[bad] / R[bad])
        print("  ratio quantiles:", torch.quantile(ratio[:100000].float(), torch.tensor([0.01, 0.25, 0.5, 0.75, 0.99])).tolist())


M = int(sys.argv[1]) if len(sys.argv) > 1 else ROWS
for dref in (False, True):
    case = _BigCase(gpu, M, 1024, 1024, transB=1, dref=dref, dact=1, seed=2)
    print("== NT dref=%s M=%d" % (dref, M))
    for v in (0, 2, 4):
        pattern(case, v)
    pattern(case, 2, bf16=True)
