"""Manual tool (not a test): the fused scorer layer-1 dgrad + cand (.) pred backward (csrc/dm_fused.hip) next to the two kernels it replaces,
at the G1 shape.  python -m tests.bench_dm_fused"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr, check
from tests.test_gemm_p3_gpu import split3
from tests.bench_gemm_p3 import timed


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    BT, N, C = 256 * 19, 50, 1024
    NC = N + 1; Rc = BT * NC
    g = torch.Generator(device=dev).manual_seed(0)
    dS1 = torch.randn(Rc, 128, device=dev, generator=g); Ws1 = torch.randn(C, 128, device=dev, generator=g) * 0.1
    Z2 = torch.tanh(torch.randn(Rc, C, device=dev, generator=g)); pred = torch.tanh(torch.randn(BT, C, device=dev, generator=g))
    Wp = split3(Ws1)
    planes = torch.empty(3, Rc, C, dtype=torch.bfloat16, device=dev); dpred = torch.empty(BT, C, device=dev); b2p = torch.empty(BT, C, device=dev)
    dM = torch.empty(Rc, C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fused = lambda: check(lib.cham_dm_mulpred_p3(ptr(dS1), 128, 128, ptr(Wp), C * 128, ptr(Z2), ptr(pred), C, BT, N, ptr(planes), Rc * C, ptr(dpred), ptr(b2p), st), "fused")
    gemm = lambda: check(lib.cham_gemm_f32x3(ptr(dS1), 128, 0, ptr(Ws1), 128, 1, ptr(dM), C, Rc, C, 128, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), "x3")
    mulp = lambda: check(lib.cham_mulpred_bwd_p3(ptr(dM), ptr(Z2), ptr(pred), C, BT, N, ptr(dpred), ptr(planes), Rc * C, ptr(b2p), st), "mulpred")
    t_f, t_g, t_m = timed(fused), timed(gemm), timed(mulp)
    by = 4.0 * Rc * 128 + 4.0 * Rc * C + 6.0 * Rc * C
    print("fused k_dm_mulpred_fused: %.3f ms (%.2f TB/s of its %.2f GB, %.1f TFLOP/s of fp32 work)" % (t_f, by / t_f / 1e9, by / 1e9, 2.0 * Rc * C * 128 / t_f / 1e9))
    print("unfused: dM GEMM (x3) %.3f ms + k_mulpred_bwd_p3 %.3f ms = %.3f ms" % (t_g, t_m, t_g + t_m))


if __name__ == "__main__":
    main()
