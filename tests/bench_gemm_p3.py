"""Manual tool (not a test): the plane-product GEMM over pre-split planes (csrc/gemm_p3.hip) next to the on-the-fly split (gemm_x3.hip)
at the three CAR shapes of the G1 step - time, fp32-equivalent TFLOP/s, fraction of the 416.7 TFLOP/s plane-product ceiling, error against
float64.  python -m tests.bench_gemm_p3 [rows]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr, check
from tests.test_gemm_p3_gpu import split3


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 248064
    C = 1024
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(R, C, device=dev, generator=g); W = torch.randn(C, C, device=dev, generator=g) * 0.03
    D = torch.randn(R, C, device=dev, generator=g); Y = torch.randn(R, C, device=dev, generator=g)
    bias = torch.randn(C, device=dev, generator=g)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev)
    Ap, Dp, Wp, WTp, Yh = split3(A), split3(D), split3(W), split3(W.t().contiguous()), Y.to(torch.bfloat16)
    Yf = Yh.float()
    rows = torch.arange(0, R, max(1, R // 2048), device=dev)[:2048]
    flops = 2.0 * R * C * C
    cases = [
        ("CAR fwd  tanh(A W + b)",
         lambda: check(lib.cham_gemm_p3(ptr(Ap), R * C, C, ptr(WTp), C * C, C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, 0, None, 0, 1, st), "p3"),
         lambda: check(lib.cham_gemm_f32x3(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), "x3"),
         lambda: Out[rows], lambda: torch.tanh(A[rows].double() @ W.double() + bias.double())),
        ("CAR dgrad (D W^T) leaky'",
         lambda: check(lib.cham_gemm_p3(ptr(Dp), R * C, C, ptr(Wp), C * C, C, 0, ptr(Out), C, R, C, C, None, 0, ptr(Yh), C, 1, 0, None, 0, 1, st), "p3"),
         lambda: check(lib.cham_gemm_f32x3(ptr(D), C, 0, ptr(W), C, 1, ptr(Out), C, R, C, C, None, 0, ptr(Yf), C, 1, None, 0, 1, 0, None, 0, 1, st), "x3"),
         lambda: Out[rows], lambda: (D[rows].double() @ W.double().t()) * torch.where(Yh[rows].double() > 0, 1.0, 0.2)),
        ("W2 wgrad A^T D split-K",
         lambda: check(lib.cham_gemm_p3(ptr(Ap), R * C, C, ptr(Dp), R * C, C, 1, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4, 0, st), "p3"),
         lambda: check(lib.cham_gemm_f32x3(ptr(A), C, 1, ptr(D), C, 0, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, None, 0, 1, 0, ptr(ws), ws.numel() * 4, 0, st), "x3"),
         lambda: Wg, lambda: A.double().t() @ D.double()),
    ]
    res = {}
    only = os.environ.get("P3_ONLY")          # PMC passes (scripts/p3_pmc.sh): only the default plane kernel, few launches
    for name, p3, x3, out, ref in cases:
        Rf = ref(); scale = float(Rf.abs().max())
        for tag, fn in (("planes in HBM (p3)", p3), ("split on the fly (x3)", x3)):
            if only and tag != "planes in HBM (p3)":
                continue
            out().zero_()
            fn(); torch.cuda.synchronize()
            err = float((out().double() - Rf).abs().max()) / scale
            ms = timed(fn, 3 if only else 10)
            tf = flops / ms / 1e9
            print("%-26s %-22s: %7.3f ms %6.1f TFLOP/s (%.3f of 416.7)  max err / max|ref| %.2e" % (name, tag, ms, tf, tf / 416.7, err), flush=True)
            res[(name, tag)] = ms
    if only:
        return
    # the producers' side of the bargain: splitting a [R, C] matrix once (what k_combine_fwd_cand / k_mulpred_bwd add to their stores)
    P = torch.empty(3, R, C, dtype=torch.bfloat16, device=dev)
    ms = timed(lambda: check(lib.cham_split3(ptr(A), R, C, C, ptr(P), R * C, C, None, 0, 0, st), "split3"), 5)
    print("cham_split3 of [%d, %d] (stand-alone pass: read 4 B + write 6 B per element): %.3f ms = %.2f TB/s" % (R, C, ms, R * C * 10 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
