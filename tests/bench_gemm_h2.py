"""Manual tool (not a test): the two-fp16-plane GEMM (csrc/gemm_h2.hip, three plane products) next to the three-bf16-plane one
(csrc/gemm_p3.hip, six products) at the three CAR shapes of the G1 step - time, fp32-equivalent TFLOP/s, fraction of the plane-product
ceilings (2500 / 3 and 2500 / 6 TFLOP/s), error against float64 next to the native fp32 MFMA's.  python -m tests.bench_gemm_h2 [rows]
H2_ONLY=1: only the h2 kernels, few launches (PMC passes: scripts/h2_pmc.sh).  H2_NT_WIDE=0: the NT forms on round 4's 32-byte-piece kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr, check
from tests.test_gemm_p3_gpu import split3
from tests.test_gemm_h2_gpu import split2h_dev


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
    lib.cham_gemm_h2_set_nt_wide(int(os.environ.get("H2_NT_WIDE", "1")))       # NT forms: 64-byte-source-piece kernel (1, default) or round 4's (0)
    dev = torch.device("cuda:0")
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 248064
    C = 1024
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(R, C, device=dev, generator=g); W = torch.randn(C, C, device=dev, generator=g) * 0.03
    D = torch.randn(R, C, device=dev, generator=g) * torch.exp2(-20 * torch.rand(R, 1, device=dev, generator=g))
    Y = torch.randn(R, C, device=dev, generator=g)
    bias = torch.randn(C, device=dev, generator=g)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev)
    only = os.environ.get("H2_ONLY")
    (Ah, ra), (Dh, rd) = split2h_dev(A), split2h_dev(D)
    Wh, WTh, rw = split2h_dev(W, transposed=True)
    Yh2 = split2h_dev(Y)[0][0].contiguous()
    if not only:
        Ap, Dp, Wp, WTp, Yh = split3(A), split3(D), split3(W), split3(W.t().contiguous()), Y.to(torch.bfloat16)
    rows = torch.arange(0, R, max(1, R // 2048), device=dev)[:2048]
    flops = 2.0 * R * C * C

    def h2(Ap_, a_ps, lda, ra_, Bp_, b_ps, ldb, rb_, tn, Cc, M, N, K, bias_=None, act=0, dref=None, dact=0, splits=1):
        return lambda: check(lib.cham_gemm_h2(ptr(Ap_), a_ps, lda, ptr(ra_), ptr(Bp_), b_ps, ldb, ptr(rb_), tn, ptr(Cc), N, M, N, K, ptr(bias_), act, ptr(dref), N, dact,
                                              0, ptr(ws) if splits != 1 else None, ws.numel() * 4 if splits != 1 else 0, splits, st), "h2")

    def native(A_, tA, B_, tB, Cc, M, N, K, bias_=None, act=0, dref=None, dact=0, splits=1):
        return lambda: check(lib.cham_gemm_f32(ptr(A_), A_.shape[1], tA, ptr(B_), B_.shape[1], tB, ptr(Cc), N, M, N, K, ptr(bias_), act, ptr(dref), N, dact, None, 0, 1, 0,
                                               ptr(ws) if splits != 1 else None, ws.numel() * 4 if splits != 1 else 0, splits, st), "native")
    cases = [
        ("CAR fwd  tanh(A W + b)",
         h2(Ah, R * C, C, ra, WTh, C * C, C, rw, 0, Out, R, C, C, bias_=bias, act=2),
         None if only else (lambda: check(lib.cham_gemm_p3(ptr(Ap), R * C, C, ptr(WTp), C * C, C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, 0, None, 0, 1, st), "p3")),
         native(A, 0, W, 0, Out, R, C, C, bias_=bias, act=2),
         lambda: Out[rows], lambda: torch.tanh(A[rows].double() @ W.double() + bias.double())),
        ("CAR fwd, bias only (no tanh)",
         h2(Ah, R * C, C, ra, WTh, C * C, C, rw, 0, Out, R, C, C, bias_=bias, act=0), None, native(A, 0, W, 0, Out, R, C, C, bias_=bias, act=0),
         lambda: Out[rows], lambda: A[rows].double() @ W.double() + bias.double()),
        ("CAR dgrad (D W^T) leaky'",
         h2(Dh, R * C, C, rd, Wh, C * C, C, rw, 0, Out, R, C, C, dref=Yh2, dact=1),
         None if only else (lambda: check(lib.cham_gemm_p3(ptr(Dp), R * C, C, ptr(Wp), C * C, C, 0, ptr(Out), C, R, C, C, None, 0, ptr(Yh), C, 1, 0, None, 0, 1, st), "p3")),
         native(D, 0, W, 1, Out, R, C, C, dref=Y, dact=1),
         lambda: Out[rows], lambda: (D[rows].double() @ W.double().t()) * torch.where(Y[rows].double() > 0, 1.0, 0.2)),
        ("W2 wgrad A^T D split-K",
         h2(Ah, R * C, C, ra, Dh, R * C, C, rd, 1, Wg, C, C, R, splits=0),
         None if only else (lambda: check(lib.cham_gemm_p3(ptr(Ap), R * C, C, ptr(Dp), R * C, C, 1, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4, 0, st), "p3")),
         native(A, 1, D, 0, Wg, C, C, R, splits=0),
         lambda: Wg, lambda: A.double().t() @ D.double()),
    ]
    for name, f_h2, f_p3, f_nat, out, ref in cases:
        Rf = ref(); scale = float(Rf.abs().max())
        for tag, fn, peak in (("two fp16 planes (h2)", f_h2, 2500.0 / 3), ("three bf16 planes (p3)", f_p3, 2500.0 / 6), ("native fp32 MFMA", f_nat, 157.3)):
            if fn is None or (only and tag != "two fp16 planes (h2)"):
                continue
            out().zero_()
            fn(); torch.cuda.synchronize()
            err = float((out().double() - Rf).abs().max()) / scale
            ms = timed(fn, 3 if only else 10)
            tf = flops / ms / 1e9
            print("%-26s %-24s: %7.3f ms %6.1f TFLOP/s (%.3f of %.1f)  max err / max|ref| %.2e" % (name, tag, ms, tf, tf / peak, peak, err), flush=True)
    if only:
        return
    # the producers' side: splitting a [R, C] matrix once as its own pass (what the fused producers add to their stores: 4 B per element)
    P = torch.empty(2, R, C, dtype=torch.float16, device=dev)
    rec = torch.zeros(8, device=dev)
    ms = timed(lambda: check(lib.cham_split2h(ptr(A), R, C, C, ptr(P), R * C, C, None, 0, 0, ptr(rec), 1, st), "split2h"), 5)
    print("cham_split2h of [%d, %d] (max pass + split pass: read 8 B + write 4 B per element): %.3f ms = %.2f TB/s" % (R, C, ms, R * C * 12 / ms / 1e9), flush=True)
    dS1 = torch.randn(R, 128, device=dev, generator=g)
    ms = timed(lambda: check(lib.cham_h2_scale_rownorm(ptr(dS1), R, 128, 128, None, ptr(rec), st), "rownorm"), 10)
    print("cham_h2_scale_rownorm of [%d, 128]: %.4f ms = %.2f TB/s" % (R, ms, R * 128 * 4 / ms / 1e9), flush=True)
    U = torch.randn(4864, C, device=dev, generator=g); V = torch.randn(10729, C, device=dev, generator=g)
    ms = timed(lambda: check(lib.cham_h2_scale_absmax(ptr(U), U.numel(), ptr(V), V.numel(), ptr(rec), st), "absmax"), 10)
    print("cham_h2_scale_absmax of U [4864, 1024] + V [10729, 1024]: %.4f ms" % ms, flush=True)


if __name__ == "__main__":
    main()
