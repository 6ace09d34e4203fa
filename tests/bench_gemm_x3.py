"""Manual tool (not a test): the bf16x3 fp32 GEMM (csrc/gemm_x3.hip) against the native fp32 MFMA GEMM on the G1-step shapes -
time per tile variant and error against float64 on the same operands.  python -m tests.bench_gemm_x3 [rows]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr, check


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 248064
    C = 1024
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(R, C, device=dev, generator=g); W = torch.randn(C, C, device=dev, generator=g) * 0.03
    D = torch.randn(R, C, device=dev, generator=g); Y = torch.tanh(torch.randn(R, C, device=dev, generator=g))
    Ws1 = torch.randn(C, 128, device=dev, generator=g) * 0.03; D1 = torch.randn(R, 128, device=dev, generator=g)
    bias = torch.randn(C, device=dev, generator=g)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev); O1 = torch.empty(R, 128, device=dev); Wg1 = torch.empty(C, 128, device=dev)
    # name -> (call(fn), flops, output tensor, float64 reference on a row sample)
    rows = torch.arange(0, R, max(1, R // 2048), device=dev)[:2048]

    def ref_fwd():
        return torch.tanh(A[rows].double() @ W.double() + bias.double())

    def ref_dgrad():
        return (D[rows].double() @ W.double().t()) * torch.where(Y[rows].double() > 0, 1.0, 0.2)

    def ref_wgrad():
        return A.double().t() @ D.double()

    def ref_s1():
        v = A[rows].double() @ Ws1.double() + bias[:128].double()
        return torch.where(v > 0, v, 0.2 * v)

    def ref_dm():
        return D1[rows].double() @ Ws1.double().t()

    def ref_wg1():
        return A.double().t() @ D1.double()
    cases = {
        "CAR fwd NN tanh": (lambda f: f(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), 2.0 * R * C * C, lambda: Out[rows], ref_fwd),
        "CAR dgrad NT leaky'": (lambda f: f(ptr(D), C, 0, ptr(W), C, 1, ptr(Out), C, R, C, C, None, 0, ptr(Y), C, 1, None, 0, 1, 0, None, 0, 1, st), 2.0 * R * C * C, lambda: Out[rows], ref_dgrad),
        "CAR wgrad TN splitK": (lambda f: f(ptr(A), C, 1, ptr(D), C, 0, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, None, 0, 1, 0, ptr(ws), ws.numel() * 4, 0, st), 2.0 * R * C * C, lambda: Wg, ref_wgrad),
        "S1 fwd NN N128": (lambda f: f(ptr(A), C, 0, ptr(Ws1), 128, 0, ptr(O1), 128, R, 128, C, ptr(bias), 1, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), 2.0 * R * C * 128, lambda: O1[rows], ref_s1),
        "dM NT K128": (lambda f: f(ptr(D1), 128, 0, ptr(Ws1), 128, 1, ptr(Out), C, R, C, 128, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1, st), 2.0 * R * C * 128, lambda: Out[rows], ref_dm),
        "Ws1 wgrad TN": (lambda f: f(ptr(A), C, 1, ptr(D1), 128, 0, ptr(Wg1), 128, C, 128, R, None, 0, None, 0, 0, None, 0, 1, 0, ptr(ws), ws.numel() * 4, 0, st), 2.0 * R * C * 128, lambda: Wg1, ref_wg1),
    }
    variants = [int(x) for x in os.environ.get("X3_VARIANTS", "-1,0,2").split(",")]
    only = os.environ.get("X3_ONLY")
    for name, (call, flops, out, ref) in cases.items():
        if only and not name.startswith("CAR dgrad"):
            continue
        Rf = ref()
        scale = float(Rf.abs().max())
        line = []
        check(call(lib.cham_gemm_f32), name)
        torch.cuda.synchronize()
        e_native = float((out().double() - Rf).abs().max()) / scale
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call(lib.cham_gemm_f32)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%-22s native fp32 MFMA  : %7.3f ms %6.1f TF  max err / max|ref| %.2e" % (name, ms, flops / ms / 1e9, e_native), flush=True)
        for v in variants:
            lib.cham_gemm_f32x3_set_variant(v)
            out().zero_()
            rc = call(lib.cham_gemm_f32x3)
            if rc != 0:
                print("%-22s x3 variant %2d: rc %d" % (name, v, rc)); continue
            for _ in range(2):
                call(lib.cham_gemm_f32x3)
            torch.cuda.synchronize()
            err = float((out().double() - Rf).abs().max()) / scale
            e0.record()
            for _ in range(5):
                call(lib.cham_gemm_f32x3)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("%-22s bf16x3 variant %2d  : %7.3f ms %6.1f TF  max err / max|ref| %.2e" % (name, v, ms, flops / ms / 1e9, err), flush=True)
        lib.cham_gemm_f32x3_set_variant(-1)


if __name__ == "__main__":
    main()
