"""Manual tool (not a test): times the fp32 MFMA GEMM tile variants on the three dominant shapes of the G1 step.
python -m tests.bench_gemm"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr, check


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    R, C = 252928, 1024
    A = torch.randn(R, C, device=dev); W = torch.randn(C, C, device=dev) * 0.03; D = torch.randn(R, C, device=dev)
    Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev)
    bias = torch.randn(C, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * R * C * C
    shapes = {
        "NN fwd tanh": lambda: lib.cham_gemm_f32(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, None, 0, 1, 0, None, 0, 1, st),
        "NN fwd none": lambda: lib.cham_gemm_f32(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, None, 0, None, 0, 0, None, 0, 1, 0, None, 0, 1, st),
        "NT dgrad   ": lambda: lib.cham_gemm_f32(ptr(D), C, 0, ptr(W), C, 1, ptr(Out), C, R, C, C, None, 0, ptr(A), C, 1, None, 0, 1, 0, None, 0, 1, st),
        "TN wgrad   ": lambda: lib.cham_gemm_f32(ptr(A), C, 1, ptr(D), C, 0, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, None, 0, 1, 0, ptr(ws), ws.numel() * 4, 0, st),
    }
    names = {0: "128x128x16 4w", 1: "128x128x32 4w", 2: "256x128x16 8w", 3: "256x128x32 8w", 4: "256x256x16 8w", 5: "256x128x16 4w (128x64/wave)", 6: "128x256x16 4w (64x128/wave)", 7: "256x256x16 4w (128x128/wave, ds_read/MFMA interleave hint)"}
    names.update({100 + k: v + " (K-loop software pipeline OFF)" for k, v in list(names.items())})
    for v in [int(x) for x in sys.argv[1:]] or [0, 2, 4]:
        lib.cham_gemm_set_variant(v)
        for name, fn in shapes.items():
            for _ in range(2):
                check(fn(), name)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("variant %d (%s) %s: %.3f ms  %.1f TFLOP/s" % (v, names[v], name, ms, flops / ms / 1e9), flush=True)
    lib.cham_gemm_set_variant(-1)


if __name__ == "__main__":
    main()
