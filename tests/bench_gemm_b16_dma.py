"""Stand-alone throughput of the bf16 CAR GEMMs at the G1 shapes: register-staged (cham_gemm_b16) vs the LDS-DMA core
(cham_gemm_b16_dma).  Not a test: python tests/bench_gemm_b16_dma.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import check, ptr

lib = _lib.load()
gpu = torch.device("cuda:0")
R, C = 256 * 19 * 51, 1024
ws = torch.empty(64 << 20, dtype=torch.float32, device=gpu)
s = torch.cuda.current_stream().cuda_stream


def timeit(run, n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


X = torch.randn(R, C, device=gpu).bfloat16()
D = torch.randn(R, C, device=gpu).bfloat16()
W = torch.randn(C, C, device=gpu).bfloat16()
Y = torch.randn(R, C, device=gpu).bfloat16()
bias = torch.randn(C, device=gpu)
ob = torch.empty(R, C, device=gpu, dtype=torch.bfloat16)
of = torch.empty(C, C, device=gpu)
fl = 2.0 * R * C * C
for name, old, new in (
    ("CAR forward  tanh(X W^T + b)",
     lambda: check(lib.cham_gemm_b16(ptr(X), C, 0, ptr(W), C, 1, ptr(ob), C, 0, R, C, C, ptr(bias), 2, None, 0, 0, 0, None, 0, 1, s), "b16"),
     lambda: check(lib.cham_gemm_b16_dma(ptr(X), C, ptr(W), C, 0, ptr(ob), C, R, C, C, ptr(bias), 2, None, 0, 0, 0, None, 0, 1, s), "dma")),
    ("CAR forward  plain bf16(X W^T)",
     lambda: check(lib.cham_gemm_b16(ptr(X), C, 0, ptr(W), C, 1, ptr(ob), C, 0, R, C, C, None, 0, None, 0, 0, 0, None, 0, 1, s), "b16"),
     lambda: check(lib.cham_gemm_b16_dma(ptr(X), C, ptr(W), C, 0, ptr(ob), C, R, C, C, None, 0, None, 0, 0, 0, None, 0, 1, s), "dma")),
    ("CAR dgrad    (D W) x leaky'",
     lambda: check(lib.cham_gemm_b16(ptr(D), C, 0, ptr(W), C, 1, ptr(ob), C, 0, R, C, C, None, 0, ptr(Y), C, 1, 0, None, 0, 1, s), "b16"),
     lambda: check(lib.cham_gemm_b16_dma(ptr(D), C, ptr(W), C, 0, ptr(ob), C, R, C, C, None, 0, ptr(Y), C, 1, 0, None, 0, 1, s), "dma")),
    ("W2 wgrad     X^T D (split-K)",
     lambda: check(lib.cham_gemm_b16(ptr(X), C, 1, ptr(D), C, 0, ptr(of), C, 1, C, C, R, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4, 0, s), "b16"),
     lambda: check(lib.cham_gemm_b16_dma(ptr(X), C, ptr(D), C, 1, ptr(of), C, C, C, R, None, 0, None, 0, 0, 0, ptr(ws), ws.numel() * 4, int(os.environ.get("SPLITS", "0")), s), "dma")),
):
    a, b = timeit(old), timeit(new)
    print("%-32s register-staged %.3f ms %7.1f TFLOP/s (%.3f of 2500) | LDS-DMA %.3f ms %7.1f TFLOP/s (%.3f)" %
          (name, a, fl / a / 1e9, fl / a / 1e9 / 2500, b, fl / b / 1e9, fl / b / 1e9 / 2500))
