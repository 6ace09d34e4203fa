"""Stand-alone throughput of the bf16-resident GEMMs at the G1 step's shapes (not a test): python tests/bench_gemm_b16.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import check, ptr

lib = _lib.load()
gpu = torch.device("cuda:0")
R = 256 * 19 * 51
ws = torch.empty(64 << 20, dtype=torch.float32, device=gpu)


def bench(name, M, N, K, mode, variant, **kw):
    nt = mode == 'nt'
    A = torch.randn((M, K) if nt else (K, M), device=gpu).bfloat16()
    B = torch.randn((N, K) if nt else (K, N), device=gpu).bfloat16()
    out_f32 = kw.get('out_f32', 0)
    C = torch.empty(M, N, device=gpu, dtype=torch.float32 if out_f32 else torch.bfloat16)
    bias = torch.randn(N, device=gpu) if kw.get('bias') else None
    dref = torch.randn(M, N, device=gpu).bfloat16() if kw.get('dref') else None
    s = torch.cuda.current_stream().cuda_stream
    lib.cham_gemm_b16_set_variant(variant)

    def run():
        check(lib.cham_gemm_b16(ptr(A), A.shape[1], 0 if nt else 1, ptr(B), B.shape[1], 1 if nt else 0, ptr(C), N, out_f32, M, N, K,
                                ptr(bias), kw.get('act', 0), ptr(dref), N, kw.get('dact', 0), 0, ptr(ws), ws.numel() * 4,
                                kw.get('splits', 1), s), name)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = 2.0 * (M * K + N * K) + (4.0 if out_f32 else 2.0) * M * N + (2.0 * M * N if dref is not None else 0)
    print("%-34s v%d  %8.3f ms  %7.1f TFLOP/s  %6.2f TB/s (algorithmic)" % (name, variant, ms, 2.0 * M * N * K / ms / 1e9, by / ms / 1e9), flush=True)
    lib.cham_gemm_b16_set_variant(-1)


variants = [int(x) for x in os.environ.get("B16_VARIANTS", "0,1,2").split(",")]
for v in variants:
    bench("CAR fwd NT bias+tanh", R, 1024, 1024, 'nt', v, bias=True, act=2)
    bench("CAR dgrad NT x leaky'", R, 1024, 1024, 'nt', v, dref=True, dact=1)
    bench("W2 wgrad TN split-K", 1024, 1024, R, 'tn', v, out_f32=1, splits=0)
    if v <= 2:
        bench("scorer L1 fwd NT K=1024 N=128", R, 128, 1024, 'nt', v, bias=True, act=1)
        bench("scorer L1 dgrad NT K=128", R, 1024, 128, 'nt', v)
        bench("Ws1 wgrad TN", 1024, 128, R, 'tn', v, out_f32=1, splits=0)
if os.environ.get("B16_SMALL", "1") == "1":
    bench("scorer L2 fwd NT N=64", R, 64, 128, 'nt', -1, bias=True, act=1)
    bench("scorer L3 fwd NT N=32", R, 32, 64, 'nt', -1, bias=True, act=1)
    bench("scorer L2 dgrad", R, 128, 64, 'nt', -1, dref=True, dact=1)
    bench("scorer L3 dgrad", R, 64, 32, 'nt', -1, dref=True, dact=1)
