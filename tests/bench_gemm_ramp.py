"""Manual tool: does a memory-bound phase slow the GEMM launches that follow it?  Alternates an HBM-bound kernel phase of X ms
with a burst of NN-tanh GEMM launches and prints the duration of every launch of the burst (HIP events)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    R, C = 123904, 1024
    A = torch.randn(R, C, device=dev); W = torch.randn(C, C, device=dev) * 0.03
    Out = torch.empty(R, C, device=dev); bias = torch.randn(C, device=dev)
    big = torch.empty(1 << 28, device=dev)            # 1 GB scratch for the memory-bound phase
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * R * C * C

    def gemm():
        lib.cham_gemm_f32(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, None, 0, 1, 0, None, 0, 1, st)
    for _ in range(40):
        gemm()
    torch.cuda.synchronize()
    for mem_iters, label in [(0, "no memory phase"), (1, "1 x fill 1 GB (~0.25 ms)"), (8, "8 x fill (~2 ms)"), (40, "40 x fill (~10 ms)")]:
        rows = []
        for rep in range(6):
            for _ in range(mem_iters):
                big.add_(1.0)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            evs[0].record()
            for i in range(8):
                gemm(); evs[i + 1].record()
            torch.cuda.synchronize()
            rows.append([evs[i].elapsed_time(evs[i + 1]) for i in range(8)])
        rows = rows[2:]
        avg = [sum(r[i] for r in rows) / len(rows) for i in range(8)]
        print("%-28s TFLOP/s per launch after the phase: %s" % (label, " ".join("%.1f" % (flops / ms / 1e9) for ms in avg)), flush=True)
    # sleep (idle) instead of memory phase
    import time
    rows = []
    for rep in range(5):
        torch.cuda.synchronize(); time.sleep(0.02)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        evs[0].record()
        for i in range(8):
            gemm(); evs[i + 1].record()
        torch.cuda.synchronize()
        rows.append([evs[i].elapsed_time(evs[i + 1]) for i in range(8)])
    avg = [sum(r[i] for r in rows) / len(rows) for i in range(8)]
    print("%-28s TFLOP/s per launch after the phase: %s" % ("20 ms idle", " ".join("%.1f" % (flops / ms / 1e9) for ms in avg)), flush=True)


if __name__ == "__main__":
    main()
