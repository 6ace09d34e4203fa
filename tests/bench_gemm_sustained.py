"""Manual tool: sustained fwd/dgrad/wgrad rotation of the CAR layer-2 GEMMs (does the short-burst TFLOP/s of
tests/bench_gemm.py survive seconds of continuous load?).  Prints per-window rates and the sclk sampled by rocm-smi."""
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    R, C = 252928, 1024
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    A = torch.randn(R, C, device=dev) * scale; W = torch.randn(C, C, device=dev) * 0.03; D = torch.randn(R, C, device=dev) * scale
    Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev); bias = torch.randn(C, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * R * C * C
    fns = [
        ("fwd", lambda: lib.cham_gemm_f32(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, None, 0, 1, 0, None, 0, 1, st)),
        ("dgrad", lambda: lib.cham_gemm_f32(ptr(D), C, 0, ptr(W), C, 1, ptr(Out), C, R, C, C, None, 0, ptr(A), C, 1, None, 0, 1, 0, None, 0, 1, st)),
        ("wgrad", lambda: lib.cham_gemm_f32(ptr(A), C, 1, ptr(D), C, 0, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, None, 0, 1, 0, ptr(ws), ws.numel() * 4, 0, st)),
    ]
    stop = [False]
    clocks = []

    def sample():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                s = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l]
                clocks.append((time.time(), " | ".join(x.split(":", 1)[-1].strip() if False else x[-60:] for x in s)))
            except Exception as e:
                clocks.append((time.time(), repr(e)))
            time.sleep(0.3)
    th = threading.Thread(target=sample); th.start()
    t0 = time.time()
    for w in range(12):
        evs = []
        for name, fn in fns:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            evs.append((name, e0, e1))
        torch.cuda.synchronize()
        print("t=%.2fs " % (time.time() - t0) + "  ".join("%s %.1f" % (n, flops / (a.elapsed_time(b) / 8) / 1e9) for n, a, b in evs), flush=True)
    stop[0] = True; th.join()
    for t, c in clocks:
        print("smi t=%.2f %s" % (t - t0, c))


if __name__ == "__main__":
    main()
