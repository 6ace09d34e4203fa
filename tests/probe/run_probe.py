"""Manual tool: which part of the GEMM K loop costs what?  Builds tests/probe/libprobe.so from probe_gemm.hip (hipcc) and
times the NN 256x128x16 kernel with parts ablated (results in profiles/r01_notes.md).
ABL bits: 1 no global loads / LDS writes in the loop, 2 no barrier, 4 no LDS fragment reads, 8 minimal epilogue."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
LIB = os.path.join(HERE, "libprobe.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "../../chameleon_recsys_amd/csrc/gemm.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-I", os.path.join(HERE, "../../chameleon_recsys_amd/csrc"), os.path.join(HERE, "probe_gemm.hip"), "-o", LIB])


def main(libpath=LIB):
    import torch
    print("==", os.path.basename(libpath), flush=True)
    lib = ctypes.CDLL(libpath)
    lib.cham_gemm_probe.restype = ctypes.c_int
    lib.cham_gemm_probe.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    M, N = 248064 // 256 * 256, 1024
    for K in (1024,):
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda") * 0.03; C = torch.empty(M, N, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for abl, name in [(0, "full kernel"), (8, "minimal epilogue"), (9, "+ no global loads / LDS writes in loop"),
                          (11, "+ no barrier"), (15, "+ no LDS fragment reads (pure MFMA issue)"), (1, "no loads, full epilogue"),
                          (24, "min. epilogue, loads kept, NO LDS writes"), (40, "min. epilogue, LDS writes kept, NO loads")]:
            for _ in range(2):
                assert lib.cham_gemm_probe(abl, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.cham_gemm_probe(abl, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("K=%5d ABL=%2d %-48s %.3f ms  %.1f TFLOP/s" % (K, abl, name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        del A, B, C


if __name__ == "__main__":
    build()
    if "--build-only" not in sys.argv:
        main()
        alt = os.path.join(HERE, "libprobe_b128.so")
        if os.path.exists(alt):
            main(alt)
