"""Manual A/B tool: times the fp32 GEMM entry point of two builds of gemm.hip (stand-alone .so files, e.g. ab/libgemm_old.so and
ab/libgemm_new.so built with `hipcc -shared` from two revisions) on the G1-step shapes.  python scripts/ab_gemm.py A.so B.so"""
import ctypes
import sys
import torch

R, C = 248064, 1024
dev = torch.device('cuda:0')
vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def load(path):
    lib = ctypes.CDLL(path)
    lib.cham_gemm_f32.restype = ci
    lib.cham_gemm_f32.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp, ci, ci, ci, vp, sz, ci, vp]
    return lib


def main():
    libs = [(p, load(p)) for p in sys.argv[1:]]
    g = torch.Generator(device=dev).manual_seed(0)
    bufs = {}

    def T(name, *shape):
        if name not in bufs:
            bufs[name] = torch.randn(*shape, device=dev, generator=g)
        return bufs[name]
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    # (name, M, N, K, transA, transB, bias?, act, dref?, dact, splits)
    cases = [("CAR fwd NN tanh", R, C, C, 0, 0, 1, 2, 0, 0, 1), ("CAR dgrad NT leaky'", R, C, C, 0, 1, 0, 0, 1, 1, 1),
             ("CAR wgrad TN", C, C, R, 1, 0, 0, 0, 0, 0, 0),
             ("S1 fwd NN K1024 N128", R, 128, C, 0, 0, 1, 1, 0, 0, 1), ("S2 fwd NN K128 N64", R, 64, 128, 0, 0, 1, 1, 0, 0, 1),
             ("S3 fwd NN K64 N32", R, 32, 64, 0, 0, 1, 1, 0, 0, 1),
             ("dS2 NT K32 N64", R, 64, 32, 0, 1, 0, 0, 1, 1, 1), ("dS1 NT K64 N128", R, 128, 64, 0, 1, 0, 0, 1, 1, 1),
             ("dM NT K128 N1024", R, C, 128, 0, 1, 0, 0, 0, 0, 1),
             ("Ws1 wgrad TN 1024x128", C, 128, R, 1, 0, 0, 0, 0, 0, 0), ("Ws2 wgrad TN 128x64", 128, 64, R, 1, 0, 0, 0, 0, 0, 0)]
    for name, M, N, K, tA, tB, hb, act, hd, dact, splits in cases:
        A = T("A%d_%d" % ((K, M) if tA else (M, K)), *((K, M) if tA else (M, K)))
        B = T("B%d_%d" % ((N, K) if tB else (K, N)), *((N, K) if tB else (K, N)))
        bias = T("bias%d" % N, N) if hb else None
        ref = torch.tanh(T("ref%d_%d" % (M, N), M, N)) if hd else None
        outs = []
        line = "%-24s" % name
        for path, lib in libs:
            Cc = torch.zeros(M, N, device=dev)
            fn = lambda: lib.cham_gemm_f32(ptr(A), A.shape[1], tA, ptr(B), B.shape[1], tB, ptr(Cc), N, M, N, K, ptr(bias), act, ptr(ref), N, dact,
                                           None, 0, 1, 0, ptr(ws), ws.numel() * 4, splits, st)
            for _ in range(3):
                rc = fn()
                assert rc == 0, (name, path, rc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            byts = 4.0 * (M * K + K * N + M * N * (2 if hd else 1))
            line += "  %8.3f ms %6.1f TF %6.0f GB/s" % (ms, 2.0 * M * N * K / ms / 1e9, byts / ms / 1e6)
            outs.append(Cc)
        if len(outs) == 2:
            line += "  maxdiff %.2e" % float((outs[0] - outs[1]).abs().max())
        print(line, flush=True)


main()
