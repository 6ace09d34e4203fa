"""Ablation probe runner (see scripts/ab_x3.sh): times the NT dgrad / NN fwd / TN wgrad CAR shapes on every ab/libx3_<bits>.so."""
import ctypes, glob, re, sys
import torch
R, C = 248064, 1024
dev = torch.device('cuda:0')
vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
A = torch.randn(R, C, device=dev); W = torch.randn(C, C, device=dev); Y = torch.randn(R, C, device=dev); bias = torch.randn(C, device=dev)
Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev); ws = torch.empty(64 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream
libs = sorted(glob.glob('ab/libx3_*.so'), key=lambda p: int(re.findall(r'_(\d+)\.so', p)[0]))
for path in libs:
    lib = ctypes.CDLL(path)
    f = lib.cham_gemm_f32x3
    f.restype = ci
    f.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp, ci, ci, ci, vp, sz, ci, vp]
    lib.cham_gemm_f32x3_set_variant.argtypes = [ci]
    line = "ABL %3s" % re.findall(r'_(\d+)\.so', path)[0]
    for variant in [int(x) for x in (sys.argv[1:] or ['2'])]:
        lib.cham_gemm_f32x3_set_variant(variant)
        cases = {"NT": lambda: f(ptr(A), C, 0, ptr(W), C, 1, ptr(Out), C, R, C, C, None, 0, ptr(Y), C, 1, None, 0, 1, 0, None, 0, 1, st),
                 "NN": lambda: f(ptr(A), C, 0, ptr(W), C, 0, ptr(Out), C, R, C, C, ptr(bias), 2, None, 0, 0, None, 0, 1, 0, None, 0, 1, st),
                 "TN": lambda: f(ptr(A), C, 1, ptr(Y), C, 0, ptr(Wg), C, C, C, R, None, 0, None, 0, 0, None, 0, 1, 0, ptr(ws), ws.numel() * 4, 0, st)}
        for name, fn in cases.items():
            for _ in range(2):
                assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            line += "  v%d %s %6.3f ms" % (variant, name, e0.elapsed_time(e1) / 5)
    print(line, flush=True)
