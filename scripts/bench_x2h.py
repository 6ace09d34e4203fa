"""Manual tool: the scorer's first layer ([248 064 candidate rows, 1024] (.) pred -> 128, bias + leaky-ReLU) and its weight gradient on the
six-product bf16x3 kernel (cham_gemm_f32x3) next to the three-product two-fp16-plane form (cham_gemm_f32x2h) - stand-alone times."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import ptr, check


def timed(fn, n=20):
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
    dev = torch.device("cuda:0")
    BT, NC, C = 4864, 51, 1024
    R = BT * NC
    g = torch.Generator(device=dev).manual_seed(0)
    Z2 = torch.tanh(torch.randn(R, C, device=dev, generator=g)); pred = torch.tanh(torch.randn(BT, C, device=dev, generator=g))
    Ws1 = torch.randn(C, 128, device=dev, generator=g) * 0.03; bs1 = torch.randn(128, device=dev, generator=g) * 0.01
    dS1 = torch.randn(R, 128, device=dev, generator=g) * 1e-4
    S1 = torch.empty(R, 128, device=dev); gW = torch.empty(C, 128, device=dev)
    ws = torch.empty(32 << 20, dtype=torch.float32, device=dev)
    unit = torch.tensor([16384.0, 1 / 16384.0, 1.0, 0, 0, 0, 0, 0], device=dev)
    sw, sd, scratch = (torch.zeros(8, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    check(lib.cham_h2_scale_rownorm(ptr(Ws1), C, 128, 128, None, ptr(sw), st), "rn")
    check(lib.cham_h2_scale_rownorm2(ptr(dS1), R, 128, 128, sw.data_ptr() + 8, ptr(scratch), ptr(sd), st), "rn2")
    torch.cuda.synchronize()
    fwd3 = lambda: check(lib.cham_gemm_f32x3(ptr(Z2), C, 0, ptr(Ws1), 128, 0, ptr(S1), 128, R, 128, C, ptr(bs1), 1, None, 0, 0, ptr(pred), C, NC, 0, None, 0, 1, st), "x3")
    fwd2 = lambda: check(lib.cham_gemm_f32x2h(ptr(Z2), C, 0, ptr(Ws1), 128, 0, ptr(S1), 128, R, 128, C, ptr(bs1), 1, ptr(pred), C, NC, 0, None, 0, 1, ptr(unit), ptr(sw), st), "x2h")
    wg3 = lambda: check(lib.cham_gemm_f32x3(ptr(Z2), C, 1, ptr(dS1), 128, 0, ptr(gW), 128, C, 128, R, None, 0, None, 0, 0, ptr(pred), C, NC, 0, ptr(ws), ws.numel() * 4, 0, st), "x3")
    wg2 = lambda: check(lib.cham_gemm_f32x2h(ptr(Z2), C, 1, ptr(dS1), 128, 0, ptr(gW), 128, C, 128, R, None, 0, ptr(pred), C, NC, 0, ptr(ws), ws.numel() * 4, 0, ptr(unit), ptr(sd), st), "x2h")
    fwdn = lambda: check(lib.cham_gemm_f32(ptr(Z2), C, 0, ptr(Ws1), 128, 0, ptr(S1), 128, R, 128, C, ptr(bs1), 1, None, 0, 0, ptr(pred), C, NC, 0, None, 0, 1, st), "f32")
    wgn = lambda: check(lib.cham_gemm_f32(ptr(Z2), C, 1, ptr(dS1), 128, 0, ptr(gW), 128, C, 128, R, None, 0, None, 0, 0, ptr(pred), C, NC, 0, ptr(ws), ws.numel() * 4, 0, st), "f32")
    fl = 2.0 * R * C * 128
    idx = torch.arange(0, R, 97, device=dev)
    ref_f = torch.nn.functional.leaky_relu((Z2[idx].double() * pred[idx // NC].double()) @ Ws1.double() + bs1.double(), 0.2)
    ref_w = (Z2.double() * pred.double().repeat_interleave(NC, 0)).t() @ dS1.double()
    for name, f3, f2, fnat, out, ref in (("scorer layer 1 forward", fwd3, fwd2, fwdn, lambda: S1[idx], ref_f), ("Ws1 weight gradient", wg3, wg2, wgn, lambda: gW, ref_w)):
        for tag, fn, peak in (("six bf16 products", f3, 2500 / 6), ("three fp16 products", f2, 2500 / 3), ("native fp32 MFMA", fnat, 157.3)):
            fn(); torch.cuda.synchronize()
            err = float((out().double() - ref).abs().max() / ref.abs().max())
            rms = float((out().double() - ref).pow(2).mean().sqrt() / ref.abs().max())
            bias = float((out().double() - ref).mean() / ref.abs().max())
            ms = timed(fn)
            print("%-24s %-20s %.3f ms  %6.1f TFLOP/s (%.3f of %.0f)  %.2f TB/s of operand reads  err max %.2e rms %.2e mean %.1e" % (
                name, tag, ms, fl / ms / 1e9, fl / ms / 1e9 / peak, peak, (R * C * 4 + R * 128 * 4) / ms / 1e9, err, rms, bias), flush=True)
    for v in (0, 2):
        lib.cham_gemm_f32x3_set_variant(v)
        print("variant %d: fwd x2h %.3f ms, wgrad x2h %.3f ms, fwd x3 %.3f, wgrad x3 %.3f" % (v, timed(fwd2), timed(wg2), timed(fwd3), timed(wg3)), flush=True)
    lib.cham_gemm_f32x3_set_variant(-1)


if __name__ == "__main__":
    main()
