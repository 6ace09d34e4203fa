"""Stand-alone timing of the three candidate-row CAR GEMMs (csrc/gemm_h2.hip) on ROW-MAJOR against TILE-BLOCKED planes (round 6), and of the two
plane producers in both layouts.  python scripts/bench_h2_blocked.py [rows]   (the block padding is a build option: CHAM_BUILD_DEFINES=-DH2B_PAD=n)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chameleon_recsys_amd import _lib
from chameleon_recsys_amd._lib import check, ptr
from chameleon_recsys_amd.nar.nar_model import blocked_plane_elements


def timed(fn, n=8):
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
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 248064
    C, BT, N = 1024, (R // 51), 50
    R = BT * 51
    st = torch.cuda.current_stream().cuda_stream
    tiles = -(-R // 256) + 1
    bps = blocked_plane_elements(tiles, C)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd16 = lambda *s: (torch.randn(*s, device=dev, generator=g) * 100).half()
    # planes with plausible contents (timing only): same values in both layouts is not needed
    Zr, Dr = rnd16(2, R, C), rnd16(2, R, C)
    Zb, Db = rnd16(2, bps), rnd16(2, bps)
    W = rnd16(2, C, C)
    rec = torch.zeros(8, device=dev); rec[0] = 1.0; rec[1] = 1.0
    bias = torch.randn(C, device=dev, generator=g)
    Out = torch.empty(R, C, device=dev); Wg = torch.empty(C, C, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)

    def gemm(A, a_ps, a_t, B, b_ps, b_t, tn, Cc, M, Nn, K, bias_=None, act=0, dref=None, dblk=0, splits=1):
        return lambda: check(lib.cham_gemm_h2b(ptr(A), a_ps, C, ptr(rec), ptr(B), b_ps, C, ptr(rec), tn, ptr(Cc), Nn, M, Nn, K, ptr(bias_), act, ptr(dref), C,
                                               1 if dref is not None else 0, 0, ptr(ws) if splits != 1 else None, ws.numel() * 4 if splits != 1 else 0, splits,
                                               a_t, b_t, dblk, st), "h2b")
    flops = 2.0 * R * C * C
    print("block elements %d (pad %d), rows %d" % (lib.cham_h2b_block_elements(), lib.cham_h2b_block_elements() - 8192, R))
    for name, rm, bl in (
        ("CAR forward (NT, tanh)", gemm(Zr, R * C, 0, W, C * C, 0, 0, Out, R, C, C, bias_=bias, act=2), gemm(Zb, bps, tiles, W, C * C, 0, 0, Out, R, C, C, bias_=bias, act=2)),
        ("CAR dgrad (NT, leaky')", gemm(Dr, R * C, 0, W, C * C, 0, 0, Out, R, C, C, dref=Zr[0]), gemm(Db, bps, tiles, W, C * C, 0, 0, Out, R, C, C, dref=Zb[0], dblk=1)),
        ("W2 wgrad (TN, split-K)", gemm(Zr, R * C, 0, Dr, R * C, 0, 1, Wg, C, C, R, splits=32), gemm(Zb, bps, tiles, Db, bps, tiles, 1, Wg, C, C, R, splits=32)),
    ):
        a, b = timed(rm), timed(bl)
        a2, b2 = timed(rm), timed(bl)
        print("%-26s row-major %.3f / %.3f ms (%.3f of 833)   blocked %.3f / %.3f ms (%.3f)" % (name, a, a2, flops / min(a, a2) / 1e9 / 833.3, b, b2, flops / min(b, b2) / 1e9 / 833.3), flush=True)
    # producers
    U = torch.randn(BT, C, device=dev, generator=g); V = torch.randn(2 * BT + 1001, C, device=dev, generator=g)
    neg_slot = torch.randint(0, 1001, (BT, N), device=dev, generator=g, dtype=torch.int32)
    check(lib.cham_h2_scale_absmax(ptr(U), U.numel(), ptr(V), V.numel(), ptr(rec), st), "absmax")
    for blk, Z, ps in ((0, Zr, R * C), (1, Zb, bps)):
        ms = timed(lambda: check(lib.cham_combine_fwd_h2b(ptr(U), ptr(V), C, BT, N, 1000, ptr(neg_slot), ptr(Z), ps, ptr(rec), blk, st), "combine"))
        print("cham_combine_fwd_h2b blocked=%d: %.3f ms (%.2f TB/s of plane writes)" % (blk, ms, R * C * 4 / ms / 1e9), flush=True)
    dS1 = torch.randn(R, 128, device=dev, generator=g) * 1e-3
    Ws1 = torch.randn(C, 128, device=dev, generator=g) * 0.05
    Z2c = torch.tanh(torch.randn(R, C, device=dev, generator=g)); pred = torch.tanh(torch.randn(BT, C, device=dev, generator=g))
    sw, so, sd = (torch.zeros(8, device=dev) for _ in range(3))
    check(lib.cham_h2_scale_rownorm(ptr(Ws1), C, 128, 128, None, ptr(sw), st), "rn")
    check(lib.cham_h2_scale_rownorm2(ptr(dS1), R, 128, 128, sw.data_ptr() + 8, ptr(so), ptr(sd), st), "rn2")
    Wh = torch.zeros(2, C, 128, dtype=torch.float16, device=dev)
    check(lib.cham_split2h(ptr(Ws1), C, 128, 128, ptr(Wh), C * 128, 128, None, 0, 0, ptr(sw), 0, st), "split")
    dpred, b2 = torch.zeros(BT, C, device=dev), torch.zeros(BT, C, device=dev)
    f_rm = lambda: check(lib.cham_dm_mulpred_h2h(ptr(dS1), 128, 128, ptr(Wh), C * 128, ptr(sd), ptr(sw), ptr(Z2c), ptr(pred), C, BT, N, ptr(Dr), R * C, ptr(so), ptr(dpred), ptr(b2), st), "rm")
    f_bl = lambda: check(lib.cham_dm_mulpred_h2_blk(ptr(dS1), 128, 128, ptr(Wh), C * 128, ptr(sd), ptr(sw), ptr(Z2c), ptr(pred), C, BT, N, ptr(Db), bps, ptr(so), ptr(dpred), ptr(b2), st), "bl")
    print("fused scorer dgrad: row-major %.3f ms, blocked %.3f ms" % (timed(f_rm), timed(f_bl)), flush=True)


if __name__ == "__main__":
    main()
