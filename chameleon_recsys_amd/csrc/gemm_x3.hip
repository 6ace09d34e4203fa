// fp32 GEMM on the bf16 matrix cores: every fp32 operand is split into three bf16 planes and six plane products are accumulated
// in fp32 ("bf16x3"; same contract, arguments and epilogues as cham_gemm_f32 in gemm.hip).
//
//   a = a_h + a_m + a_l   (a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m); 3 x 8 significand bits = fp32's 24)
//   a b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m)  [+ a_m b_l + a_l b_m + a_l b_l: two terms of <= 2^-24 |a b| each (worst case 2^-23, tests/test_split3_cpu.py), dropped]
//
// bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result carries the same kind of
// error as the native fp32 MFMA path (fp32 accumulation rounding; tests/test_gemm_x3_gpu.py measures both against float64 on the
// same operands).  Why: gfx950's fp32 matrix rate is 1/16 of its bf16 rate (157 vs 2 500 TFLOP/s dense), so six bf16 products cost
// 6/16 of one fp32 product - an fp32-accurate GEMM with a 417 TFLOP/s ceiling instead of 157.  Per 16 k of a 256x128 tile every wave
// issues 24 MFMAs (768 matrix-pipe cycles) and ~117 other instructions (3 global float4 loads, the split of 12 elements, 12
// ds_read_b128, 5 LDS writes); measured 184-194 TFLOP/s stand-alone at the CAR shapes = 0.44-0.47 of the ceiling, 1.5x the native
// fp32 MFMA kernels (profiles/r02_notes.md item 4 for what bounds it).
//
// Storage, windows, tile swizzle, split-K placement and the epilogue are those of gemm.hip (gemm_shared.h).  LDS image per plane:
// [row][k] bf16, row stride BK + 8 = 24 elements (48 B = 3 x 16-B slots, coprime with the 16 slots of a bank row), planes and the
// two pipeline buffers behind each other: 2 x 3 x (BM + BN) x 48 B = 110.6 KB for 256x128, 73.7 KB for 128x128.
// Inf / NaN: an infinite operand gives NaN (inf - inf in the split) where fp32 would give inf; operands below 2^-100 lose their lowest
// plane to bf16's subnormal range (absolute error <= 2^-133 per element); everything else is exact (tests/test_split3_cpu.py).
// Ablation bits for probe builds (scripts/ab_x3.sh; 0 in the product library; results are wrong with most of them, timing only):
//   1 no split arithmetic (raw bits stored) | 2 no global loads in the loop | 4 no fragment ds_reads | 8 one MFMA pass instead of six
//   16 no split / LDS writes | 64 accumulators pinned to AccVGPRs (inline asm) | 128 role split: waves 0-3 only MFMAs, waves 4-7 only
//   the other streams | 512 no s_barrier in the loop | 1024 no lgkmcnt(0) before it
#ifndef X3_ABL
#define X3_ABL 0
#endif
#include "gemm_shared.h"
#include <stdlib.h>


typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 x2h_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x3_split(float a, __bf16& h, __bf16& m, __bf16& l) {
    if (X3_ABL & 1) { const unsigned u = __float_as_uint(a); h = __builtin_bit_cast(__bf16, (unsigned short)(u >> 16)); m = h; l = h; }
    else split3(a, h, m, l);
}

// One fp32 operand tile [BF x BK] -> three bf16 planes in LDS.
//   XK  (reduction index contiguous in memory: A of NN / NT, B of NT): unit = 8 consecutive k of one row (2 float4); LDS image
//       [row][k], row stride BK + 8; one ds_write_b128 per plane; fragments are plain ds_read_b128.
//   !XK (free index contiguous: B of NN, A and B of TN): unit = one float4 = 4 consecutive rows of one k; LDS image [k][row], row
//       stride BF + 8; one ds_write_b64 per plane straight through - the transposition happens in the fragment read
//       (ds_read_b64_tr_b16, as in gemm_b16.hip), not in registers.
//   NP = 3: three bf16 planes (x3_split) | NP = 2: TWO fp16 planes of x * scale (split2h of common.h; plane 0 = h, plane 1 = l)
template <int BF, int BK, bool XK, int NTH, bool RS, int NP = 3>
struct StageX3 {
    static constexpr int LD = XK ? BK + 8 : BF + 8;
    static constexpr int PLANE = XK ? BF * LD : BK * LD;                    // elements of one plane of one buffer
    static constexpr int UK = (XK && BF * BK / 8 >= NTH) ? 8 : 4;            // XK: k per unit (4 when 8 would leave threads without a unit:
                                                                             // a divergent skip would split the loop body into basic blocks)
    static constexpr int NU = XK ? BF * BK / UK : BK * BF / 4;              // units in the tile
    static constexpr int NV = (NU + NTH - 1) / NTH;                          // units per thread
    static constexpr int NL = (XK && UK == 8) ? 2 : 1;                       // float4 loads per unit
    struct Regs {            // one tile in flight (two sets alternate: one is converted while the other is being loaded)
        u32x4 r[NV][NL];
        u32x4 sc[RS ? NV : 1][RS ? NL : 1];
    };
    unsigned off[NV];        // window-local byte offset of the unit (OOB_OFF: free index out of range / no unit)
    unsigned soff[NV];
    int k0u[NV];             // tile-local k of the unit's first element
    float sc;                // NP == 2: the operand's power-of-two scale (H2Scale record, wave-uniform)
    // one element -> its planes (x = h, z = l, y = the middle bf16 plane of NP == 3)
    __device__ __forceinline__ void split(float a, __bf16& x, __bf16& y, __bf16& z) const {
        if constexpr (NP == 2) {
            _Float16 h, l;
            split2h(a * sc, h, l);
            x = __builtin_bit_cast(__bf16, h); z = __builtin_bit_cast(__bf16, l); y = z;
        } else {
            x3_split(a, x, y, z);
        }
    }

    __device__ __forceinline__ void init(int ld, int limF, int f0, int ldrs, int rs_div, bool has_rs, float scale = 1.f) {
        const int tid = threadIdx.x;
        sc = scale;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * NTH;
            unsigned o; bool fok;
            if (XK) {
                const int fr = u / (BK / UK), kc = u % (BK / UK);
                o = ((unsigned)fr * (unsigned)ld + (unsigned)(kc * UK)) * 4u; k0u[i] = kc * UK; fok = fr < limF;
                soff[i] = has_rs ? ((unsigned)((f0 + fr) / rs_div) * (unsigned)ldrs + (unsigned)(kc * UK)) * 4u : 0u;
            } else {
                const int k = u / (BF / 4), f4 = u % (BF / 4);
                o = ((unsigned)k * (unsigned)ld + (unsigned)f4 * 4u) * 4u; k0u[i] = k; fok = f4 * 4 < limF;
                soff[i] = 0u;
            }
            if (NU % NTH != 0 && u >= NU) fok = false;
            off[i] = fok ? o : OOB_OFF;
        }
    }
    // limK <= 0 (a tile beyond the reduction range): every offset is out of window, zeros are staged
    __device__ __forceinline__ void load(Regs& g, __amdgpu_buffer_rsrc_t win, int limK) const {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const unsigned o = (off[i] != OOB_OFF && k0u[i] + 4 * j < limK) ? off[i] + 16u * j : OOB_OFF;
                g.r[i][j] = __builtin_amdgcn_raw_buffer_load_b128(win, o, 0, 0);
            }
    }
    __device__ __forceinline__ void load_one(Regs& g, __amdgpu_buffer_rsrc_t win, int limK, int idx) const {
        const int i = idx / NL, j = idx % NL;
        const unsigned o = (off[i] != OOB_OFF && k0u[i] + 4 * j < limK) ? off[i] + 16u * j : OOB_OFF;
        g.r[i][j] = __builtin_amdgcn_raw_buffer_load_b128(win, o, 0, 0);
    }
    // row-broadcast scale (A only): XK -> the scale row of the unit's row, same k; !XK -> the scale row of stored row k / rs_div
    __device__ __forceinline__ void load_scale(Regs& g, __amdgpu_buffer_rsrc_t rsw, int kglob0, int f0, int ldrs, int rs_div, int limK) const {
        if constexpr (RS) {
            const int tid = threadIdx.x;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const bool ok = off[i] != OOB_OFF && k0u[i] + 4 * j < limK;
                    if (XK) {
                        g.sc[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? soff[i] + 16u * j : OOB_OFF, kglob0 * 4, 0);
                    } else {
                        const int f4 = (tid + i * NTH) % (BF / 4);
                        const unsigned o = ((unsigned)((kglob0 + k0u[i]) / rs_div) * (unsigned)ldrs + (unsigned)(f0 + f4 * 4)) * 4u;
                        g.sc[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? o : OOB_OFF, 0, 0);
                    }
                }
        }
    }
    __device__ __forceinline__ void store3(Regs& g, __bf16* __restrict__ S) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * NTH;
            if (NU % NTH != 0 && u >= NU) continue;
            if constexpr (RS) {
#pragma unroll
                for (int j = 0; j < NL; ++j) mul4(g.r[i][j], g.sc[i][j]);
            }
            if (XK && UK == 8) {
                const int fr = u / (BK / 8), kc = u % (BK / 8);
                bf16x8 h, m, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) { __bf16 x, y, z; split(__uint_as_float(comp(g.r[i][j >> 2], j & 3)), x, y, z); h[j] = x; m[j] = y; l[j] = z; }
                __bf16* d = S + fr * LD + kc * 8;
                *reinterpret_cast<bf16x8*>(d) = h;
                if constexpr (NP == 3) *reinterpret_cast<bf16x8*>(d + PLANE) = m;
                *reinterpret_cast<bf16x8*>(d + (NP - 1) * PLANE) = l;
            } else if (XK) {
                const int fr = u / (BK / 4), kc = u % (BK / 4);
                bf16x4 h, m, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { __bf16 x, y, z; split(__uint_as_float(comp(g.r[i][0], j)), x, y, z); h[j] = x; m[j] = y; l[j] = z; }
                __bf16* d = S + fr * LD + kc * 4;
                *reinterpret_cast<bf16x4*>(d) = h;
                if constexpr (NP == 3) *reinterpret_cast<bf16x4*>(d + PLANE) = m;
                *reinterpret_cast<bf16x4*>(d + (NP - 1) * PLANE) = l;
            } else {
                const int k = u / (BF / 4), f4 = u % (BF / 4);
                bf16x4 h, m, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { __bf16 x, y, z; split(__uint_as_float(comp(g.r[i][0], j)), x, y, z); h[j] = x; m[j] = y; l[j] = z; }
                __bf16* d = S + k * LD + f4 * 4;
                *reinterpret_cast<bf16x4*>(d) = h;
                if constexpr (NP == 3) *reinterpret_cast<bf16x4*>(d + PLANE) = m;
                *reinterpret_cast<bf16x4*>(d + (NP - 1) * PLANE) = l;
            }
        }
    }
    // the same split, one element at a time (the K loop places one element's ~6 VALU instructions under each MFMA)
    static constexpr int UE = (XK && UK == 8) ? 8 : 4;                       // elements per unit
    static constexpr int NE = NV * UE;                                       // elements per thread and tile
    typedef __bf16 vec_t __attribute__((ext_vector_type(UE)));
    struct Conv { vec_t h[NV], m[NV], l[NV]; };
    __device__ __forceinline__ void convert_one(const Regs& g, Conv& c, int e) const {
        const int i = e / UE, k = e % UE;
        float a = __uint_as_float(comp(g.r[i][k >> 2], k & 3));
        if constexpr (RS) a *= __uint_as_float(comp(g.sc[i][k >> 2], k & 3));
        __bf16 x, y, z;
        split(a, x, y, z);
        c.h[i][k] = x; c.m[i][k] = y; c.l[i][k] = z;
    }
    __device__ __forceinline__ void write(const Conv& c, __bf16* __restrict__ S) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * NTH;
            if (NU % NTH != 0 && u >= NU) continue;
            __bf16* d;
            if (XK) { const int fr = u / (BK / UK), kc = u % (BK / UK); d = S + fr * LD + kc * UK; }
            else { const int k = u / (BF / 4), f4 = u % (BF / 4); d = S + k * LD + f4 * 4; }
            *reinterpret_cast<vec_t*>(d) = c.h[i];
            if constexpr (NP == 3) *reinterpret_cast<vec_t*>(d + PLANE) = c.m[i];
            *reinterpret_cast<vec_t*>(d + (NP - 1) * PLANE) = c.l[i];
        }
    }
};

// fragment = 8 consecutive k (8 * (lane >> 5) ...) of row f0 + (lane & 31) of one plane
template <bool XK, int LD>
__device__ __forceinline__ bf16x8 x3_frag(const __bf16* __restrict__ S, int f0, int lane) {
    if (XK) {
        return *reinterpret_cast<const bf16x8*>(S + (f0 + (lane & 31)) * LD + 8 * (lane >> 5));
    } else {
        // [k][row] image: each 16-lane group transposes a 4(k) x 16(row) block (gemm_b16.hip frag_read)
        const int i = lane & 15;
        const __bf16* q = S + (8 * (lane >> 5) + (i >> 2)) * LD + f0 + 16 * ((lane >> 4) & 1) + (i & 3) * 4;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(q + 4 * LD));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8, v);
    }
}

// Software pipeline, one __syncthreads per 16-k tile, four independent streams inside a phase:
//   MFMA   on the register fragments of tile t        (read from LDS during phase t-1)
//   ds_read of the fragments of tile t+1              (LDS buffer (t+1) & 1, written during phase t-1)
//   split + ds_write of tile t+2                      (global registers loaded during phase t-1 -> LDS buffer t & 1, whose reads
//                                                      finished before the barrier that ended phase t-1)
//   buffer_load of tile t+3                           (into the register set converted during phase t+1)
// so nothing after the barrier waits for LDS or HBM before the matrix pipe has work.  The loop is unrolled by two (fragment /
// register sets alternate by name, no copies); a tile beyond the reduction range stages zeros.
//
// NP = 2 (round 5, cham_gemm_f32x2h): the same kernel over TWO fp16 planes of (operand x its power-of-two scale, H2Scale records p.sa / p.sb as
// in gemm_h2.hip) split while staged, THREE products v_mfma_f32_32x32x16_f16 (a_l b_h, a_h b_l, a_h b_h) instead of six, the accumulators
// scaled back by 1 / (s_a s_b) in front of the shared epilogue.  Half the matrix-pipe work per fp32 product; the conversions of a tile go
// two to an MFMA slice.  Used where the operand is produced in fp32 anyway and a bound of its magnitude is known on the device (the scorer's
// first layer over cand (.) pred: |tanh x tanh| <= 1, and its weight gradient).
template <int BM, int BN, int WM, int WN, bool AK, bool BKC, int EPI, bool RS, int NP>
__global__ __launch_bounds__(WM * WN * 64) void gemm_x3_kernel(GemmParams p) {
    constexpr int BK = 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NTH = WM * WN * 64;
    using LA = StageX3<BM, BK, AK, NTH, RS, NP>;
    using LB = StageX3<BN, BK, BKC, NTH, false, NP>;
    constexpr int APL = LA::PLANE, BPL = LB::PLANE;
    constexpr int ASZ = NP * APL, BSZ = NP * BPL;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);      // [2][3][plane]
    __bf16* Bs = As + 2 * ASZ;                         // [2][3][plane]

    const int nwg = p.nbm * p.nbn;
    int tile_m, tile_n, split;
    if (p.xcd_split) {                                 // one K-split per XCD (see gemm.hip)
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
        split = (lin & 7) + 8 * (slot / nwg);
        const int t = slot % nwg;
        tile_m = t / p.nbn; tile_n = t % p.nbn;
    } else {
        const int id = blockIdx.x;
        const int q = nwg / 8, rr = nwg % 8, xcd = id % 8;
        const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + id / 8;
        tile_m = swz / p.nbn; tile_n = swz % p.nbn;
        split = blockIdx.y;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const char* abase = reinterpret_cast<const char*>(p.A) + (AK ? ((size_t)m0 * p.lda + kbeg) : ((size_t)kbeg * p.lda + m0)) * 4;
    const char* bbase = reinterpret_cast<const char*>(p.B) + (BKC ? ((size_t)n0 * p.ldb + kbeg) : ((size_t)kbeg * p.ldb + n0)) * 4;
    const size_t astep = (AK ? (size_t)BK : (size_t)BK * p.lda) * 4, bstep = (BKC ? (size_t)BK : (size_t)BK * p.ldb) * 4;
    const __amdgpu_buffer_rsrc_t rsw = make_window(p.rs);

    LA la; LB lb;
    float sa_inv = 1.f, sb_inv = 1.f, sa = 1.f, sb = 1.f;      // (the inverse scales are applied one after the other: their product can be denormal - ADVICE r05)
    if constexpr (NP == 2) { sa = p.sa[0]; sb = p.sb[0]; sa_inv = p.sa[1]; sb_inv = p.sb[1]; }
    la.init(p.lda, p.M - m0, m0, p.ldrs, p.rs_div, RS, sa);
    lb.init(p.ldb, p.N - n0, n0, 0, 1, false, sb);
    const int nk = (kend - kbeg + BK - 1) / BK;

    struct Frags { bf16x8 a[NP][TM], b[NP][TN]; };
    Frags F0, F1;
    typename LA::Regs RA[4];
    typename LB::Regs RB[4];

    auto gload = [&](typename LA::Regs& ra, typename LB::Regs& rb, int t) {          // global -> registers, tile t
        const int k0 = kbeg + t * BK;
        la.load(ra, make_window(abase + (size_t)t * astep), kend - k0);
        lb.load(rb, make_window(bbase + (size_t)t * bstep), kend - k0);
        la.load_scale(ra, rsw, k0, m0, p.ldrs, p.rs_div, kend - k0);
    };
    auto fread = [&](Frags& f, int buf) {
        const __bf16* Ac = As + buf * ASZ;
        const __bf16* Bc = Bs + buf * BSZ;
        // in the order the MFMA passes consume them: (a_l, b_h), (a_h, b_l), (a_m, b_m)
        constexpr int QA[3] = {NP - 1, 0, 1}, QB[3] = {0, NP - 1, 1};
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[QA[q]][i] = x3_frag<AK, LA::LD>(Ac + QA[q] * APL, wm0 + i * 32, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) f.b[QB[q]][j] = x3_frag<BKC, LB::LD>(Bc + QB[q] * BPL, wn0 + j * 32, lane);
        }
    };
    // smallest terms first; every pass walks all TM x TN accumulators, so dependent MFMAs are TM * TN instructions apart
    constexpr int PA[6] = {NP - 1, 0, NP == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NP - 1, NP == 3 ? 1 : 0, 0, 1, 0};          // NP == 2: the first three
    // Phase t: MFMAs on tile t; fragments of tile t + 1 read; tile t + 2 split (register set `c`) and written to LDS; tile t + 5
    // requested from memory into register set `l` (the set phase t - 1 finished splitting).  The operands stream from HBM - the
    // workgroups that share an A panel walk K in step, so every request sees a miss, ~2 us under load, longer than a phase: one
    // phase of distance left the waves parked at vmcnt for a third of their cycles (SQ_WAIT_ANY, profiles/r02_notes.md).
    constexpr bool BRING = !BKC && !AK;          // B streams from HBM too (wgrad: both operands run along the rows); a weight matrix is L2 resident
    auto phase = [&](const Frags& fc, Frags& fn, typename LA::Regs& rac, typename LB::Regs& rbc, typename LA::Regs& ral,
                     typename LB::Regs& rbl, int t) {
        const int cur = t & 1;
        typename LA::Conv ca;
        typename LB::Conv cb;
        constexpr int NEA = LA::NE, NEB = LB::NE, NMF = NPROD * TM * TN, NFR = NP * (TM + TN);
        constexpr int NGA = LA::NV * LA::NL, NGB = LB::NV * LB::NL;
        // slices: one MFMA each, plus - fragment reads in the first NFR slices (they must have landed by the barrier), the global
        // requests in the first NGA + NGB, B's split / LDS write, then A's (CPS conversions per slice: one where they fit, two under
        // the three-product form's twelve MFMAs)
        constexpr int CPS = (NEB + 1 + NEA < NMF) ? 1 : 2;
        constexpr int SB_W = (NEB + CPS - 1) / CPS, SA_0 = SB_W + 1, SA_W = SA_0 + (NEA + CPS - 1) / CPS;
        static_assert(SA_W < NMF && NFR <= NMF && NGA + NGB + 1 <= NMF, "the streams must fit under the MFMAs of one phase");
        const int kn = kbeg + (t + 5) * BK, knb = kbeg + (t + (BRING ? 5 : 3)) * BK;
        const __amdgpu_buffer_rsrc_t awn = make_window(abase + (size_t)(t + 5) * astep);
        const __amdgpu_buffer_rsrc_t bwn = make_window(bbase + (size_t)(t + (BRING ? 5 : 3)) * bstep);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NPROD; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int m = (s * TM + i) * TN + j;
                    if ((!(X3_ABL & 8) || s == 0) && (!(X3_ABL & 128) || wave < WM * WN / 2)) {
                        if constexpr (NP == 2)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x2h_half8, fc.a[PA[s]][i]),
                                                                               __builtin_bit_cast(x2h_half8, fc.b[PB[s]][j]), acc[i][j], 0, 0, 0);
                        else if (X3_ABL & 64)       // probe: accumulators pinned to the AccVGPR file
                            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fc.a[PA[s]][i]), "v"(fc.b[PB[s]][j]));
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.a[PA[s]][i], fc.b[PB[s]][j], acc[i][j], 0, 0, 0);
                    }
                    if (!(X3_ABL & 2) && (!(X3_ABL & 128) || wave >= WM * WN / 2)) {     // one global float4 of tile t + 5 per slice
                        if (m < NGA) la.load_one(ral, awn, kend - kn, m);
                        if (BRING) { if (m >= NGA && m < NGA + NGB) lb.load_one(rbl, bwn, kend - knb, m - NGA); }
                        else { if (m > SB_W && m <= SB_W + NGB) lb.load_one(rbc, bwn, kend - knb, m - SB_W - 1); }          // reload the set just split
                        if constexpr (RS) { if (m == NGA + NGB) la.load_scale(ral, rsw, kn, m0, p.ldrs, p.rs_div, kend - kn); }
                    }
                    if (!(X3_ABL & 4) && (!(X3_ABL & 128) || wave >= WM * WN / 2) && m < NFR) {          // fragment m of tile t + 1
                        const int q = m / (TM + TN), r = m % (TM + TN);
                        constexpr int QA[3] = {NP - 1, 0, 1}, QB[3] = {0, NP - 1, 1};
                        if (r < TM) fn.a[QA[q]][r] = x3_frag<AK, LA::LD>(As + (cur ^ 1) * ASZ + QA[q] * APL, wm0 + r * 32, lane);
                        else fn.b[QB[q]][r - TM] = x3_frag<BKC, LB::LD>(Bs + (cur ^ 1) * BSZ + QB[q] * BPL, wn0 + (r - TM) * 32, lane);
                    }
                    if (!(X3_ABL & 16) && (!(X3_ABL & 128) || wave >= WM * WN / 2)) {
#pragma unroll
                        for (int c = 0; c < CPS; ++c) {
                            const int eb = m * CPS + c, ea = (m - SA_0) * CPS + c;
                            if (m < SB_W) { if (eb < NEB) lb.convert_one(rbc, cb, eb); }
                            else if (m >= SA_0 && m < SA_W) { if (ea < NEA) la.convert_one(rac, ca, ea); }
                        }
                        if (m == SB_W) lb.write(cb, Bs + cur * BSZ);
                        if (m == SA_W) la.write(ca, As + cur * ASZ);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        // LDS-only barrier: __syncthreads() may also drain vmcnt, i.e. wait for the global requests in flight - the very latency the
        // register staging exists to hide.  This wave's ds_writes / ds_reads are complete at lgkmcnt(0).
        // (builtins, not inline asm: the compiler's wait-count pass must SEE the drain, or it keeps believing the fragment reads are
        // pending and puts lgkmcnt(n) waits in front of next phase's MFMAs - which then also wait for that phase's young requests)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (!(X3_ABL & 1024)) __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0), vmcnt / expcnt untouched
        if (!(X3_ABL & 512)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {
        gload(RA[0], RB[0], 0);
        gload(RA[1], RB[1], 1);
        la.store3(RA[0], As); lb.store3(RB[0], Bs);
        la.store3(RA[1], As + ASZ); lb.store3(RB[1], Bs + BSZ);
        if (BRING) {
            gload(RA[2], RB[2], 2);
            gload(RA[3], RB[3], 3);
            gload(RA[0], RB[0], 4);
        } else {          // B: one register set, tile 2 now, tile t + 3 reloaded in phase t
            gload(RA[2], RB[0], 2);
            la.load(RA[3], make_window(abase + (size_t)3 * astep), kend - (kbeg + 3 * BK));
            la.load_scale(RA[3], rsw, kbeg + 3 * BK, m0, p.ldrs, p.rs_div, kend - (kbeg + 3 * BK));
            la.load(RA[0], make_window(abase + (size_t)4 * astep), kend - (kbeg + 4 * BK));
            la.load_scale(RA[0], rsw, kbeg + 4 * BK, m0, p.ldrs, p.rs_div, kend - (kbeg + 4 * BK));
        }
        __syncthreads();
        fread(F0, 0);
        __syncthreads();          // every wave holds tile 0 in registers before phase 0 overwrites LDS buffer 0
        for (int kt = 0; kt < nk; kt += 4) {          // tile i lives in register set i % 4
            // (no early exit: up to three trailing phases multiply staged zeros - gemm_plan makes the K chunk a multiple of 64, so only
            // a ragged last chunk or a K that is not a multiple of 64 pays for them; exits in the middle of the unrolled body made the
            // compiler keep a second copy of the accumulators)
            phase(F0, F1, RA[2], RB[BRING ? 2 : 0], RA[1], RB[BRING ? 1 : 0], kt);
            phase(F1, F0, RA[3], RB[BRING ? 3 : 0], RA[2], RB[BRING ? 2 : 0], kt + 1);
            phase(F0, F1, RA[0], RB[0], RA[3], RB[BRING ? 3 : 0], kt + 2);
            phase(F1, F0, RA[1], RB[BRING ? 1 : 0], RA[0], RB[0], kt + 3);
        }
    }
    // (the lane id goes through an opaque move: otherwise the epilogue's per-lane offsets are computed above the K loop and held in
    // VGPRs through it - the loop has none to spare)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int kl = lane_e >> 5, fl = lane_e & 31;
    if constexpr (NP == 2) {          // back to true units (a power of two: exact)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] * sa_inv) * sb_inv;
    }
    gemm_epilogue<EPI, TM, TN>(p, acc, m0, n0, wm0, wn0, split, kl, fl);
}

// launch counters: [0] 128x128, [1] 256x128, [3] delegated to the native fp32 kernels (N <= 64), [4] / [5] the two tiles of the
// two-fp16-plane form (cham_gemm_f32x2h), [6] EPI and [7] K-splits of the last launch
static long long g_x3_launches[8];
extern "C" void cham_gemm_f32x3_launch_counts(long long* out8, int reset) {
    for (int i = 0; i < 8; ++i) { if (out8) out8[i] = g_x3_launches[i]; if (reset) g_x3_launches[i] = 0; }
}

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, int EPI, int NP>
static int x3_launch_epi(GemmParams& p, hipStream_t st) {
    g_x3_launches[6] = EPI; g_x3_launches[7] = p.splits;
    constexpr size_t smem = (size_t)2 * NP * (StageX3<BM, 16, AK, WM * WN * 64, false>::PLANE + StageX3<BN, 16, BKC, WM * WN * 64, false>::PLANE) * 2;
    constexpr bool RSI = (EPI == 1 && AK && !BKC) || ((EPI == 6 || EPI == 0) && !AK && !BKC);      // as in gemm.hip
    if (p.rs != nullptr && !RSI) return -CHAM_ERR_ARG;
    const dim3 grid(p.nbm * p.nbn, p.splits, 1), block(WM * WN * 64);
    if (RSI && p.rs != nullptr) {
        auto k = gemm_x3_kernel<BM, BN, WM, WN, AK, BKC, EPI, RSI, NP>;
        CHAM_SET_DYNAMIC_LDS(k, (int)smem);
        hipLaunchKernelGGL(k, grid, block, smem, st, p);
        CHAM_CHECK_LAUNCH();
        return CHAM_OK;
    }
    auto k = gemm_x3_kernel<BM, BN, WM, WN, AK, BKC, EPI, false, NP>;
    CHAM_SET_DYNAMIC_LDS(k, (int)smem);
    hipLaunchKernelGGL(k, grid, block, smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, int NP>
static int x3_launch_cfg(GemmParams& p, hipStream_t st) {
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.N + BN - 1) / BN;
    // epilogues per layout as in gemm.hip's launch_cfg: NN plain | bias (+ leaky | tanh); NT plain | x act'(dref); TN plain | split-K partial
    constexpr bool NN = AK && !BKC, NT = AK && BKC, TN = !AK && !BKC;
    if (p.splits > 1) {
        if constexpr (TN) {
            p.xcd_split = (p.splits % 8 == 0) ? 1 : 0;
            const int rc = x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 6, NP>(p, st);
            if (rc != CHAM_OK) return rc;
            launch_splitk_reduce(p, st);
            CHAM_CHECK_LAUNCH();
            return CHAM_OK;
        } else {
            return -CHAM_ERR_ARG;
        }
    }
    if (p.dref) {
        if (p.bias || p.act != ACT_NONE) return -CHAM_ERR_ARG;
        if constexpr (NT) {
            if (p.dact == ACT_LEAKY) return x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 3, NP>(p, st);
            if (p.dact == ACT_TANH) return x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 4, NP>(p, st);
        }
        return -CHAM_ERR_ARG;
    }
    if (p.bias || p.act != ACT_NONE) {
        if (!p.bias || p.accumulate) return -CHAM_ERR_ARG;
        if constexpr (NN) {
            if (p.act == ACT_LEAKY) return x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 1, NP>(p, st);
            if (p.act == ACT_TANH) return x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 2, NP>(p, st);
            return x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 5, NP>(p, st);
        }
        return -CHAM_ERR_ARG;
    }
    return x3_launch_epi<BM, BN, WM, WN, BK, AK, BKC, 0, NP>(p, st);
}

static int g_x3_variant = -1;     // -1 = automatic; 0 = 128x128 / 4 waves, 2 = 256x128 / 8 waves
extern "C" void cham_gemm_f32x3_set_variant(int v) { g_x3_variant = v; }

template <bool AK, bool BKC, int NP>
static int x3_by_shape(GemmParams& p, hipStream_t st) {
    auto grid = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.splits; };
    int v = ((long)p.M * p.N >= (1L << 20) && grid(256, 128) >= 256) ? 2 : 0;
    if (g_x3_variant >= 0) v = g_x3_variant;
    switch (v) {
        case 2: ++g_x3_launches[NP == 2 ? 5 : 1]; return x3_launch_cfg<256, 128, 4, 2, 16, AK, BKC, NP>(p, st);
        default: ++g_x3_launches[NP == 2 ? 4 : 0]; return x3_launch_cfg<128, 128, 2, 2, 16, AK, BKC, NP>(p, st);
    }
}

extern "C" int cham_gemm_f32(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc, int M, int N, int K,
                             const float* bias, int act, const float* dref, int ldr, int dact, const float* rowscale, int ldrs, int rs_div,
                             int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, void* stream);

extern "C" int cham_gemm_f32x3(const float* A, int lda, int transA, const float* B, int ldb, int transB,
                               float* C, int ldc, int M, int N, int K,
                               const float* bias, int act,
                               const float* dref, int ldr, int dact,
                               const float* rowscale, int ldrs, int rs_div,
                               int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                               void* stream) {
    // narrow outputs (scorer layers 2-3 and their twins) are HBM-bound: the native fp32 kernels already stream them; plain TN products with
    // M <= 128 (small-output weight gradients over a long reduction) go to csrc/gemm.hip's VALU kernel through the same entry point
    if (N <= 64 || (transA && !transB && M <= 128 && K >= 512 && !bias && act == ACT_NONE && !dref && !rowscale)) {
        ++g_x3_launches[3];
        return cham_gemm_f32(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, dref, ldr, dact, rowscale, ldrs, rs_div,
                             accumulate, workspace, workspace_bytes, splits_hint, stream);
    }
    GemmParams p;
    const int rc = gemm_plan(p, A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, dref, ldr, dact, rowscale, ldrs, rs_div,
                             accumulate, workspace, workspace_bytes, splits_hint);
    if (rc != CHAM_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!transA && !transB) return x3_by_shape<true, false, 3>(p, st);
    if (!transA && transB) return x3_by_shape<true, true, 3>(p, st);
    if (transA && !transB) return x3_by_shape<false, false, 3>(p, st);
    return -CHAM_ERR_ARG;
}

// The same GEMM over two fp16 planes of each operand, split while staged (NP = 2 above): sa_rec / sb_rec = the operands' H2Scale records
// (device; {scale, 1 / scale, ...}: cham_h2_scale_absmax / _rownorm of gemm_h2.hip, or a constant record for an operand with a known bound).
// Every |element x scale| must stay below 65 504 (fp16): the caller's bound is the contract, as for cham_gemm_h2.  NN (+ bias, leaky | tanh,
// row-broadcast scale on A) and TN (plain | split-K, row-broadcast scale on A) forms with N > 64 - the shapes of the scorer's first layer and
// of its weight gradient (reference nar_model.py:447-451, 478-495: matching_dense_layer_1 over cand (.) pred).
extern "C" int cham_gemm_f32x2h(const float* A, int lda, int transA, const float* B, int ldb, int transB,
                                float* C, int ldc, int M, int N, int K,
                                const float* bias, int act,
                                const float* rowscale, int ldrs, int rs_div,
                                int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                                const float* sa_rec, const float* sb_rec, void* stream) {
    if (!sa_rec || !sb_rec || N <= 64 || transB) return -CHAM_ERR_ARG;
    if (((uintptr_t)sa_rec | (uintptr_t)sb_rec) & 3) return -CHAM_ERR_ARG;
    GemmParams p;
    const int rc = gemm_plan(p, A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, nullptr, 0, 0, rowscale, ldrs, rs_div,
                             accumulate, workspace, workspace_bytes, splits_hint);
    if (rc != CHAM_OK) return rc;
    p.sa = sa_rec; p.sb = sb_rec;
    hipStream_t st = (hipStream_t)stream;
    if (!transA) return x3_by_shape<true, false, 2>(p, st);
    return x3_by_shape<false, false, 2>(p, st);
}
