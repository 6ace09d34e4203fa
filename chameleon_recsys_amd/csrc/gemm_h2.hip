// fp32-grade GEMM over operands that live in HBM as TWO fp16 planes + a power-of-two scale (round 4; gfx950, wave64).
//
//   x = (x_h + x_l) / s_x  (split2h of common.h),   C = (a_h b_h + a_h b_l + a_l b_h) / (s_a s_b), fp32 accumulate
//
// Why: the three candidate-row CAR GEMMs of the step (nar_model.py:374-405 of the reference: CAR forward, its dgrad, the W2 weight
// gradient - 86 % of the step's FLOPs) ran as SIX bf16 plane products per fp32 product (csrc/gemm_p3.hip) with the matrix pipe busy
// 73-81 % of the cycles and the chip clocking at 1.6 GHz: power-limited, so the only lever left was issuing fewer MFMAs
// (profiles/r03_notes.md section 1).  fp16 carries 11 significand bits against bf16's 8: two planes hold 22 bits and THREE products
// (the dropped a_l b_l term is <= 2^-22 |a b|) give a dot-product error of the native fp32 MFMA's class - what that costs is fp16's
// 5-bit exponent, paid by a per-matrix power-of-two scale that the matrix's PRODUCER derives on the device from a rigorous bound of
// max |x| (k_h2_scale_* below; no host synchronisation): |x s| <= 2^15 < 65 504, elements within 2^-18 of the bound keep all 22
// bits, smaller ones an absolute error <= 2^-40 of the bound (fp16 subnormals are kept).  Planes are 4 B / element - the footprint
// of the fp32 matrix they replace, 2/3 of the three bf16 planes.  tests/test_split2h_cpu.py (arithmetic), tests/test_gemm_h2_gpu.py
// (error next to the native fp32 MFMA on the same operands, against float64).
//
//   NT  C[M,N] = epi(A[M,K] B[N,K]^T)    planes k-contiguous: CAR forward (B = planes of W2^T), CAR dgrad (B = planes of W2)
//   TN  C[M,N] = A[K,M]^T B[K,N]         planes with the FREE index contiguous (W2 wgrad: Z1^T dZ2, K = candidate rows); split-K
//
// Core = the LDS-DMA core of gemm_p3.hip (256 x 256 x 16 tile, 8 waves as 2 x 4, 4 x 2 MFMA tiles of 32x32x16 per wave, slabs of
// 8 KB with the same source-side swizzles, `buffer_load_dwordx4 ... lds`), re-cut for four slabs per 16-k chunk:
//   stage = (A_h, A_l, B_h, B_l) = 32 KB; ring of FOUR stages = 128 KB of LDS, one workgroup per CU.
//   step i (slot i & 3):
//     top   DMA of chunk i + 3 into slot (i + 3) & 3 (its last reader was P0 of step i - 1, separated by that step's barrier)
//     P0    A_l x B_h | reads the LATE fragments of chunk i (A_h, B_l)
//     P1    A_h x B_h
//     mid   s_waitcnt vmcnt(8) - this wave's requests for chunk i + 1 have landed, the eight of chunks i + 2, i + 3 stay in flight -
//           then s_barrier
//     P2    A_h x B_l | reads the EARLY fragments of chunk i + 1 (B_h, A_l) into the registers whose last use has passed
//   Per step and wave: 24 MFMAs, 12 ds_read_b128 (24 ds_read_b64_tr_b16 for TN), 4 DMA requests, one barrier; 128 accumulator + 48
//   fragment registers.  A request has 2.5 steps (~3 800 cycles at two waves per SIMD) to land.
// Epilogues: x 1/(s_a s_b), then plain | + bias | + bias -> tanh (CAR forward) | x leaky'(sign of the saved activation's h plane) (CAR
// dgrad) | split-K partial (wgrad; partials are stored unscaled-back, i.e. in true units: the shared fixed-order reduction adds them).
#include "gemm_shared.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short h2_s16x4 __attribute__((ext_vector_type(4)));

struct H2Params {
    const _Float16* A; const _Float16* B;      // plane 0 (h) of each operand; plane 1 (l) `*_ps` elements further
    long long a_ps, b_ps;
    int lda, ldb;
    const float* sa; const float* sb;          // H2Scale records of the operands (device): [1] = 1 / scale
    float* C; int ldc;
    int M, N, K;
    const float* bias;
    const unsigned short* dref; int ldr;       // h plane of the saved activation (dgrad)
    int kchunk, splits; float* partial;
    int nbm, nbn, xcd_split, accumulate;
    // TILE-BLOCKED operands (common.h h2b_index; round 6): *_tiles > 0 = the operand's planes are stored [row tiles][ld / 32][256][32] with
    // that many row tiles ALLOCATED (rows beyond the matrix: zeros); 0 = row-major.  NT: A only (B = the weight planes); TN: A and / or B.
    int a_tiles, b_tiles, r_blk;               // r_blk: the dgrad's saved-activation h plane (dref) is tile-blocked
    // GROUP SUMS in the dgrad epilogue (EPI 3; round 6): rows come in groups of `ggrp` consecutive rows (one click's candidates) whose column
    // sums the caller needs next (dU of the PreCAR combine, csrc/scorer.hip) - 1 GB of fp32 C re-read by a kernel of its own before.  Every
    // wave adds up what its 128-row chunk q = row / 128 holds of each group it touches and writes piece (q, k) = k-th group of the chunk
    // (group (128 q) / ggrp + k) to gsum[(q * gsk + k) * N + column], gsk = 127 / ggrp + 2 = the groups a 128-row chunk can touch (<= 5 for
    // ggrp >= 32); the consumer adds the pieces of a group in chunk order.
    float* gsum; int ggrp, gsk;
};

#define H2_SLAB 8192
#define H2_STAGE (4 * H2_SLAB)
#define H2_RING 4

__device__ __forceinline__ u32x4 h2_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r.x = (unsigned)a; r.y = (unsigned)(a >> 32) & 0xFFFFu; r.z = bytes; r.w = 0x00020000u;
    return r;
}

// four LDS-DMA requests of one stage: 16 bytes per lane, LDS destination = M0 + lane * 16 (wave-uniform), source = descriptor base +
// voffset (per lane; the K advance is part of it so that the descriptor's range check sees it).  M0 is compiler-reserved: saved and
// restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void h2_dma_stage(unsigned lds0, unsigned va, unsigned vb, const u32x4& ra0, const u32x4& ra1, const u32x4& rb0,
                                             const u32x4& rb1) {
    unsigned keep;
    const unsigned l1 = lds0 + H2_SLAB, l2 = lds0 + 2 * H2_SLAB, l3 = lds0 + 3 * H2_SLAB;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, 0 offen lds\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, 0 offen lds\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %9, 0 offen lds\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %10, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0), "s"(l1), "s"(l2), "s"(l3), "v"(va), "v"(vb), "s"(ra0), "s"(ra1), "s"(rb0), "s"(rb1)
        : "memory");
}

// (TN: `hi` = byte distance of k-row + 4 in the slab image: 4 * 512 for a row-major operand's image, 4 * 64 for a tile-blocked one's)
template <bool TN>
__device__ __forceinline__ half8 h2_frag(const unsigned char* __restrict__ s, unsigned hi_off = 4 * 512) {
    if constexpr (!TN) {
        return *reinterpret_cast<const half8*>(s);
    } else {
        typedef __attribute__((address_space(3))) h2_s16x4 lds_s4;
        const h2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(s));
        const h2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(s + hi_off));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(half8, v);
    }
}

// LDS-only barrier: builtins so that the wait-count pass sees the drain; vmcnt is handled by hand (the DMA requests are invisible to
// the compiler).
__device__ __forceinline__ void h2_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    __builtin_amdgcn_sched_barrier(0);
}

// acc[i][j][e] = element (row wm0 + 32 i + (e & 3) + 8 (e >> 2) + 4 kl, column wn0 + 32 j + fl) of the tile at (m0, n0)
template <int EPI, int TM, int TNN>
__device__ __forceinline__ void h2_epilogue(const H2Params& p, floatx16 (&acc)[TM][TNN], int m0, int n0, int wm0, int wn0, int split, int lane) {
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int kl = lane_e >> 5, fl = lane_e & 31;
    // back to true units: two exact power-of-two factors, applied one after the other (their product could leave the fp32 range)
    const float ia = p.sa[1], ib = p.sb[1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] * ia) * ib;
    if constexpr (EPI == 3) {
        // dgrad: x leaky'(saved activation); the activation's sign is the h plane's bit pattern read as a signed 16-bit integer
        // (positive and non-zero <=> > 0; split2h keeps the sign of a value that underflows)
        const int limM = p.M - m0, limN = p.N - n0;
        const __amdgpu_buffer_rsrc_t cw = make_window(p.C + (size_t)m0 * p.ldc + n0);
        // saved activation: row-major [M, ldr], or tile-blocked (m0 is a multiple of 256: this tile's rows are ONE row tile; the window
        // starts at its column block n0 / 32; element (row, column block cb, column c) sits at cb * 16 KB + row * 64 + c * 2)
        const bool rb = p.r_blk != 0;
        const __amdgpu_buffer_rsrc_t dw = make_window(rb ? p.dref + ((size_t)(m0 >> 8) * (size_t)(p.ldr >> 5) + (size_t)(n0 >> 5)) * H2B_BLOCK
                                                         : p.dref + (size_t)m0 * p.ldr + n0);
        const int rstride = rb ? 64 : p.ldr * 2;              // bytes from a row to the next inside the window
        // ---- group sums (see H2Params::gsum).  Per 32-row block i the rows split at most once (ggrp >= 32): local rows < bnd belong to the
        // group the block starts in, the rest to the next one.  Lane (kl, fl) holds column fl, local rows 4 kl + (e & 3) + 8 (e >> 2): it adds its
        // values in ascending e, the two halves of the wave are added at the end of a group (a + b = b + a: both halves hold the same sum).
        const bool gs = p.gsum != nullptr;
        const int G = gs ? p.ggrp : 1, rbase = m0 + wm0;
        float* gw = gs ? p.gsum + (size_t)(rbase >> 7) * (size_t)p.gsk * (size_t)p.N + n0 + wn0 + fl : nullptr;
        float cur[TNN], hik[TNN];
#pragma unroll
        for (int j = 0; j < TNN; ++j) { cur[j] = 0.f; hik[j] = 0.f; }
        int kk = 0;
        auto emit = [&]() {
#pragma unroll
            for (int j = 0; j < TNN; ++j) {
                const float x = cur[j] + __shfl_xor(cur[j], 32, 64);
                if (kl == 0 && wn0 + j * 32 + fl < limN) gw[(size_t)kk * p.N + j * 32] = x;
            }
            ++kk;
        };
        int bnd = 64;
        auto block_begin = [&](int i) {           // wave-uniform
            if (!gs) return;
            const int s0 = rbase + 32 * i, gf = s0 / G;
            bnd = (gf + 1) * G - s0;
            if (i > 0 && bnd == G) {              // the block starts a group: the running sum is the previous group's
                emit();
#pragma unroll
                for (int j = 0; j < TNN; ++j) cur[j] = 0.f;
            }
        };
        auto block_add = [&](int j, const float (&v)[16]) {
            if (!gs) return;
            float lo = 0.f;
            if (bnd >= 32) {
#pragma unroll
                for (int e = 0; e < 16; ++e) lo += v[e];
            } else {
                float hi = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool first = (e & 3) + 8 * (e >> 2) + 4 * kl < bnd;
                    lo += first ? v[e] : 0.f;
                    hi += first ? 0.f : v[e];
                }
                hik[j] = hi;
            }
            cur[j] += lo;
        };
        auto block_end = [&]() {
            if (!gs || bnd >= 32) return;
            emit();
#pragma unroll
            for (int j = 0; j < TNN; ++j) cur[j] = hik[j];
        };
        if (limM >= wm0 + TM * 32 && limN >= wn0 + TNN * 32) {
            // interior wave tile (wave-uniform test; round 5): the row part of an element's address is uniform - c(e) * ld with
            // c(e) = (e & 3) + 8 (e >> 2) - and rides in the SGPR soffset of the buffer instructions, one VGPR offset per 32x32 block
            // and matrix: no per-element address arithmetic or range selects (the generic form below: ~10 VALU per element of an
            // epilogue that is VALU-bound; same device as gemm_epilogue's interior path)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                block_begin(i);
#pragma unroll
                for (int j = 0; j < TNN; ++j) {
                    const unsigned col = (unsigned)(wn0 + j * 32 + fl), rowb = (unsigned)(wm0 + i * 32 + 4 * kl);
                    const unsigned voff = (rowb * (unsigned)p.ldc + col) * 4u;
                    const unsigned voffr = rb ? (unsigned)((wn0 >> 5) + j) * (unsigned)(H2B_BLOCK * 2) + rowb * 64u + (unsigned)fl * 2u
                                              : (rowb * (unsigned)p.ldr + col) * 2u;
                    unsigned short y[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ce = (e & 3) + 8 * (e >> 2);
                        y[e] = __builtin_amdgcn_raw_buffer_load_b16(dw, voffr, ce * rstride, 0);
                    }
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ce = (e & 3) + 8 * (e >> 2);
                        v[e] = acc[i][j][e] * ((short)y[e] > 0 ? 1.f : 0.2f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), cw, voff, ce * p.ldc * 4, 0);
                    }
                    block_add(j, v);
                }
                block_end();
            }
            if (gs) emit();
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            block_begin(i);
#pragma unroll
            for (int j = 0; j < TNN; ++j) {
                const int col = wn0 + j * 32 + fl;
                const bool cok = col < limN;
                unsigned short y[16];
                unsigned offs[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                    const bool ok = cok && row < limM;
                    offs[e] = ok ? ((unsigned)row * (unsigned)p.ldc + (unsigned)col) * 4u : OOB_OFF;
                    const unsigned ro = rb ? (unsigned)((wn0 >> 5) + j) * (unsigned)(H2B_BLOCK * 2) + (unsigned)row * 64u + (unsigned)fl * 2u
                                           : ((unsigned)row * (unsigned)p.ldr + (unsigned)col) * 2u;
                    y[e] = __builtin_amdgcn_raw_buffer_load_b16(dw, ok ? ro : OOB_OFF, 0, 0);
                }
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    // (rows at or beyond M: the operand's rows arrive as zeros - v = 0, nothing is added to a group sum)
                    v[e] = acc[i][j][e] * ((short)y[e] > 0 ? 1.f : 0.2f);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), cw, offs[e], 0, 0);
                }
                block_add(j, v);
            }
            block_end();
        }
        if (gs) emit();
    } else {
        GemmParams g;
        g.A = nullptr; g.B = nullptr; g.C = p.C; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = 0; g.ldb = 0; g.ldc = p.ldc;
        g.bias = p.bias; g.act = ACT_NONE; g.dref = nullptr; g.ldr = 0; g.dact = ACT_NONE; g.rs = nullptr; g.ldrs = 0; g.rs_div = 1;
        g.accumulate = p.accumulate; g.kchunk = p.kchunk; g.splits = p.splits; g.partial = p.partial; g.nbm = p.nbm; g.nbn = p.nbn; g.xcd_split = p.xcd_split;
        gemm_epilogue<EPI, TM, TNN>(g, acc, m0, n0, wm0, wn0, split, kl, fl);
    }
}

// (Round 4 also tried the NT epilogue THROUGH LDS - accumulators transposed in the free ring, float4 per lane, 32 dwordx4 stores and 32
// eight-byte activation loads per wave and tile instead of 128 + 128: bit-identical, 7 % SLOWER stand-alone (1.78 vs 1.65 ms) and neutral
// in the step.  The dword form already writes whole 128-byte lines; what an NT tile pays beside its K loop - ~16 us of 103 us - is the
// epilogue's VALU work (tanh: ~2 400 instructions per lane) and the pipeline prologue, not store issue.  profiles/r04_notes.md.)
// EPI: 0 plain (/ accumulate), 2 bias + tanh, 3 x leaky'(dref h plane), 5 bias, 6 split-K partial
template <bool TN, int EPI>
__global__ __launch_bounds__(512) void gemm_h2_kernel(H2Params p) {
    constexpr int BM = 256, BN = 256, BK = 16, TM = 4, TNN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char h2_smem[];

    const int nwg = p.nbm * p.nbn;
    int tile_m, tile_n, split;
    if (p.xcd_split) {                                 // one K-split per XCD (gemm.hip): every K panel is fetched from HBM once
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
        split = (lin & 7) + 8 * (slot / nwg);
        const int t = slot % nwg;
        tile_m = t / p.nbn; tile_n = t % p.nbn;
    } else {                                           // XCD-aware bijective swizzle: the column tiles of an A panel share an L2
        const int id = blockIdx.x;
        const int q = nwg / 8, rr = nwg % 8, xcd = id % 8;
        const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + id / 8;
        tile_m = swz / p.nbn; tile_n = swz % p.nbn;
        split = blockIdx.y;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

    // ---- descriptors (one per plane: rows / k-rows beyond the operand arrive as zeros) and per-lane source offsets
    u32x4 ra[2], rb[2];
    unsigned va, vb, stepa, stepb;
    unsigned tba = 0, tbb = 0;                 // TN, tile-blocked operand: bytes of a row tile (0: row-major)
    int klim = 0;                              // TN, blocked operands: k-rows of this split from this lane's k-row on (see chunk_a)
    if constexpr (!TN) {
        // (the window starts kbeg elements into the first row: it ends that much earlier, so that requests past the reduction range
        // - issued, never consumed - cannot leave the operand's allocation)
        const size_t aall = (size_t)max(p.M - m0, 0) * p.lda * 2, ball = (size_t)max(p.N - n0, 0) * p.ldb * 2, koff = (size_t)kbeg * 2;
        const size_t abytes = aall > koff ? aall - koff : 0, bbytes = ball > koff ? ball - koff : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[q] = h2_rsrc(p.A + q * p.a_ps + (size_t)m0 * p.lda + kbeg, (unsigned)min(abytes, (size_t)0xFFFFFFF0u));
            rb[q] = h2_rsrc(p.B + q * p.b_ps + (size_t)n0 * p.ldb + kbeg, (unsigned)min(bbytes, (size_t)0xFFFFFFF0u));
        }
        // lane l of wave w fills LDS piece (row 32 w + l / 2, half l & 1); that piece holds source half (l & 1) ^ bit 3 of the row
        const int row = 32 * wave + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
        va = ((unsigned)row * (unsigned)p.lda + 8u * half) * 2u;
        vb = ((unsigned)row * (unsigned)p.ldb + 8u * half) * 2u;
        stepa = stepb = BK * 2;
    } else {
        const size_t abytes = (size_t)max(kend - kbeg, 0) * p.lda * 2, bbytes = (size_t)max(kend - kbeg, 0) * p.ldb * 2;
        // lane l of wave w fills LDS piece (k-row 2 w + l / 32, piece l & 31); its 64-byte group index is XOR-ed with k & 3
        const int k = 2 * wave + (lane >> 5), jp = lane & 31, j = ((((jp >> 2) ^ (k & 3)) << 2) | (jp & 3));
        klim = kend - kbeg - (lane >> 2);       // (tile-blocked operands: lane l fetches k-row l / 4 of its wave's column block)
        // Row-major operand: the window starts at (k-row kbeg, column m0) and ends with the split's last k-row.  Tile-blocked operand
        // (common.h h2b_index): ANOTHER lane -> piece map, because there a (16 k-row x 32-column) block is 1 KB contiguous: wave w fetches
        // column block w whole - lane l = (k-row l / 4, piece l & 3), source offset l * 16: one fully contiguous 1 KB request (the first
        // version kept the row-major map - 2 k-rows x 32 pieces = sixteen 64-byte runs 16 KB apart - and the kernel was 6 % SLOWER than on
        // row-major operands: profiles/r06_notes.md) - and the slab image becomes [column block][k-row][64 bytes] (fragment offsets below;
        // k-rows 64 bytes apart: a half-wave's transposed read covers 256 contiguous bytes, no swizzle needed).
        // The window starts at (row tile of kbeg, column block m0 / 32) and ends with the ALLOCATION; a k-row at or
        // beyond the split's end is the next split's (or, past K, whatever an earlier and longer step left in the tile): those lanes
        // request an out-of-range offset and receive zeros (chunk_a / chunk_b).  The piece (k-row, 8 m) of a lane is 16 contiguous
        // bytes in either layout - the LDS image is the same.
        if (p.a_tiles > 0) {
            tba = (unsigned)(p.lda >> 5) * (unsigned)(H2B_BLOCK * 2);
            const size_t t0 = (size_t)(kbeg >> 8), all = ((size_t)p.a_tiles - t0) * tba, skip = (size_t)(m0 >> 5) * (H2B_BLOCK * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                ra[q] = h2_rsrc(p.A + q * p.a_ps + (t0 * (size_t)(p.lda >> 5) + (size_t)(m0 >> 5)) * H2B_BLOCK, (unsigned)min(all > skip ? all - skip : (size_t)0, (size_t)0xFFFFFFF0u));
            va = (unsigned)wave * (unsigned)(H2B_BLOCK * 2) + (unsigned)lane * 16u;       // wave w: column block w, 16 k-rows x 64 bytes = 1 KB CONTIGUOUS
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)      // (M, N multiples of 256: the m / n extent of a tile never leaves its k-row)
                ra[q] = h2_rsrc(p.A + q * p.a_ps + (size_t)kbeg * p.lda + m0, (unsigned)min(abytes > (size_t)m0 * 2 ? abytes - (size_t)m0 * 2 : (size_t)0, (size_t)0xFFFFFFF0u));
            va = ((unsigned)k * (unsigned)p.lda + 8u * j) * 2u;
        }
        if (p.b_tiles > 0) {
            tbb = (unsigned)(p.ldb >> 5) * (unsigned)(H2B_BLOCK * 2);
            const size_t t0 = (size_t)(kbeg >> 8), all = ((size_t)p.b_tiles - t0) * tbb, skip = (size_t)(n0 >> 5) * (H2B_BLOCK * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                rb[q] = h2_rsrc(p.B + q * p.b_ps + (t0 * (size_t)(p.ldb >> 5) + (size_t)(n0 >> 5)) * H2B_BLOCK, (unsigned)min(all > skip ? all - skip : (size_t)0, (size_t)0xFFFFFFF0u));
            vb = (unsigned)wave * (unsigned)(H2B_BLOCK * 2) + (unsigned)lane * 16u;
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                rb[q] = h2_rsrc(p.B + q * p.b_ps + (size_t)kbeg * p.ldb + n0, (unsigned)min(bbytes > (size_t)n0 * 2 ? bbytes - (size_t)n0 * 2 : (size_t)0, (size_t)0xFFFFFFF0u));
            vb = ((unsigned)k * (unsigned)p.ldb + 8u * j) * 2u;
        }
        stepa = (unsigned)BK * (unsigned)p.lda * 2u; stepb = (unsigned)BK * (unsigned)p.ldb * 2u;
    }
    // source offset of 16-k chunk c of this split (wave-uniform): c * step, or - tile-blocked - (row tile) * tile bytes + (row in tile) * 64
    // with the row counted from the first row of kbeg's row tile
    const unsigned kin = (unsigned)(kbeg & 255);
    auto chunk_a = [&](int c) -> unsigned {
        if (!TN || tba == 0) return va + (unsigned)c * stepa;
        const unsigned r = kin + 16u * (unsigned)c;
        return 16 * c < klim ? va + (r >> 8) * tba + (r & 255u) * 64u : 0x80000000u;
    };
    auto chunk_b = [&](int c) -> unsigned {
        if (!TN || tbb == 0) return vb + (unsigned)c * stepb;
        const unsigned r = kin + 16u * (unsigned)c;
        return 16 * c < klim ? vb + (r >> 8) * tbb + (r & 255u) * 64u : 0x80000000u;
    };

    // ---- per-lane fragment offsets inside a slab (gemm_p3.hip's layouts: 16-bit elements, the element type does not matter)
    unsigned fa[TM], fb[TNN];
    if constexpr (!TN) {
        const int l31 = lane & 31;
        const unsigned fo = (unsigned)l31 * 32u + (unsigned)((lane >> 5) ^ ((l31 >> 3) & 1)) * 16u;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = (unsigned)(wm0 + 32 * i) * 32u + fo;
#pragma unroll
        for (int j = 0; j < TNN; ++j) fb[j] = (unsigned)(wn0 + 32 * j) * 32u + fo;
    } else {
        const int i16 = lane & 15, kq = 8 * (lane >> 5) + (i16 >> 2);
        const unsigned within = 32u * ((lane >> 4) & 1) + 8u * (i16 & 3);
        // row-major operand: slab = [k-row][column block ^ (k & 3)][64 B]; tile-blocked operand: slab = [column block][k-row][64 B]
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[i] = tba ? (unsigned)((wm0 >> 5) + i) * 1024u + (unsigned)kq * 64u + within
                        : (unsigned)kq * 512u + (unsigned)((((wm0 >> 5) + i) ^ (kq & 3))) * 64u + within;
#pragma unroll
        for (int j = 0; j < TNN; ++j)
            fb[j] = tbb ? (unsigned)((wn0 >> 5) + j) * 1024u + (unsigned)kq * 64u + within
                        : (unsigned)kq * 512u + (unsigned)((((wn0 >> 5) + j) ^ (kq & 3))) * 64u + within;
    }

    const unsigned hia = tba ? 4u * 64u : 4u * 512u, hib = tbb ? 4u * 64u : 4u * 512u;

    floatx16 acc[TM][TNN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_u8*)h2_smem;
    const unsigned wave_off = (unsigned)wave * 1024u;
    half8 AH[TM], AL[TM], BH[TNN], BL[TNN];

    auto mma = [&](const half8 (&X)[TM], const half8 (&Y)[TNN]) {
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
            for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[ii], Y[j], acc[ii][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {
        // prologue: chunks 0, 1, 2 (a chunk beyond the reduction range is requested all the same - it is never consumed, and the wait
        // counts stay the same in every step)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            h2_dma_stage(lds_base + (unsigned)s * H2_STAGE + wave_off, chunk_a(s), chunk_b(s), ra[0], ra[1], rb[0], rb[1]);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        h2_barrier();
        {   // early fragments of chunk 0
#pragma unroll
            for (int j = 0; j < TNN; ++j) BH[j] = h2_frag<TN>(h2_smem + 2 * H2_SLAB + fb[j], hib);
#pragma unroll
            for (int i = 0; i < TM; ++i) AL[i] = h2_frag<TN>(h2_smem + 1 * H2_SLAB + fa[i], hia);
        }
        int cur = 0;                                    // slot of chunk i
        for (int i = 0; i < nk; ++i) {
            const int nxt = (cur + 1) & (H2_RING - 1), nx3 = (cur + 3) & (H2_RING - 1);
            const unsigned char* Sc = h2_smem + cur * H2_STAGE;
            const unsigned char* Sn = h2_smem + nxt * H2_STAGE;
            __builtin_amdgcn_sched_barrier(0);
            // top: chunk i + 3
            h2_dma_stage(lds_base + (unsigned)nx3 * H2_STAGE + wave_off, chunk_a(i + 3), chunk_b(i + 3), ra[0], ra[1], rb[0], rb[1]);
            __builtin_amdgcn_sched_barrier(0);
            // P0: A_l x B_h; late fragments of this chunk
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AH[ii] = h2_frag<TN>(Sc + 0 * H2_SLAB + fa[ii], hia);
#pragma unroll
            for (int j = 0; j < TNN; ++j) BL[j] = h2_frag<TN>(Sc + 3 * H2_SLAB + fb[j], hib);
            mma(AL, BH);
            // P1: A_h x B_h
            mma(AH, BH);
            // mid: this wave's requests for chunk i + 1 have landed (the eight of chunks i + 2, i + 3 stay in flight); after the barrier
            // every wave's have
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            h2_barrier();
            // P2: A_h x B_l; early fragments of chunk i + 1 into the registers that are dead
#pragma unroll
            for (int j = 0; j < TNN; ++j) BH[j] = h2_frag<TN>(Sn + 2 * H2_SLAB + fb[j], hib);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AL[ii] = h2_frag<TN>(Sn + 1 * H2_SLAB + fa[ii], hia);
            mma(AH, BL);
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no request may outlive the workgroup's LDS allocation
    }

    h2_epilogue<EPI, TM, TNN>(p, acc, m0, n0, wm0, wn0, split, lane);
}

// ================================================================================================================================
// NT with 64-BYTE SOURCE PIECES (round 5).  The NT kernel above stages a 16-k chunk per step: an LDS-DMA request of a wave fetches
// 32 rows x 32 bytes - 32 DIFFERENT 128-byte lines, a quarter of each - and the next quarter of the same lines a step (~2 000 cycles
// and 1 000 other lines) later, when the 32 KB vector L1 has long dropped them: per step and CU 1 024 line fills = 128 KB through a
// 64 B / clk L1 fill path = 2 048 cycles, against 1 536 cycles of MFMA work per step (48 x 32) - which is the 2 160 cycles per step
// the NT forms measured (matrix pipe busy 52 % of the kernel against the TN form's 89 %, whose k-row pieces are 512 contiguous
// bytes: profiles/r04_h2_sq_counters.txt; profiles/r05_notes.md).  Here a request fetches 16 rows x 64 bytes (two 16-k chunks of a
// row: half the line fills per byte used, 1 024 cycles per step - under the MFMA time), i.e. the ring holds TWO buffers of a 32-k
// double chunk D_j = chunks (2j, 2j + 1) instead of four single-chunk stages (same 128 KB):
//   slab = [256 rows][64 bytes], four 16-byte pieces per row; LDS piece q of row r holds SOURCE piece q ^ ((r >> 2) & 3) (the swizzle
//   goes on the source address of the lane that owns a destination piece - LDS-DMA writes lane-linear - and again on the fragment
//   read): a ds_read_b128 of 16 lanes = 16 consecutive rows, one k-piece: bank group (4 r + (x ^ ((r >> 2) & 3))) mod 16, all
//   different.  k-half h of the double chunk = source pieces 2h, 2h + 1: fragment offset of half 1 = offset of half 0 ^ 32.
//   step 2j   (half 0 of D_j): top   the B half of D_{j+1} (4 requests; L2-resident operand, 1.5 steps to land)
//                              P0 | P1 | P2 as above, NO barrier: chunk 2j + 1 sits in the buffer that is being read
//   step 2j+1 (half 1 of D_j): P0 | P1 | s_waitcnt vmcnt(0) + barrier (D_{j+1} has landed; D_j's last fragment reads are behind
//                              every wave) | the A half of D_{j+2} into D_j's buffer (4 requests; the HBM-streamed operand, 2 steps
//                              to land) | P2 reads the early fragments of chunk 2j + 2 from D_{j+1}
// One barrier per TWO steps, 8 DMA requests per two steps as before.  K % 32 == 0 (the caller keeps the kernel above otherwise).
#define H2W_SLAB 16384
#define H2W_BUF (4 * H2W_SLAB)

// four requests: (r0 @ v0 -> l0), (r0 @ v1 -> l0 + 1024), (r1 @ v0 -> l1), (r1 @ v1 -> l1 + 1024): the two 16-row halves of a wave's
// 32 rows in the h plane's slab and in the l plane's slab
__device__ __forceinline__ void h2w_dma4(unsigned l0, unsigned l1, unsigned v0, unsigned v1, const u32x4& r0, const u32x4& r1) {
    unsigned keep;
    const unsigned l0b = l0 + 1024u, l1b = l1 + 1024u;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, 0 offen lds\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %7, 0 offen lds\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, 0 offen lds\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(l0), "s"(l0b), "s"(l1), "s"(l1b), "v"(v0), "v"(v1), "s"(r0), "s"(r1)
        : "memory");
}

// EPI: 0 plain, 2 bias + tanh, 3 x leaky'(dref h plane), 5 bias
// ABLK (round 6): A is TILE-BLOCKED (common.h h2b_index) - the 256 rows of this workgroup are one row tile, double chunk D_j is its column
// block j: 16 KB contiguous per plane, a request (16 rows x 64 bytes) is 1 KB of eight WHOLE 128-byte lines where the row-major operand
// gives sixteen half lines (the 1.24-1.29 x over-fetch and the L1 line-fill limit of profiles/r05_h2_sq_counters.txt).  Same 16-byte
// pieces into the same LDS slots: bit-identical results.
template <int EPI, bool ABLK>
__global__ __launch_bounds__(512) void gemm_h2w_kernel(H2Params p) {
    constexpr int BM = 256, BN = 256, TM = 4, TNN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char h2_smem[];

    const int nwg = p.nbm * p.nbn;
    const int id = blockIdx.x;
    const int q8 = nwg / 8, rr = nwg % 8, xcd = id % 8;       // XCD-aware bijective swizzle: the column tiles of an A panel share an L2
    const int swz = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + id / 8;
    const int tile_m = swz / p.nbn, tile_n = swz % p.nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nd = p.K >> 5;                                   // double chunks
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

    // ---- descriptors (one per plane: rows beyond the operand arrive as zeros) and per-lane source offsets
    u32x4 ra[2], rb[2];
    const size_t aall = (size_t)max(p.M - m0, 0) * p.lda * 2, ball = (size_t)max(p.N - n0, 0) * p.ldb * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if constexpr (ABLK) ra[q] = h2_rsrc(p.A + q * p.a_ps + (size_t)tile_m * (size_t)(p.lda >> 5) * H2B_BLOCK,
                                            (unsigned)min((size_t)(p.lda >> 5) * (H2B_BLOCK * 2), (size_t)0xFFFFFFF0u));      // this row tile (allocated whole)
        else ra[q] = h2_rsrc(p.A + q * p.a_ps + (size_t)m0 * p.lda, (unsigned)min(aall, (size_t)0xFFFFFFF0u));
        rb[q] = h2_rsrc(p.B + q * p.b_ps + (size_t)n0 * p.ldb, (unsigned)min(ball, (size_t)0xFFFFFFF0u));
    }
    // lane l of wave w, request half r: LDS piece (row 32 w + 16 r + l / 4, piece l & 3) <- source piece (l & 3) ^ ((row >> 2) & 3),
    // and (row >> 2) & 3 == (l >> 4) & 3 for every w, r
    const unsigned row = 32u * (unsigned)wave + (unsigned)(lane >> 2), sp = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    unsigned va0 = ABLK ? row * 64u + 16u * sp : (row * (unsigned)p.lda + 8u * sp) * 2u;
    unsigned va1 = va0 + (ABLK ? 16u * 64u : 16u * (unsigned)p.lda * 2u);
    constexpr unsigned stepA = ABLK ? (unsigned)(H2B_BLOCK * 2) : 64u;          // bytes from a double chunk to the next
    unsigned vb0 = (row * (unsigned)p.ldb + 8u * sp) * 2u, vb1 = vb0 + 16u * (unsigned)p.ldb * 2u;

    // ---- per-lane fragment offsets inside a slab, k-half 0 (half 1: ^ 32)
    unsigned fa0[TM], fb0[TNN], fa1[TM], fb1[TNN];
    {
        const int l31 = lane & 31;
        const unsigned fo = (unsigned)l31 * 64u + (unsigned)((lane >> 5) ^ ((l31 >> 2) & 3)) * 16u;
#pragma unroll
        for (int i = 0; i < TM; ++i) { fa0[i] = (unsigned)(wm0 + 32 * i) * 64u + fo; fa1[i] = fa0[i] ^ 32u; }
#pragma unroll
        for (int j = 0; j < TNN; ++j) { fb0[j] = (unsigned)(wn0 + 32 * j) * 64u + fo; fb1[j] = fb0[j] ^ 32u; }
    }

    floatx16 acc[TM][TNN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_u8*)h2_smem;
    const unsigned wave_off = (unsigned)wave * 2048u;
    half8 AH[TM], AL[TM], BH[TNN], BL[TNN];

    auto mma = [&](const half8 (&X)[TM], const half8 (&Y)[TNN]) {
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
            for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[ii], Y[j], acc[ii][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto dma_a = [&](unsigned buf) {          // (A_h, A_l) of the next double chunk of A
        h2w_dma4(lds_base + buf + 0 * H2W_SLAB + wave_off, lds_base + buf + 1 * H2W_SLAB + wave_off, va0, va1, ra[0], ra[1]);
        va0 += stepA; va1 += stepA;
    };
    auto dma_b = [&](unsigned buf) {          // (B_h, B_l)
        h2w_dma4(lds_base + buf + 2 * H2W_SLAB + wave_off, lds_base + buf + 3 * H2W_SLAB + wave_off, vb0, vb1, rb[0], rb[1]);
        vb0 += 64u; vb1 += 64u;
    };

    if (nd > 0) {
        // prologue: D_0 whole, the A half of D_1 (its B half goes out at the top of step 0, like every D_{j+1})
        dma_a(0u); dma_b(0u);
        if (nd > 1) {
            dma_a((unsigned)H2W_BUF);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        h2_barrier();
        {   // early fragments of chunk 0
#pragma unroll
            for (int j = 0; j < TNN; ++j) BH[j] = *reinterpret_cast<const half8*>(h2_smem + 2 * H2W_SLAB + fb0[j]);
#pragma unroll
            for (int i = 0; i < TM; ++i) AL[i] = *reinterpret_cast<const half8*>(h2_smem + 1 * H2W_SLAB + fa0[i]);
        }
        for (int j = 0; j < nd; ++j) {
            const unsigned b0 = (unsigned)(j & 1) * (unsigned)H2W_BUF, b1 = (unsigned)H2W_BUF - b0;
            const unsigned char* S0 = h2_smem + b0;           // D_j
            const unsigned char* S1 = h2_smem + b1;           // D_{j+1}
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 2j: k-half 0 of D_j.  top: the B half of D_{j+1} (its buffer's last B reads - late fragments of chunk 2j - 1 -
            // are behind the barrier of step 2j - 1)
            if (j + 1 < nd) dma_b(b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AH[ii] = *reinterpret_cast<const half8*>(S0 + 0 * H2W_SLAB + fa0[ii]);
#pragma unroll
            for (int jn = 0; jn < TNN; ++jn) BL[jn] = *reinterpret_cast<const half8*>(S0 + 3 * H2W_SLAB + fb0[jn]);
            mma(AL, BH);                                       // P0: A_l x B_h
            mma(AH, BH);                                       // P1: A_h x B_h
#pragma unroll
            for (int jn = 0; jn < TNN; ++jn) BH[jn] = *reinterpret_cast<const half8*>(S0 + 2 * H2W_SLAB + fb1[jn]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AL[ii] = *reinterpret_cast<const half8*>(S0 + 1 * H2W_SLAB + fa1[ii]);
            mma(AH, BL);                                       // P2: A_h x B_l; early fragments of chunk 2j + 1 (same buffer: no barrier)
            // ---- step 2j + 1: k-half 1 of D_j
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AH[ii] = *reinterpret_cast<const half8*>(S0 + 0 * H2W_SLAB + fa1[ii]);
#pragma unroll
            for (int jn = 0; jn < TNN; ++jn) BL[jn] = *reinterpret_cast<const half8*>(S0 + 3 * H2W_SLAB + fb1[jn]);
            mma(AL, BH);
            mma(AH, BH);
            // mid: every request of this wave for D_{j+1} has landed; after the barrier every wave's have, and every wave is past its last
            // fragment read of D_j
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            h2_barrier();
            if (j + 2 < nd) dma_a(b0);                         // the A half of D_{j+2} into D_j's buffer
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jn = 0; jn < TNN; ++jn) BH[jn] = *reinterpret_cast<const half8*>(S1 + 2 * H2W_SLAB + fb0[jn]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AL[ii] = *reinterpret_cast<const half8*>(S1 + 1 * H2W_SLAB + fa0[ii]);
            mma(AH, BL);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no request may outlive the workgroup's LDS allocation
    }

    h2_epilogue<EPI, TM, TNN>(p, acc, m0, n0, wm0, wn0, 0, lane);
}

// ================================================================================================================================
// Scales.  A record (H2Scale, 32 bytes, zero-initialised once by the caller) receives {scale, 1 / scale, bound}; words 4-6 are the
// kernels' scratch (running maxima as the bit patterns of non-negative floats - monotone as unsigned integers - and a ticket): every
// workgroup folds its maximum in with atomicMax (order-independent: bit-reproducible), the LAST one to finish - atomic ticket behind a
// __threadfence - derives the scale and clears the scratch for the next launch on the same record.
__device__ __forceinline__ void h2_finish_scale(H2Scale* rec, float bound) {
    float s = 1.f, inv = 1.f;
    if (bound > 0.f && bound < __builtin_inff()) {
        int e;
        (void)frexpf(bound, &e);                       // bound = m 2^e, 0.5 <= m < 1  ->  bound 2^(15 - e) in [2^14, 2^15)
        int k = 15 - e;
        k = k < -110 ? -110 : (k > 110 ? 110 : k);     // scale and 1 / scale stay normal fp32 numbers
        s = ldexpf(1.f, k); inv = ldexpf(1.f, -k);
    }
    rec->scale = s; rec->inv = inv; rec->bound = bound;
}

__device__ __forceinline__ float h2_block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
    __syncthreads();
    return t;
}

// bound = max |x0| + max |x1| (x1 may be NULL): a sum of two matrices (PreCAR output = leaky(U + V), |leaky(t)| <= |t|), or one matrix
__global__ __launch_bounds__(256) void k_h2_scale_absmax(const float* __restrict__ x0, size_t n0, const float* __restrict__ x1, size_t n1,
                                                         H2Scale* __restrict__ rec) {
    __shared__ float red[4];
    __shared__ unsigned last;
    float m0 = 0.f, m1 = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
#pragma unroll 4
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n0 / 4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x0)[i];
        m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (x1)
#pragma unroll 4
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n1 / 4; i += stride) {
            const float4 v = reinterpret_cast<const float4*>(x1)[i];
            m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    m0 = h2_block_max(m0, red);
    m1 = h2_block_max(m1, red);
    if (threadIdx.x == 0) {
        atomicMax(&rec->max_bits, __float_as_uint(m0));           // (every workgroup's atomics hit ONE address: they serialise at L2, so the
        if (x1) atomicMax(&rec->pad1, __float_as_uint(m1));       //  grid is capped at 512 workgroups - 1024 took 43 us on a 4 MB matrix)
        __threadfence();
        last = atomicAdd(&rec->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        if (last) {
            __threadfence();
            const float a = __uint_as_float(atomicMax(&rec->max_bits, 0u)), b = __uint_as_float(atomicMax(&rec->pad1, 0u));
            h2_finish_scale(rec, a + b);
            rec->max_bits = 0u; rec->pad1 = 0u; rec->ticket = 0u;
        }
    }
}

// bound = max over rows of ||X[r, 0:K]||_2, times *factor when given: the Cauchy-Schwarz bound of |d[r,:] . w[c,:]| over all (r, c) is
// (max row norm of d) x (max row norm of w).  K == 128: half a wave per row; otherwise a wave per row.  rec_plain (optional): a second
// record from the same pass WITHOUT the factor - max row norm of X >= max |X|, the scale of X itself as a two-plane operand
// (cham_gemm_f32x2h: dS1 in the scorer's layer-1 weight gradient).
__global__ __launch_bounds__(256) void k_h2_scale_rownorm(const float* __restrict__ X, long R, int K, int ld, const float* __restrict__ factor,
                                                          H2Scale* __restrict__ rec, H2Scale* __restrict__ rec_plain) {
    __shared__ float red[4];
    __shared__ unsigned last;
    const int lane = threadIdx.x & 63;
    const long wave_g = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    float mx = 0.f;
    if (K == 128) {
        const int hl = lane & 31, hw = lane >> 5;
        const long stride = 2 * nwaves;
        long r = 2 * wave_g + hw;
        // four rows in flight per half-wave (round 6: one 16-byte load per lane and pass left the 127 MB of dS1 at 2.6 TB/s - 49 us on the main
        // lane between the scorer's backward and the fused dgrad); a maximum: the order does not matter, the record is bit-identical
        for (; r + 3 * stride < R; r += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(X + (size_t)(r + u * stride) * ld)[hl];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s = v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                mx = fmaxf(mx, s);
            }
        }
        for (; r < R; r += stride) {
            const float4 v = reinterpret_cast<const float4*>(X + (size_t)r * ld)[hl];
            float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            mx = fmaxf(mx, s);
        }
    } else {
        for (long r = wave_g; r < R; r += nwaves) {
            float s = 0.f;
            for (int k = lane; k < K; k += 64) { const float v = X[(size_t)r * ld + k]; s += v * v; }
            mx = fmaxf(mx, wave_sum(s));
        }
    }
    mx = h2_block_max(mx, red);
    if (threadIdx.x == 0) {
        atomicMax(&rec->max_bits, __float_as_uint(mx));
        __threadfence();
        last = atomicAdd(&rec->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        if (last) {
            __threadfence();
            const float sq = __uint_as_float(atomicMax(&rec->max_bits, 0u));
            // (1 + 2^-10): the row norms are fp32 sums and the product they bound is computed in plane arithmetic - a bound must not
            // be missed by a rounding
            const float b = sqrtf(sq) * (factor ? *factor : 1.f) * 1.0009765625f;
            h2_finish_scale(rec, b);
            if (rec_plain) h2_finish_scale(rec_plain, sqrtf(sq) * 1.0009765625f);
            rec->max_bits = 0u; rec->ticket = 0u;
        }
    }
}

extern "C" int cham_h2_scale_absmax(const float* x0, size_t n0, const float* x1, size_t n1, void* rec, void* stream) {
    if (!x0 || !rec || (n0 & 3) || (x1 && (n1 & 3)) || (((uintptr_t)x0 | (uintptr_t)x1) & 15) || ((uintptr_t)rec & 15)) return -CHAM_ERR_ARG;
    size_t n = n0 > n1 ? n0 : n1;
    int blocks = (int)((n / 4 + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(k_h2_scale_absmax, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x0, n0, x1, x1 ? n1 : (size_t)0, reinterpret_cast<H2Scale*>(rec));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_h2_scale_rownorm2(const float* X, long R, int K, int ld, const float* factor, void* rec, void* rec_plain, void* stream) {
    if (!X || !rec || R < 0 || K <= 0 || ld < K || (((uintptr_t)rec | (uintptr_t)rec_plain) & 15) || rec == rec_plain) return -CHAM_ERR_ARG;
    if (K == 128 && ((ld & 3) || ((uintptr_t)X & 15))) return -CHAM_ERR_ARG;
    const long per = K == 128 ? 8 : 4;                 // rows per workgroup and pass
    long blocks = (R + per - 1) / per;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);       // (same-address atomics per workgroup: see cham_h2_scale_absmax)
    hipLaunchKernelGGL(k_h2_scale_rownorm, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, R, K, ld, factor, reinterpret_cast<H2Scale*>(rec),
                       reinterpret_cast<H2Scale*>(rec_plain));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_h2_scale_rownorm(const float* X, long R, int K, int ld, const float* factor, void* rec, void* stream) {
    return cham_h2_scale_rownorm2(X, R, K, ld, factor, rec, nullptr, stream);
}

// ---- split of an fp32 matrix into its two planes with the scale of `rec` (weights, once per step; test helper for whole operands)
// dst[q][r][c] (plane stride ps) = plane q of X[r][c] * scale; dstT[q][c][r] likewise for the transposed matrix (either may be NULL)
__global__ __launch_bounds__(256) void k_split2h(const float* __restrict__ X, int R, int Cc, int ld, _Float16* __restrict__ dst, long long ps,
                                                 int ldd, _Float16* __restrict__ dstT, long long psT, int lddT, const H2Scale* __restrict__ rec) {
    const float s = rec->scale;
    const size_t n = (size_t)R * Cc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / Cc), c = (int)(i % Cc);
        const float x = X[(size_t)r * ld + c];
        _Float16 h, l;
        split2h(x * s, h, l);
        const unsigned short hb = h2_keep_sign(h, x);
        if (dst) { _Float16* d = dst + (size_t)r * ldd + c; *reinterpret_cast<unsigned short*>(d) = hb; d[ps] = l; }
        if (dstT) { _Float16* d = dstT + (size_t)c * lddT + r; *reinterpret_cast<unsigned short*>(d) = hb; d[psT] = l; }
    }
}

// scale from max |X| into rec (recompute = 1; 0: rec already holds the scale to use), then the split
extern "C" int cham_split2h(const float* X, int R, int Cc, int ld, void* dst, long long plane_stride, int ldd, void* dstT,
                            long long plane_strideT, int lddT, void* rec, int recompute, void* stream) {
    if (!X || R <= 0 || Cc <= 0 || (!dst && !dstT) || !rec || ((uintptr_t)rec & 15)) return -CHAM_ERR_ARG;
    const size_t n = (size_t)R * Cc;
    if (recompute) {
        if (ld != Cc || (n & 3) || ((uintptr_t)X & 15)) return -CHAM_ERR_ARG;
        const int rc = cham_h2_scale_absmax(X, n, nullptr, 0, rec, stream);
        if (rc != CHAM_OK) return rc;
    }
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_split2h, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, R, Cc, ld, reinterpret_cast<_Float16*>(dst), plane_stride,
                       ldd, reinterpret_cast<_Float16*>(dstT), plane_strideT, lddT, reinterpret_cast<const H2Scale*>(rec));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// elements from one (256-row x 32-column) block of a tile-blocked plane to the next (8192 + the build's padding): a plane of `tiles` row
// tiles and leading dimension ld occupies tiles * (ld / 32) * this many elements
extern "C" int cham_h2b_block_elements(void) { return H2B_BLOCK; }

// launch counters: [0] NT launches, [1] TN launches, [2] NT launches on the 64-byte-piece kernel, [3] NT launches with a tile-blocked A,
// [4] TN launches with a tile-blocked operand, [6] epilogue and [7] K-splits of the last launch
static long long g_h2_launches[8];
extern "C" void cham_gemm_h2_launch_counts(long long* out8, int reset) {
    for (int i = 0; i < 8; ++i) { if (out8) out8[i] = g_h2_launches[i]; if (reset) g_h2_launches[i] = 0; }
}

// NT launches take the 64-byte-piece kernel (gemm_h2w_kernel) when K % 32 == 0, unless switched off (A/B arm, tests): counter [2]
static int g_h2_nt_wide = 1;
extern "C" int cham_gemm_h2_set_nt_wide(int on) { const int was = g_h2_nt_wide; g_h2_nt_wide = on ? 1 : 0; return was; }

template <int EPI, bool ABLK>
static int h2w_launch_l(H2Params& p, hipStream_t st) {
    g_h2_launches[6] = EPI; g_h2_launches[7] = 1; ++g_h2_launches[2];
    if (ABLK) ++g_h2_launches[3];
    constexpr int smem = 2 * H2W_BUF;
    auto k = gemm_h2w_kernel<EPI, ABLK>;
    CHAM_SET_DYNAMIC_LDS(k, smem);
    hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, 1, 1), dim3(512), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
template <int EPI>
static int h2w_launch(H2Params& p, hipStream_t st) {
    return p.a_tiles > 0 ? h2w_launch_l<EPI, true>(p, st) : h2w_launch_l<EPI, false>(p, st);
}

template <bool TN, int EPI>
static int h2_launch(H2Params& p, hipStream_t st) {
    g_h2_launches[6] = EPI; g_h2_launches[7] = p.splits;
    constexpr int smem = H2_RING * H2_STAGE;
    auto k = gemm_h2_kernel<TN, EPI>;
    CHAM_SET_DYNAMIC_LDS(k, smem);
    hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, p.splits, 1), dim3(512), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
// C[M,N] = epi((sum of the three plane products) / (s_a s_b)) - see the header.  A, B: plane 0 (fp16 h plane), the l plane
// `*_plane_stride` elements further; a_scale / b_scale: the operands' H2Scale records (device memory, written by the kernels above).
//   tn = 0 (NT): A [M, lda], B [N, ldb], k contiguous; K % 16 == 0.  bias (+ act = CHAM_ACT_TANH), or dref_h + dact = CHAM_ACT_LEAKY:
//     x leaky'(saved activation) with dref_h the h plane [M, ldr] of that activation.
//   tn = 1 (TN): A stored [K, lda >= M], B stored [K, ldb >= N]; M % 256 == 0, N % 256 == 0, any K; split-K through `workspace`
//     (splits_hint: 1 none, 0 automatic, n at most n; fixed-order reduction), accumulate adds to C.
// Returns -CHAM_ERR_ARG for shapes it does not take (the caller keeps cham_gemm_p3 / cham_gemm_f32x3 for those).
// cham_gemm_h2b: the same with TILE-BLOCKED operands (common.h h2b_index: [row tiles of 256][ld / 32][256][32]; round 6).  a_tiles /
// b_tiles > 0: that operand's planes are tile-blocked with that many row tiles allocated per plane (>= ceil(rows / 256), the rows beyond
// the matrix ZERO; plane stride >= tiles * 256 * ld); 0: row-major.  dref_blocked: the dgrad's saved-activation plane likewise.
//   NT: A may be blocked (needs K % 32 == 0 and the 64-byte-piece kernel, ld == K); B (the weight planes) is row-major.
//   TN: A and B independently.  Results are bit-identical to the row-major operands' (tests/test_gemm_h2_gpu.py).
extern "C" size_t cham_gemm_h2_groupsum_bytes(int M, int N, int group_rows) {
    if (M <= 0 || N <= 0 || group_rows < 32) return 0;
    return (size_t)2 * ((M + 255) / 256) * (size_t)(127 / group_rows + 2) * (size_t)N * sizeof(float);
}
static int h2_run(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride,
                  int ldb, const float* b_scale, int tn, float* C, int ldc, int M, int N, int K, const float* bias, int act,
                  const void* dref_h, int ldr, int dact, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                  int a_tiles, int b_tiles, int dref_blocked, int group_rows, float* groupsum, size_t groupsum_bytes, void* stream) {
    if (!A || !B || !C || !a_scale || !b_scale || M <= 0 || N <= 0 || K <= 0 || a_tiles < 0 || b_tiles < 0) return -CHAM_ERR_ARG;
    if (groupsum && (tn || !dref_h || group_rows < 32 || groupsum_bytes < cham_gemm_h2_groupsum_bytes(M, N, group_rows) || ((uintptr_t)groupsum & 15)))
        return -CHAM_ERR_ARG;
    {   // tile-blocked operands: whole column blocks, enough row tiles, planes that do not overlap
        const long a_rows = tn ? K : M, b_rows = tn ? K : N;
        if (a_tiles && ((lda & 31) || (long)a_tiles * 256 < a_rows || a_plane_stride < (long long)a_tiles * (lda >> 5) * H2B_BLOCK)) return -CHAM_ERR_ARG;
        if (b_tiles && ((ldb & 31) || (long)b_tiles * 256 < b_rows || b_plane_stride < (long long)b_tiles * (ldb >> 5) * H2B_BLOCK)) return -CHAM_ERR_ARG;
        if (!tn && (b_tiles || (a_tiles && (lda != K || (K & 31) || !g_h2_nt_wide)))) return -CHAM_ERR_ARG;
        if (dref_blocked && (!dref_h || (ldr & 31) || tn)) return -CHAM_ERR_ARG;      // (the caller allocates ceil(M / 256) row tiles of it)
    }
    if ((lda & 7) || (ldb & 7) || (a_plane_stride & 7) || (b_plane_stride & 7) || (N & 3) || (ldc & 3)) return -CHAM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return -CHAM_ERR_ARG;
    if ((size_t)ldc * 4 * 256 >= WINDOW_BYTES || (size_t)ldr * 2 * 256 >= WINDOW_BYTES) return -CHAM_ERR_ARG;
    H2Params p;
    p.A = reinterpret_cast<const _Float16*>(A); p.B = reinterpret_cast<const _Float16*>(B); p.a_ps = a_plane_stride; p.b_ps = b_plane_stride;
    p.lda = lda; p.ldb = ldb; p.sa = a_scale; p.sb = b_scale; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.dref = reinterpret_cast<const unsigned short*>(dref_h); p.ldr = ldr; p.partial = workspace; p.xcd_split = 0; p.accumulate = 0;
    p.nbm = (M + 255) / 256; p.nbn = (N + 255) / 256;
    p.a_tiles = a_tiles; p.b_tiles = b_tiles; p.r_blk = dref_blocked ? 1 : 0;
    p.gsum = groupsum; p.ggrp = group_rows; p.gsk = groupsum ? 127 / group_rows + 2 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (!tn) {
        if ((K & 15) || accumulate) return -CHAM_ERR_ARG;
        if ((size_t)256 * lda * 2 >= (1ull << 31) || (size_t)256 * ldb * 2 >= (1ull << 31)) return -CHAM_ERR_ARG;
        p.kchunk = K; p.splits = 1;
        if (dref_h && (bias || act != ACT_NONE || dact != ACT_LEAKY)) return -CHAM_ERR_ARG;
        if (act != ACT_NONE && !(bias && act == ACT_TANH)) return -CHAM_ERR_ARG;
        ++g_h2_launches[0];
        if (groupsum) ++g_h2_launches[5];
        if (g_h2_nt_wide && (K & 31) == 0) {
            if (dref_h) return h2w_launch<3>(p, st);
            if (bias) return act == ACT_TANH ? h2w_launch<2>(p, st) : h2w_launch<5>(p, st);
            return h2w_launch<0>(p, st);
        }
        if (dref_h) return h2_launch<false, 3>(p, st);
        if (bias) return act == ACT_TANH ? h2_launch<false, 2>(p, st) : h2_launch<false, 5>(p, st);
        return h2_launch<false, 0>(p, st);
    }
    if ((M & 255) || (N & 255) || bias || act != ACT_NONE || dref_h) return -CHAM_ERR_ARG;
    if ((size_t)16 * lda * 2 >= (1ull << 31) || (size_t)16 * ldb * 2 >= (1ull << 31)) return -CHAM_ERR_ARG;
    const long tiles = (long)p.nbm * p.nbn;
    int splits = 1;
    if (splits_hint != 1 && workspace) {
        long want = splits_hint > 1 ? splits_hint : (tiles >= 192 ? 1 : (256 + tiles - 1) / tiles);      // one workgroup per CU
        const long maxk = (K + 511) / 512;
        if (want > maxk) want = maxk;
        const long maxw = (long)(workspace_bytes / ((size_t)M * N * sizeof(float)));
        if (want > maxw) want = maxw;
        if (splits_hint <= 0 && want >= 8) want = want / 8 * 8;      // (an explicit count is taken as given)
        if (want > 1) splits = (int)want;
    }
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + 15) / 16) * 16;
    p.kchunk = kchunk;
    p.splits = (K + kchunk - 1) / kchunk;
    if ((size_t)kchunk * (lda > ldb ? lda : ldb) * 2 >= 0xFFFFFFF0ull) return -CHAM_ERR_ARG;
    if ((a_tiles || b_tiles) && ((size_t)kchunk + 512 + 64) * (size_t)(lda > ldb ? lda : ldb) * 2 >= 0xFFFFFFF0ull) return -CHAM_ERR_ARG;      // offsets from kbeg's row tile
    ++g_h2_launches[1];
    if (a_tiles || b_tiles) ++g_h2_launches[4];
    if (p.splits > 1) {
        p.xcd_split = (p.splits % 8 == 0) ? 1 : 0;
        const int rc = h2_launch<true, 6>(p, st);
        if (rc != CHAM_OK) return rc;
        GemmParams g;
        g.A = nullptr; g.B = nullptr; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = 0; g.ldb = 0; g.ldc = ldc; g.bias = nullptr; g.act = ACT_NONE;
        g.dref = nullptr; g.ldr = 0; g.dact = ACT_NONE; g.rs = nullptr; g.ldrs = 0; g.rs_div = 1; g.accumulate = accumulate;
        g.kchunk = kchunk; g.splits = p.splits; g.partial = workspace; g.nbm = p.nbm; g.nbn = p.nbn; g.xcd_split = p.xcd_split;
        launch_splitk_reduce(g, st);
        CHAM_CHECK_LAUNCH();
        return CHAM_OK;
    }
    p.accumulate = accumulate;
    return h2_launch<true, 0>(p, st);
}

extern "C" int cham_gemm_h2b(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride,
                             int ldb, const float* b_scale, int tn, float* C, int ldc, int M, int N, int K, const float* bias, int act,
                             const void* dref_h, int ldr, int dact, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                             int a_tiles, int b_tiles, int dref_blocked, void* stream) {
    return h2_run(A, a_plane_stride, lda, a_scale, B, b_plane_stride, ldb, b_scale, tn, C, ldc, M, N, K, bias, act, dref_h, ldr, dact, accumulate,
                  workspace, workspace_bytes, splits_hint, a_tiles, b_tiles, dref_blocked, 0, nullptr, 0, stream);
}
// The CAR dgrad (NT, x leaky'(dref_h)) with GROUP SUMS from its epilogue: rows come in groups of `group_rows` >= 32 consecutive rows; every
// 128-row chunk q writes the column sums of what it holds of its k-th group (group (128 q) / group_rows + k) to
// groupsum[(q * (127 / group_rows + 2) + k) * N + column] (cham_gemm_h2_groupsum_bytes(M, N, group_rows) bytes; consumer: cham_combine_bwd_gs).  C is written as by cham_gemm_h2b.
extern "C" int cham_gemm_h2_dgrad_gs(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride,
                                     int ldb, const float* b_scale, float* C, int ldc, int M, int N, int K, const void* dref_h, int ldr,
                                     int a_tiles, int dref_blocked, int group_rows, float* groupsum, size_t groupsum_bytes, void* stream) {
    if (!groupsum) return -CHAM_ERR_ARG;
    return h2_run(A, a_plane_stride, lda, a_scale, B, b_plane_stride, ldb, b_scale, 0, C, ldc, M, N, K, nullptr, ACT_NONE, dref_h, ldr, ACT_LEAKY, 0,
                  nullptr, 0, 1, a_tiles, 0, dref_blocked, group_rows, groupsum, groupsum_bytes, stream);
}

extern "C" int cham_gemm_h2(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride,
                            int ldb, const float* b_scale, int tn, float* C, int ldc, int M, int N, int K, const float* bias, int act,
                            const void* dref_h, int ldr, int dact, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                            void* stream) {
    return cham_gemm_h2b(A, a_plane_stride, lda, a_scale, B, b_plane_stride, ldb, b_scale, tn, C, ldc, M, N, K, bias, act, dref_h, ldr, dact, accumulate,
                         workspace, workspace_bytes, splits_hint, 0, 0, 0, stream);
}
