// bf16-RESIDENT MFMA GEMMs for the NAR step's bf16 configuration (BASELINE configs[2]; gfx950, wave64).
//
// Replaces the same tf.layers.Dense / matmul clusters as gemm.hip (nar_model.py:374-405 CAR, :447-500 scorer) and their
// autodiff twins, for the matrices that carry one row per CANDIDATE (B*T*(1+N) rows: 248 k at G1 shape).  In the bf16
// configuration those matrices - PreCAR / CAR activations, scorer activations, and the gradients flowing back through them - live
// in HBM as bf16, the weights have a bf16 shadow next to the fp32 master copy, accumulation / bias / activation / softmax / loss /
// Adam stay fp32.  Round 1 kept fp32 storage and rounded while staging (gemm.hip gemm_bf16_kernel): no byte was saved and the
// conversions + register transposes kept the kernels at 0.1 of either roof.
//
//   NT:  C[M,N] = epi(A[M,K] * B[N,K]^T)      A, B k-contiguous bf16 (forward with the transposed weight shadow, dgrad with the
//                                             weight as stored); C bf16 or fp32
//   TN:  C[M,N] (+)= A[K,M]^T * B[K,N]        both operands stored with the FREE index contiguous (wgrad: activations^T x
//                                             gradients, K = rows); C fp32, split-K with a deterministic fixed-order reduction
//
//   * v_mfma_f32_32x32x16_bf16 with the operands SWAPPED (B fragment first): the accumulator tile is C^T, i.e. a lane owns ONE row
//     of C and 4 x 4 CONSECUTIVE columns -> bias / saved-activation loads and the output stores are 8-byte (bf16) or 16-byte (fp32)
//     vectors per lane instead of 16 scalar accesses at a row stride.
//   * global -> register -> LDS staging in 16-byte units with NO conversion or transposition on the way: k-contiguous operands land
//     as [row][k] (row stride BK + 8 halves: conflict-free ds_read_b128 fragments); free-contiguous operands (TN) land as [k][row]
//     and the fragments - 8 consecutive k of one row per lane - are gathered by the LDS transpose read ds_read_b64_tr_b16
//     (two per fragment; layout verified on the device: tests/probe/tr_probe.hip).
//   * buffer-descriptor windows, XCD-aware tile swizzle and the one-K-split-per-XCD placement as in gemm.hip.
#include "common.h"
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define OOB_OFF 0x80000000u
#define WINDOW_BYTES 0x7FFFF000

__device__ __forceinline__ __amdgpu_buffer_rsrc_t b16_window(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, WINDOW_BYTES, 0x00020000);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;           // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

struct B16Params {
    const __bf16* A; const __bf16* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;
    const __bf16* dref; int ldr;
    int accumulate;
    int kchunk, splits; float* partial;
    int nbm, nbn, xcd_split;
};

// One operand tile staged in 16-byte units.  XK: the source is k-contiguous ([row][k], 8 consecutive k per unit, LDS image
// [row][BK + 8]); !XK: the source is free-contiguous ([k][row], 8 consecutive rows per unit, LDS image [k][BF + 8]).
template <int BF, int BK, bool XK, int NTH>
struct Stage16 {
    static constexpr int LD = XK ? BK + 8 : BF + 8;
    static constexpr int NU = BF * BK / 8;
    static constexpr int NV = (NU + NTH - 1) / NTH;
    u32x4 r[NV];
    unsigned off[NV];
    int kidx[NV];
    __device__ __forceinline__ void init(int ld, int limF) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = threadIdx.x + i * NTH;
            unsigned o; bool ok;
            if (XK) {
                const int fr = u / (BK / 8), kc = u % (BK / 8);
                o = ((unsigned)fr * (unsigned)ld + (unsigned)kc * 8u) * 2u; kidx[i] = kc * 8; ok = fr < limF;
            } else {
                const int k = u / (BF / 8), f8 = u % (BF / 8);
                o = ((unsigned)k * (unsigned)ld + (unsigned)f8 * 8u) * 2u; kidx[i] = k; ok = f8 * 8 < limF;
            }
            if (NU % NTH != 0 && u >= NU) ok = false;
            off[i] = ok ? o : OOB_OFF;
        }
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t win, int limK) {
        if (limK >= BK) {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(win, off[i], 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                r[i] = __builtin_amdgcn_raw_buffer_load_b128(win, kidx[i] < limK ? off[i] : OOB_OFF, 0, 0);
        }
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = threadIdx.x + i * NTH;
            if (NU % NTH != 0 && u >= NU) continue;
            if (XK) {
                const int fr = u / (BK / 8), kc = u % (BK / 8);
                *reinterpret_cast<u32x4*>(S + fr * LD + kc * 8) = r[i];
            } else {
                const int k = u / (BF / 8), f8 = u % (BF / 8);
                *reinterpret_cast<u32x4*>(S + k * LD + f8 * 8) = r[i];
            }
        }
    }
};

// fragment = 8 consecutive k (kk + 8 * (lane >> 5) ...) of row f0 + (lane & 31)
template <bool XK, int LD>
__device__ __forceinline__ bf16x8 frag_read(const __bf16* __restrict__ S, int f0, int kk, int lane) {
    if (XK) {
        return *reinterpret_cast<const bf16x8*>(S + (f0 + (lane & 31)) * LD + kk + 8 * (lane >> 5));
    } else {
        // [k][row] image: each 16-lane group transposes a 4(k) x 16(row) block; lane i of the group supplies the address of
        // block[i / 4][(i % 4) * 4 ..] and receives column i = rows k .. k+3 of ONE row
        const int i = lane & 15;
        const __bf16* p = S + (kk + 8 * (lane >> 5) + (i >> 2)) * LD + f0 + 16 * ((lane >> 4) & 1) + (i & 3) * 4;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + 4 * LD));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int ACT> __device__ __forceinline__ float b16_act(float v) {
    if (ACT == ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (ACT == ACT_TANH) return cham_tanhf(v);
    return v;
}

// EPI: 0 plain (fp32 out: optional accumulate), 1 bias+leaky, 2 bias+tanh, 3 x leaky'(dref), 4 x tanh'(dref), 5 bias, 6 split-K partial
// MINW: __launch_bounds__ second argument = waves per SIMD the register allocation must leave room for (1 = no constraint)
template <int BM, int BN, int WM, int WN, int BK, int MINW, bool AK, bool BKC, int EPI, bool OUTF32>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_b16_kernel(B16Params p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NTH = WM * WN * 64;
    using LA = Stage16<BM, BK, AK, NTH>;
    using LB = Stage16<BN, BK, BKC, NTH>;
    constexpr int ASZ = AK ? BM * LA::LD : BK * LA::LD, BSZ = BKC ? BN * LB::LD : BK * LB::LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* As = reinterpret_cast<__bf16*>(smem_raw);        // [2][ASZ]
    __bf16* Bs = As + 2 * ASZ;                               // [2][BSZ]   (2 * ASZ * 2 bytes is a multiple of 16)

    const int nwg = p.nbm * p.nbn;
    int tile_m, tile_n, split;
    if (p.xcd_split) {          // all tiles of one K-split on one XCD (gemm.hip, same reasoning)
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
        split = (lin & 7) + 8 * (slot / nwg);
        const int t = slot % nwg;
        tile_m = t / p.nbn; tile_n = t % p.nbn;
    } else {                    // XCD-aware bijective swizzle: consecutive tile ids (same A panel) share an XCD's L2
        const int id = blockIdx.x;
        const int q = nwg / 8, rr = nwg % 8, xcd = id % 8;
        const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + id / 8;
        tile_m = swz / p.nbn; tile_n = swz % p.nbn;
        split = blockIdx.y;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const char* aw = reinterpret_cast<const char*>(p.A) + (AK ? ((size_t)m0 * p.lda + kbeg) : ((size_t)kbeg * p.lda + m0)) * 2;
    const char* bw = reinterpret_cast<const char*>(p.B) + (BKC ? ((size_t)n0 * p.ldb + kbeg) : ((size_t)kbeg * p.ldb + n0)) * 2;
    const size_t astep = (AK ? (size_t)BK : (size_t)BK * p.lda) * 2, bstep = (BKC ? (size_t)BK : (size_t)BK * p.ldb) * 2;
    LA la; LB lb;
    la.init(p.lda, p.M - m0);
    lb.init(p.ldb, p.N - n0);
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.load(b16_window(aw), kend - kbeg);
        lb.load(b16_window(bw), kend - kbeg);
        la.store(As); lb.store(Bs);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {          // tile kt+1: loads fly behind this tile's MFMAs
            aw += astep; bw += bstep;
            const int k0 = kbeg + (kt + 1) * BK;
            la.load(b16_window(aw), kend - k0);
            lb.load(b16_window(bw), kend - k0);
        }
        const __bf16* Ac = As + cur * ASZ;
        const __bf16* Bc = Bs + cur * BSZ;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag_read<AK, LA::LD>(Ac, wm0 + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag_read<BKC, LB::LD>(Bc, wn0 + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)      // operands swapped: the accumulator tile is C^T (see the header)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            la.store(As + (cur ^ 1) * ASZ);
            lb.store(Bs + (cur ^ 1) * BSZ);
        }
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][e]: row m = wm0 + 32 i + (lane & 31), column n = wn0 + 32 j + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
    const int limM = p.M - m0, limN = p.N - n0;
    const int fl = lane & 31, kl = lane >> 5;
    constexpr bool F32 = OUTF32 || EPI == 6;
    char* cbase = (EPI == 6) ? reinterpret_cast<char*>(p.partial + ((size_t)split * p.M + m0) * p.N + n0)
                             : reinterpret_cast<char*>(p.C) + ((size_t)m0 * p.ldc + n0) * (F32 ? 4 : 2);
    const unsigned ldc = (EPI == 6) ? (unsigned)p.N : (unsigned)p.ldc;
    const __amdgpu_buffer_rsrc_t cw = b16_window(cbase);
    const __amdgpu_buffer_rsrc_t dw = b16_window((EPI == 3 || EPI == 4) ? reinterpret_cast<const char*>(p.dref + (size_t)m0 * p.ldr + n0) : cbase);
    const __amdgpu_buffer_rsrc_t biasw = b16_window((EPI == 1 || EPI == 2 || EPI == 5) ? reinterpret_cast<const char*>(p.bias + n0) : cbase);
    const bool accum = (EPI == 0) && F32 && p.accumulate;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm0 + i * 32 + fl;
        const bool mok = m < limM;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = wn0 + j * 32 + 8 * q + 4 * kl;
                const bool ok = mok && n < limN;                  // N % 4 == 0: a group of 4 columns is in or out as a whole
                float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if (EPI == 1 || EPI == 2 || EPI == 5) {
                    const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(biasw, ok ? (unsigned)n * 4u : OOB_OFF, 0, 0);
                    v[0] += __uint_as_float(bv.x); v[1] += __uint_as_float(bv.y); v[2] += __uint_as_float(bv.z); v[3] += __uint_as_float(bv.w);
                    if (EPI == 1) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = b16_act<ACT_LEAKY>(v[t]);
                    } else if (EPI == 2) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = b16_act<ACT_TANH>(v[t]);
                    }
                }
                if (EPI == 3 || EPI == 4) {
                    const u32x2 y = __builtin_amdgcn_raw_buffer_load_b64(dw, ok ? ((unsigned)m * (unsigned)p.ldr + (unsigned)n) * 2u : OOB_OFF, 0, 0);
                    const float ys[4] = {bf_lo(y.x), bf_hi(y.x), bf_lo(y.y), bf_hi(y.y)};
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] *= (EPI == 3) ? (ys[t] > 0.f ? 1.f : 0.2f) : (1.f - ys[t] * ys[t]);
                }
                if (F32) {
                    const unsigned o = ok ? ((unsigned)m * ldc + (unsigned)n) * 4u : OOB_OFF;
                    if (accum) {
                        const u32x4 old = __builtin_amdgcn_raw_buffer_load_b128(cw, o, 0, 0);
                        v[0] += __uint_as_float(old.x); v[1] += __uint_as_float(old.y); v[2] += __uint_as_float(old.z); v[3] += __uint_as_float(old.w);
                    }
                    u32x4 w;
                    w.x = __float_as_uint(v[0]); w.y = __float_as_uint(v[1]); w.z = __float_as_uint(v[2]); w.w = __float_as_uint(v[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(w, cw, o, 0, 0);
                } else {
                    u32x2 w;
                    w.x = pack_bf2(v[0], v[1]); w.y = pack_bf2(v[2], v[3]);
                    __builtin_amdgcn_raw_buffer_store_b64(w, cw, ok ? ((unsigned)m * ldc + (unsigned)n) * 2u : OOB_OFF, 0, 0);
                }
            }
    }
}

// fixed-order reduction of the split-K partials
__global__ __launch_bounds__(256) void gemm_b16_splitk_reduce(const float* __restrict__ partial, int splits, int M, int N,
                                                              float* __restrict__ C, int ldc, int accumulate) {
    const size_t n4 = (size_t)M * N / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < splits; ++s) {
            const float4 x = reinterpret_cast<const float4*>(partial + (size_t)s * M * N)[i];
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        const size_t e = i * 4;
        const int row = (int)(e / N), col = (int)(e % N);
        float4* c = reinterpret_cast<float4*>(C + (size_t)row * ldc + col);
        if (accumulate) { const float4 o = *c; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *c = v;
    }
}

// Small outputs with MANY splits (the scorer's weight gradients: a [1024, 128] output has 128 K-splits and the form above walks them with
// 128 workgroups of dependent loads - 0.32 ms per launch, the largest kernel-time item of the bf16 step in rocprofv3's summary): 16 threads
// share one group of four columns, thread g sums the splits g, g + 16, ..., the sixteen sums are added in ascending g through LDS (as
// gemm_shared.h's gemm_splitk_reduce_wide) - a fixed order.
__global__ __launch_bounds__(256) void gemm_b16_splitk_reduce_wide(const float* __restrict__ partial, int splits, int M, int N,
                                                                   float* __restrict__ C, int ldc, int accumulate) {
    __shared__ float4 red[256];
    const size_t n4 = (size_t)M * N / 4;
    const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + o;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* src = reinterpret_cast<const float4*>(partial) + i;
        int s = g;
        for (; s + 48 < splits; s += 64) {
            const float4 x0 = src[(size_t)s * n4], x1 = src[(size_t)(s + 16) * n4], x2 = src[(size_t)(s + 32) * n4], x3 = src[(size_t)(s + 48) * n4];
            v.x += x0.x; v.y += x0.y; v.z += x0.z; v.w += x0.w;
            v.x += x1.x; v.y += x1.y; v.z += x1.z; v.w += x1.w;
            v.x += x2.x; v.y += x2.y; v.z += x2.z; v.w += x2.w;
            v.x += x3.x; v.y += x3.y; v.z += x3.z; v.w += x3.w;
        }
        for (; s < splits; s += 16) { const float4 x = src[(size_t)s * n4]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    if (g != 0 || i >= n4) return;
    for (int q = 1; q < 16; ++q) { const float4 x = red[q * 16 + o]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
    const size_t e = i * 4;
    const int row = (int)(e / N), col = (int)(e % N);
    float4* c = reinterpret_cast<float4*>(C + (size_t)row * ldc + col);
    if (accumulate) { const float4 old = *c; v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
    *c = v;
}

// [0] 128x128, [1] 256x128 / 8 waves, [2] 256x128 / 4 waves, [3] 256x64, [4] 256x32; of the LAST launch: [5] fp32 output, [6] epilogue
// variant, [7] K-splits (test / bench aid, not thread-safe)
static long long g_b16_launches[8];
static int g_b16_variant = -1;
extern "C" void cham_gemm_b16_set_variant(int v) { g_b16_variant = v; }
extern "C" void cham_gemm_b16_launch_counts(long long* out8, int reset) {
    for (int i = 0; i < 8; ++i) { if (out8) out8[i] = g_b16_launches[i]; if (reset && i < 5) g_b16_launches[i] = 0; }
}

template <int BM, int BN, int WM, int WN, int BK, int MINW, bool AK, bool BKC, int EPI, bool OUTF32>
static int b16_launch_epi(B16Params& p, hipStream_t st) {
    using LA = Stage16<BM, BK, AK, WM * WN * 64>;
    using LB = Stage16<BN, BK, BKC, WM * WN * 64>;
    constexpr int ASZ = AK ? BM * LA::LD : BK * LA::LD, BSZ = BKC ? BN * LB::LD : BK * LB::LD;
    const size_t smem = (size_t)2 * (ASZ + BSZ) * 2;
    g_b16_launches[5] = OUTF32; g_b16_launches[6] = EPI; g_b16_launches[7] = p.splits;
    auto k = gemm_b16_kernel<BM, BN, WM, WN, BK, MINW, AK, BKC, EPI, OUTF32>;
    CHAM_SET_DYNAMIC_LDS(k, (int)smem);
    hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, p.splits, 1), dim3(WM * WN * 64), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

template <int BM, int BN, int WM, int WN, int BK, int MINW, bool AK, bool BKC>
static int b16_launch_cfg(B16Params& p, int act, int dact, int out_f32, hipStream_t st) {
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.N + BN - 1) / BN;
    if (!AK) {                                           // TN (wgrad): fp32 out, plain or split-K
        if (p.bias || act != ACT_NONE || p.dref || !out_f32) return -CHAM_ERR_ARG;
        if (p.splits > 1) {
            p.xcd_split = (p.splits % 8 == 0 ) ? 1 : 0;
            const int rc = b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 6, true>(p, st);
            if (rc != CHAM_OK) return rc;
            const size_t n4 = (size_t)p.M * p.N / 4;
            if (n4 <= 65536 && p.splits >= 32) {
                hipLaunchKernelGGL(gemm_b16_splitk_reduce_wide, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, p.partial, p.splits, p.M, p.N,
                                   reinterpret_cast<float*>(p.C), p.ldc, p.accumulate);
            } else {
                int blocks = (int)((n4 + 255) / 256);
                if (blocks > 4096) blocks = 4096;
                hipLaunchKernelGGL(gemm_b16_splitk_reduce, dim3(blocks), dim3(256), 0, st, p.partial, p.splits, p.M, p.N,
                                   reinterpret_cast<float*>(p.C), p.ldc, p.accumulate);
            }
            CHAM_CHECK_LAUNCH();
            return CHAM_OK;
        }
        return b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 0, true>(p, st);
    }
    if (p.splits > 1 || p.accumulate) return -CHAM_ERR_ARG;
    // NT epilogues the bf16 configuration uses (bf16 out): plain, + bias -> leaky | tanh, x leaky'(dref); plain also with fp32 out.
    // (tanh' dgrads and bias-only layers of the model run on fp32-resident operands: cham_gemm_bf16.)
    if (p.dref) {
        if (p.bias || act != ACT_NONE || out_f32 || dact != ACT_LEAKY) return -CHAM_ERR_ARG;
        return b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 3, false>(p, st);
    }
    if (p.bias) {
        if (out_f32) return -CHAM_ERR_ARG;
        if (act == ACT_LEAKY) return b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 1, false>(p, st);
        if (act == ACT_TANH) return b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 2, false>(p, st);
        return -CHAM_ERR_ARG;
    }
    if (act != ACT_NONE) return -CHAM_ERR_ARG;
    return out_f32 ? b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 0, true>(p, st) : b16_launch_epi<BM, BN, WM, WN, BK, MINW, AK, BKC, 0, false>(p, st);
}

template <bool AK, bool BKC>
static int b16_by_shape(B16Params& p, int act, int dact, int out_f32, hipStream_t st) {
    if (p.N > 64) {
        // 256x128 while the grid still covers the 256 CUs; g_b16_variant: 0 = 128x128, 1 = 256x128 / 8 waves (64x64 per wave),
        // 2 = 256x128 / 4 waves (128x64 per wave: half the LDS fragment bytes per MFMA)
        int v = ((long)((p.M + 255) / 256) * ((p.N + 127) / 128) * p.splits >= 256) ? 1 : 0;
        // TN (split-K partial epilogue: 208 VGPRs, two workgroups per CU) is faster on the 4-wave 128x64-per-wave instance (W2 wgrad
        // 785 vs 707 TFLOP/s stand-alone); the NT epilogues push that instance past 256 VGPRs and it loses (425 vs 653)
        if (v == 1 && !AK && p.splits > 1) v = 2;
        if (g_b16_variant >= 0) v = g_b16_variant;
        ++g_b16_launches[v < 3 ? v : 2];
        if (v == 1) return b16_launch_cfg<256, 128, 4, 2, 32, 1, AK, BKC>(p, act, dact, out_f32, st);
        if (v == 2) return b16_launch_cfg<256, 128, 2, 2, 32, 1, AK, BKC>(p, act, dact, out_f32, st);
        return b16_launch_cfg<128, 128, 2, 2, 32, 1, AK, BKC>(p, act, dact, out_f32, st);
    }
    if (p.N > 32) { ++g_b16_launches[3]; return b16_launch_cfg<256, 64, 4, 1, 32, 1, AK, BKC>(p, act, dact, out_f32, st); }
    ++g_b16_launches[4];
    return b16_launch_cfg<256, 32, 4, 1, 32, 1, AK, BKC>(p, act, dact, out_f32, st);
}

// C[M,N] (+)= epi(op(A) op(B)) with bf16 operands resident in HBM.
//   transA = 0, transB = 1 (NT): A [M, lda] and B [N, ldb] k-contiguous; C bf16 (out_f32 = 0) or fp32; epilogues: + bias (fp32)
//     and act (CHAM_ACT_*), or x act'(dref) with dref the SAVED bf16 activation [M, ldr] (dgrad; bf16 out).
//   transA = 1, transB = 0 (TN): A stored [K, lda >= M], B stored [K, ldb >= N]; C fp32, optional accumulate; split-K through
//     `workspace` (splits_hint: 1 = none, 0 = automatic, n = at most n; deterministic fixed-order reduction).
// Extents: K % 8 == 0 (NT) / M % 8 == 0 and N % 8 == 0 (TN), N % 4 == 0, leading dimensions % 8 == 0, 16-byte aligned bases.
extern "C" int cham_gemm_b16(const void* A, int lda, int transA, const void* B, int ldb, int transB, void* C, int ldc, int out_f32,
                             int M, int N, int K, const float* bias, int act, const void* dref, int ldr, int dact, int accumulate,
                             float* workspace, size_t workspace_bytes, int splits_hint, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return -CHAM_ERR_ARG;
    if ((lda & 7) || (ldb & 7) || (N & 3) || (ldc & 3) || (dref && (ldr & 3))) return -CHAM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)dref | (uintptr_t)bias) & 15) return -CHAM_ERR_ARG;
    const bool nt = !transA && transB, tn = transA && !transB;
    if (!nt && !tn) return -CHAM_ERR_ARG;
    if (nt && (K & 7)) return -CHAM_ERR_ARG;
    if (tn && ((M & 7) || (N & 7))) return -CHAM_ERR_ARG;
    if ((size_t)lda * 2 * 256 >= WINDOW_BYTES || (size_t)ldb * 2 * 256 >= WINDOW_BYTES || (size_t)ldc * 4 * 256 >= WINDOW_BYTES ||
        (size_t)ldr * 2 * 256 >= WINDOW_BYTES)
        return -CHAM_ERR_ARG;
    B16Params p;
    p.A = reinterpret_cast<const __bf16*>(A); p.B = reinterpret_cast<const __bf16*>(B); p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.dref = reinterpret_cast<const __bf16*>(dref); p.ldr = ldr;
    p.accumulate = accumulate; p.partial = workspace; p.xcd_split = 0;
    int splits = 1;
    if (tn && splits_hint != 1 && workspace) {
        const int bm = (N > 64) ? 256 : 256, bn = (N > 64) ? 128 : (N > 32 ? 64 : 32);
        const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        long want = splits_hint > 1 ? splits_hint : (tiles >= 384 ? 1 : (512 + tiles - 1) / tiles);
        const long maxk = (K + 511) / 512;                 // at least 512 reduction steps per split
        if (want > maxk) want = maxk;
        const long maxw = (long)(workspace_bytes / ((size_t)M * N * sizeof(float)));
        if (want > maxw) want = maxw;
        if (want >= 8) want = want / 8 * 8;                // whole K-splits per XCD
        if (want > 1) splits = (int)want;
    }
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + 31) / 32) * 32;
    p.kchunk = kchunk;
    p.splits = (K + kchunk - 1) / kchunk;
    hipStream_t st = (hipStream_t)stream;
    if (nt) return b16_by_shape<true, true>(p, act, dact, out_f32, st);
    return b16_by_shape<false, false>(p, act, dact, out_f32, st);
}
