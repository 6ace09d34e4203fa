// Pieces shared by the GEMM translation units (gemm.hip: native fp32 MFMA and the on-the-fly bf16 variant; gemm_x3.hip: fp32 through
// three bf16 planes): the parameter block, tile windows, the common epilogue, the bf16 staging loader and the split-K reduction.
#pragma once
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define OOB_OFF 0x80000000u          // > every window size below: loads return 0, stores are dropped
#define WINDOW_BYTES 0x7FFFF000

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_window(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, WINDOW_BYTES, 0x00020000);
}
__device__ __forceinline__ float4 as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct GemmParams {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias; int act;
    const float* dref; int ldr; int dact;
    const float* rs; int ldrs; int rs_div;
    int accumulate;
    int kchunk; int splits; float* partial;
    int nbm, nbn;
    int xcd_split;                  // split-K workgroup placement: one K-split per XCD (see the kernels' tile mapping)
    const float* sa; const float* sb;      // gemm_x3.hip, two-fp16-plane form only: the operands' H2Scale records (device); NULL otherwise
};


template <int ACT> __device__ __forceinline__ float act_fwd_c(float v) {
    if (ACT == ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (ACT == ACT_TANH) return cham_tanhf(v);
    return v;
}
template <int ACT> __device__ __forceinline__ float act_bwd_c(float y) {
    if (ACT == ACT_LEAKY) return y > 0.f ? 1.f : 0.2f;
    if (ACT == ACT_TANH) return 1.f - y * y;
    return 1.f;
}


// Shared epilogue of the fp32 and bf16 kernels (the 32x32 MFMA C/D layout is dtype independent).
// (Round 2 tried the swapped-operand form - accumulator = C^T, one row and 4 x 4 consecutive columns per lane, 16-byte stores, as in
// gemm_b16.hip: bit-identical results, 0-8 % SLOWER stand-alone on every G1 shape including the K <= 128 scorer GEMMs, because the
// dword form already writes whole 128-byte lines (32 consecutive columns per instruction).  profiles/r02_notes.md item 3, which also
// records the gfx950 store hazard found on the way.)
// EPI: 0 = plain / accumulate, 1 = bias+leaky, 2 = bias+tanh, 3 = *leaky'(dref), 4 = *tanh'(dref), 5 = bias only,
//      6 = split-K partial store
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, floatx16 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0,
                                              int split, int kl, int fl) {
    // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ------
    // windows at the tile origin; offsets are tile-local, out-of-range elements get OOB_OFF
    const int limM = p.M - m0, limN = p.N - n0;
    float* cbase = (EPI == 6) ? p.partial + ((size_t)split * p.M + m0) * p.N + n0 : p.C + (size_t)m0 * p.ldc + n0;
    const unsigned ldc = (EPI == 6) ? (unsigned)p.N : (unsigned)p.ldc;
    const __amdgpu_buffer_rsrc_t cw = make_window(cbase);
    const __amdgpu_buffer_rsrc_t dw = make_window((EPI == 3 || EPI == 4) ? p.dref + (size_t)m0 * p.ldr + n0 : p.C);
    const __amdgpu_buffer_rsrc_t biasw = make_window((EPI == 1 || EPI == 2 || EPI == 5) ? p.bias + n0 : p.C);
    const bool accum = (EPI == 0 || EPI == 3 || EPI == 4) && p.accumulate;
    if (limM >= wm0 + TM * 32 && limN >= wn0 + TN * 32) {
        // interior wave tile (wave-uniform test): no per-element address arithmetic or range selects.  The row part of an
        // element's address is uniform - c(e) * ld * 4 with c(e) = (e&3) + 8*(e>>2) - and rides in the SGPR soffset of the buffer
        // instruction; one VGPR offset per 32x32 tile.  (VALU issue slots are MFMA issue slots: the generic path below cost
        // ~6 % of a K = 1024 GEMM.)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned col = (unsigned)(wn0 + j * 32 + fl), rowb = (unsigned)(wm0 + i * 32 + 4 * kl);
                const unsigned voff = (rowb * ldc + col) * 4u;
                const unsigned voffr = (EPI == 3 || EPI == 4) ? (rowb * (unsigned)p.ldr + col) * 4u : 0u;
                float bv = 0.f;
                if (EPI == 1 || EPI == 2 || EPI == 5) bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(biasw, col * 4u, 0, 0));
                float aux[16], old[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ce = (e & 3) + 8 * (e >> 2);
                    if (EPI == 3 || EPI == 4)
                        aux[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dw, voffr, ce * p.ldr * 4, 0));
                    if (EPI == 0 || EPI == 3 || EPI == 4)
                        old[e] = accum ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cw, voff, ce * (int)ldc * 4, 0)) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ce = (e & 3) + 8 * (e >> 2);
                    float v = acc[i][j][e];
                    if (EPI == 1) v = act_fwd_c<ACT_LEAKY>(v + bv);
                    else if (EPI == 2) v = act_fwd_c<ACT_TANH>(v + bv);
                    else if (EPI == 5) v = v + bv;
                    else if (EPI == 3) v = v * act_bwd_c<ACT_LEAKY>(aux[e]) + old[e];
                    else if (EPI == 4) v = v * act_bwd_c<ACT_TANH>(aux[e]) + old[e];
                    else if (EPI == 0) v += old[e];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), cw, voff, ce * (int)ldc * 4, 0);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn0 + j * 32 + fl;
            const bool cok = col < limN;
            float bv = 0.f;
            if (EPI == 1 || EPI == 2 || EPI == 5)
                bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(biasw, cok ? (unsigned)col * 4u : OOB_OFF, 0, 0));
            unsigned offs[16];
            float aux[16], old[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                const bool ok = cok && row < limM;
                offs[e] = ok ? ((unsigned)row * ldc + (unsigned)col) * 4u : OOB_OFF;
                if (EPI == 3 || EPI == 4)
                    aux[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        dw, ok ? ((unsigned)row * (unsigned)p.ldr + (unsigned)col) * 4u : OOB_OFF, 0, 0));
                if (EPI == 0 || EPI == 3 || EPI == 4)
                    old[e] = accum ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cw, offs[e], 0, 0)) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e];
                if (EPI == 1) v = act_fwd_c<ACT_LEAKY>(v + bv);
                else if (EPI == 2) v = act_fwd_c<ACT_TANH>(v + bv);
                else if (EPI == 5) v = v + bv;
                else if (EPI == 3) v = v * act_bwd_c<ACT_LEAKY>(aux[e]) + old[e];
                else if (EPI == 4) v = v * act_bwd_c<ACT_TANH>(aux[e]) + old[e];
                else if (EPI == 0) v += old[e];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), cw, offs[e], 0, 0);
            }
        }
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 pack8(const u32x4& lo, const u32x4& hi) {
    bf16x8 v;
    v[0] = (__bf16)__uint_as_float(lo.x); v[1] = (__bf16)__uint_as_float(lo.y);
    v[2] = (__bf16)__uint_as_float(lo.z); v[3] = (__bf16)__uint_as_float(lo.w);
    v[4] = (__bf16)__uint_as_float(hi.x); v[5] = (__bf16)__uint_as_float(hi.y);
    v[6] = (__bf16)__uint_as_float(hi.z); v[7] = (__bf16)__uint_as_float(hi.w);
    return v;
}
__device__ __forceinline__ void mul4(u32x4& a, const u32x4& b) {
    a.x = __float_as_uint(__uint_as_float(a.x) * __uint_as_float(b.x)); a.y = __float_as_uint(__uint_as_float(a.y) * __uint_as_float(b.y));
    a.z = __float_as_uint(__uint_as_float(a.z) * __uint_as_float(b.z)); a.w = __float_as_uint(__uint_as_float(a.w) * __uint_as_float(b.w));
}
__device__ __forceinline__ unsigned comp(const u32x4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }
// One operand tile [BF x BK] staged as bf16.  XK: unit = 8 consecutive k of one row (2 float4).  !XK: unit = an 8(k) x 4(f)
// block (8 float4 from 8 consecutive stored rows), transposed in registers into 4 x (8 bf16 along k).
template <int BF, int BK, bool XK, int NTH, bool RS>
struct TileLoaderBF {
    static constexpr int LDK = BK + 8;
    static constexpr int NU = XK ? BF * BK / 8 : (BK / 8) * (BF / 4);     // units in the tile
    static constexpr int NV = (NU + NTH - 1) / NTH;                        // units per thread
    static constexpr int NL = XK ? 2 : 8;                                  // float4 loads per unit
    u32x4 r[NV][NL];
    u32x4 sc[RS ? NV : 1][RS ? NL : 1];      // row-broadcast scale (only instantiated for the two GEMMs that use it)
    unsigned off[NV];        // window-local byte offset of the unit's first float4 (OOB_OFF: free index out of range / no unit)
    unsigned soff[NV];
    int k8[NV];              // tile-local k of the unit's first element
    unsigned ldb4;           // byte stride between the unit's float4s (!XK: one stored row; XK: 16)

    __device__ __forceinline__ void init(int ld, int limF, int f0, int ldrs, int rs_div, bool has_rs) {
        const int tid = threadIdx.x;
        ldb4 = XK ? 16u : (unsigned)ld * 4u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * NTH;
            unsigned o; bool fok;
            if (XK) {
                const int fr = u / (BK / 8), kc = u % (BK / 8);
                o = ((unsigned)fr * (unsigned)ld + (unsigned)kc * 8u) * 4u; k8[i] = kc * 8; fok = fr < limF;
                soff[i] = has_rs ? ((unsigned)((f0 + fr) / rs_div) * (unsigned)ldrs + (unsigned)kc * 8u) * 4u : 0u;
            } else {
                const int kb = u / (BF / 4), f4 = u % (BF / 4);
                o = ((unsigned)(kb * 8) * (unsigned)ld + (unsigned)f4 * 4u) * 4u; k8[i] = kb * 8; fok = f4 * 4 < limF;
                soff[i] = 0u;
            }
            if (NU % NTH != 0 && u >= NU) fok = false;
            off[i] = fok ? o : OOB_OFF;
        }
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t win, int limK) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int kk = k8[i] + (XK ? 4 * j : j);
                const unsigned o = (off[i] != OOB_OFF && kk < limK) ? off[i] + (unsigned)j * ldb4 : OOB_OFF;
                r[i][j] = __builtin_amdgcn_raw_buffer_load_b128(win, o, 0, 0);
            }
    }
    __device__ __forceinline__ void load_scale_xk(__amdgpu_buffer_rsrc_t rsw, int k0, int limK) {
        if constexpr (RS) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const unsigned o = (off[i] != OOB_OFF && k8[i] + 4 * j < limK) ? soff[i] + 16u * j : OOB_OFF;
                    sc[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, o, k0 * 4, 0);
                }
        }
    }
    __device__ __forceinline__ void load_scale_fk(__amdgpu_buffer_rsrc_t rsw, int krow0, int f0, int ldrs, int rs_div, int limK) {
        if constexpr (RS) {
            const int tid = threadIdx.x;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int u = tid + i * NTH;
                const int f4 = u % (BF / 4);
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const unsigned o = ((unsigned)((krow0 + k8[i] + j) / rs_div) * (unsigned)ldrs + (unsigned)(f0 + f4 * 4)) * 4u;
                    sc[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (off[i] != OOB_OFF && k8[i] + j < limK) ? o : OOB_OFF, 0, 0);
                }
            }
        }
    }
    __device__ __forceinline__ void apply_scale() {
        if constexpr (RS) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j) mul4(r[i][j], sc[i][j]);
        }
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ S) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * NTH;
            if (NU % NTH != 0 && u >= NU) continue;
            if (XK) {
                const int fr = u / (BK / 8), kc = u % (BK / 8);
                *reinterpret_cast<bf16x8*>(S + fr * LDK + kc * 8) = pack8(r[i][0], r[i][1]);
            } else {
                const int kb = u / (BF / 4), f4 = u % (BF / 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (__bf16)__uint_as_float(comp(r[i][j], c));
                    *reinterpret_cast<bf16x8*>(S + (f4 * 4 + c) * LDK + kb * 8) = v;
                }
            }
        }
    }
};


// fixed-order reduction of split-K partials (+ the generic epilogue).  Four consecutive columns per thread (16-byte loads of every
// partial slab, four slabs in flight), the slabs added in ascending split order exactly as the scalar form did (bit-identical sums).
// Round 2's one-element-per-thread form with a 64-bit divide per element cost 1.1 ms per step over its ten launches.
__device__ __forceinline__ float4 splitk_finish(const GemmParams& p, float4 v, int row, int col) {
    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    v.x = act_fwd(v.x, p.act); v.y = act_fwd(v.y, p.act); v.z = act_fwd(v.z, p.act); v.w = act_fwd(v.w, p.act);
    if (p.dref) {
        const float* d = p.dref + (size_t)row * p.ldr + col;
        v.x *= act_bwd_from_out(d[0], p.dact); v.y *= act_bwd_from_out(d[1], p.dact);
        v.z *= act_bwd_from_out(d[2], p.dact); v.w *= act_bwd_from_out(d[3], p.dact);
    }
    return v;
}
static __global__ __launch_bounds__(256) void gemm_splitk_reduce(GemmParams p) {
    const size_t n = (size_t)p.M * p.N, n4 = n / 4;           // (N % 4 == 0: gemm_plan)
    const unsigned N4 = (unsigned)p.N / 4u;
    const bool vec_c = ((p.ldc & 3) == 0) && ((reinterpret_cast<size_t>(p.C) & 15) == 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4* src = reinterpret_cast<const float4*>(p.partial) + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = 0;
        for (; s + 4 <= p.splits; s += 4) {
            const float4 x0 = src[(size_t)(s + 0) * n4], x1 = src[(size_t)(s + 1) * n4], x2 = src[(size_t)(s + 2) * n4],
                         x3 = src[(size_t)(s + 3) * n4];
            v.x += x0.x; v.y += x0.y; v.z += x0.z; v.w += x0.w;
            v.x += x1.x; v.y += x1.y; v.z += x1.z; v.w += x1.w;
            v.x += x2.x; v.y += x2.y; v.z += x2.z; v.w += x2.w;
            v.x += x3.x; v.y += x3.y; v.z += x3.z; v.w += x3.w;
        }
        for (; s < p.splits; ++s) { const float4 x = src[(size_t)s * n4]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
        const int row = (int)(i / N4), col = (int)(i % N4) * 4;
        v = splitk_finish(p, v, row, col);
        float* c = p.C + (size_t)row * p.ldc + col;
        if (vec_c) {
            float4* c4 = reinterpret_cast<float4*>(c);
            if (p.accumulate) { const float4 o = *c4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *c4 = v;
        } else {
            if (p.accumulate) { v.x += c[0]; v.y += c[1]; v.z += c[2]; v.w += c[3]; }
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        }
    }
}
// Small outputs with MANY splits (the scorer's narrow weight gradients: 128 x 64 outputs, hundreds of K-splits - the form above walked
// them with 32 threads, 0.47 ms per launch on the lane the W2 weight gradient waits on): 16 threads share one group of four columns,
// thread g sums the splits g, g + 16, ... and the sixteen partial sums are added in ascending g through LDS - a fixed order.
static __global__ __launch_bounds__(256) void gemm_splitk_reduce_wide(GemmParams p) {
    __shared__ float4 red[256];
    const size_t n = (size_t)p.M * p.N, n4 = n / 4;
    const unsigned N4 = (unsigned)p.N / 4u;
    const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + o;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* src = reinterpret_cast<const float4*>(p.partial) + i;
        int s = g;
        for (; s + 48 < p.splits; s += 64) {
            const float4 x0 = src[(size_t)s * n4], x1 = src[(size_t)(s + 16) * n4], x2 = src[(size_t)(s + 32) * n4], x3 = src[(size_t)(s + 48) * n4];
            v.x += x0.x; v.y += x0.y; v.z += x0.z; v.w += x0.w;
            v.x += x1.x; v.y += x1.y; v.z += x1.z; v.w += x1.w;
            v.x += x2.x; v.y += x2.y; v.z += x2.z; v.w += x2.w;
            v.x += x3.x; v.y += x3.y; v.z += x3.z; v.w += x3.w;
        }
        for (; s < p.splits; s += 16) { const float4 x = src[(size_t)s * n4]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    if (g != 0 || i >= n4) return;
    for (int q = 1; q < 16; ++q) { const float4 x = red[q * 16 + o]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
    const int row = (int)(i / N4), col = (int)(i % N4) * 4;
    v = splitk_finish(p, v, row, col);
    float* c = p.C + (size_t)row * p.ldc + col;
    if (p.accumulate) { v.x += c[0]; v.y += c[1]; v.z += c[2]; v.w += c[3]; }
    c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
}
static inline void launch_splitk_reduce(const GemmParams& p, hipStream_t st) {
    const size_t n4 = (size_t)p.M * p.N / 4;
    if (n4 <= 32768 && p.splits >= 32) {
        hipLaunchKernelGGL(gemm_splitk_reduce_wide, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, p);
    } else {
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
    }
}


// Argument checks + the split-K plan shared by every precision (host side).
static inline int gemm_plan(GemmParams& p, const float* A, int lda, int transA, const float* B, int ldb, int transB,
                            float* C, int ldc, int M, int N, int K, const float* bias, int act, const float* dref, int ldr, int dact,
                            const float* rowscale, int ldrs, int rs_div, int accumulate, float* workspace, size_t workspace_bytes,
                            int splits_hint) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K < 0) return -CHAM_ERR_ARG;
    if ((lda & 3) || (ldb & 3)) return -CHAM_ERR_ARG;
    // contiguous extents must be float4 multiples (model dims are padded on the host side)
    if (!transA && (K & 3)) return -CHAM_ERR_ARG;        // A[M,K] row-major, k contiguous
    if (transA && (M & 3)) return -CHAM_ERR_ARG;         // A stored [K,M]
    if (!transB && (N & 3)) return -CHAM_ERR_ARG;        // B[K,N] row-major
    if (transB && (K & 3)) return -CHAM_ERR_ARG;         // B stored [N,K]
    if (rowscale && ((ldrs & 3) || rs_div <= 0)) return -CHAM_ERR_ARG;
    // tile windows address 2^31 bytes with 32-bit offsets: a 256-row (or 32-k-row) slab of any operand must fit
    if ((size_t)lda * 4 * 256 >= WINDOW_BYTES || (size_t)ldb * 4 * 256 >= WINDOW_BYTES || (size_t)ldc * 4 * 256 >= WINDOW_BYTES ||
        (size_t)ldr * 4 * 256 >= WINDOW_BYTES)
        return -CHAM_ERR_ARG;
    if (rowscale && (size_t)((transA ? K : M) / (rs_div > 0 ? rs_div : 1) + 1) * ldrs * 4 >= WINDOW_BYTES) return -CHAM_ERR_ARG;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.act = act; p.dref = dref; p.ldr = ldr; p.dact = dact;
    p.rs = rowscale; p.ldrs = ldrs; p.rs_div = rs_div > 0 ? rs_div : 1;
    p.accumulate = accumulate; p.partial = workspace; p.xcd_split = 0; p.sa = nullptr; p.sb = nullptr;
    int splits = 1;
    // the split-K reductions (gemm_splitk_reduce / _wide) walk the output in float4 groups of one row and read the bias 16 bytes at a
    // time: an output width that is not a multiple of four (only possible with transB) or a misaligned bias takes the unsplit kernel
    const bool reduce_ok = (N & 3) == 0 && (!bias || (reinterpret_cast<size_t>(bias) & 15) == 0);
    // K-splits are a weight-gradient device (TN: long reduction, small output); the NN / NT kernels have no split-K instances
    if (splits_hint != 1 && workspace && reduce_ok && transA && !transB) {
        // long-reduction / small-output shapes (wgrad): fill >= ~1024 workgroups
        const int bm = (N > 64) ? (((long)M * N >= (1L << 20)) ? 256 : 128) : 256, bn = (N > 64) ? 128 : (N > 32 ? 64 : 32);
        const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        long want = splits_hint > 1 ? splits_hint : (tiles >= 384 ? 1 : (512 + tiles - 1) / tiles);
        const long maxk = (K + 255) / 256;               // at least 256 reduction steps per split
        if (want > maxk) want = maxk;
        const long maxw = (long)(workspace_bytes / ((size_t)M * N * sizeof(float)));
        if (want > maxw) want = maxw;
        if (want > 1) splits = (int)want;
    }
    p.splits = splits;
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + 63) / 64) * 64;                  // multiple of every BK (and of gemm_x3.hip's four-tile unrolled body)
    if (kchunk == 0) kchunk = 64;
    p.kchunk = kchunk;
    p.splits = (K + kchunk - 1) / kchunk;
    if (p.splits < 1) p.splits = 1;
    return CHAM_OK;
}
