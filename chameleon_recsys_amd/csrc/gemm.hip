// fp32 MFMA GEMM with fused prologue/epilogue for the NAR step (gfx950, wave64).
//
//   C[M,N] (+)= epi( op(A)[M,K] * op(B)[K,N] )
//
// Replaces the tf.layers.Dense / tf.matmul clusters of nar_module/nar/nar_model.py:374-405 (CAR),
// :410-426 (session FCs), :447-500 (scorer), the UGRNN input projection (:1308-1361) and all of their
// autodiff twins (dgrad = NT, wgrad = TN + split-K).
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cyc/SIMD, 157.3 TFLOP/s chip peak).  Operand
// fragments are one f32 VGPR per lane: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31]
// (cdna_hip_programming.md §3), so both LDS tiles are stored k-major with the free index contiguous:
// every ds_read_b32 of a fragment is 32 consecutive dwords per half-wave = conflict free.
//
//   * operand stored with the REDUCTION index contiguous (XK=true; A of NN/NT, B of NT): float4 global
//     reads along k, transposed on the LDS write (4 x ds_write_b32, row stride BF+pad chosen so the
//     writes of a half-wave hit distinct banks).
//   * operand stored with the FREE index contiguous (XK=false; B of NN, A/B of TN): float4 global reads,
//     ds_write_b128 straight through.
//   * register-staged double buffering: tile t+1's global loads are issued before tile t's MFMAs and
//     written to the other LDS buffer after them -> one __syncthreads per K tile.
//   * XCD-aware block swizzle (8 XCDs, private L2): consecutive tile ids (which share the A panel) land
//     on the same XCD; bijective form for any grid size.
//   * prologue: optional row-broadcast scale of A (the "candidate (.) predicted-embedding" product of
//     nar_model.py:478-495 fused into the scorer's first layer and its wgrad).
//   * epilogue: bias + {none, leaky_relu(0.2), tanh}; or multiply by act'(saved output) for dgrad;
//     optional accumulate.  Split-K (grid.y) writes raw partials, reduced in fixed order (deterministic).
#include "common.h"

struct GemmParams {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias; int act;
    const float* dref; int ldr; int dact;
    const float* rs; int ldrs; int rs_div;
    int accumulate;
    int kchunk; int splits; float* partial;
    int nbm, nbn;
};

template <int BF, int BK, bool XK, int NTH>
struct TileLoader {
    static constexpr int NF4 = BF * BK / 4;               // float4 in the tile
    static constexpr int NV = (NF4 + NTH - 1) / NTH;      // float4 per thread
    static constexpr int LD = XK ? (BF + (BK == 16 ? 2 : 1)) : (BF + 4);
    float4 r[NV];
    float4 sc[NV];                                        // row-broadcast scale (multiplied at store time)

    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int f0, int F, int k0, int kend,
                                         const float* __restrict__ rs, int ldrs, int rs_div) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTH;
            int srow, scol;   // stored row / col of this float4
            bool ok;
            if (NF4 % NTH != 0 && idx >= NF4) { r[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            if (XK) {
                const int fr = idx / (BK / 4), kq = idx % (BK / 4);
                srow = f0 + fr; scol = k0 + kq * 4;
                ok = (srow < F) && (scol < kend);
            } else {
                const int kk = idx / (BF / 4), f4 = idx % (BF / 4);
                srow = k0 + kk; scol = f0 + f4 * 4;
                ok = (srow < kend) && (scol < F);
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                v = *reinterpret_cast<const float4*>(base + (size_t)srow * ld + scol);
                if (rs) sc[i] = *reinterpret_cast<const float4*>(rs + (size_t)(srow / rs_div) * ldrs + scol);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void apply_scale() {      // after the loads have landed, just before the LDS store
#pragma unroll
        for (int i = 0; i < NV; ++i) { r[i].x *= sc[i].x; r[i].y *= sc[i].y; r[i].z *= sc[i].z; r[i].w *= sc[i].w; }
    }
    __device__ __forceinline__ void store(float* __restrict__ S) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTH;
            if (NF4 % NTH != 0 && idx >= NF4) continue;
            if (XK) {
                const int fr = idx / (BK / 4), kq = idx % (BK / 4);
                float* d = S + (kq * 4) * LD + fr;
                d[0] = r[i].x; d[LD] = r[i].y; d[2 * LD] = r[i].z; d[3 * LD] = r[i].w;
            } else {
                const int kk = idx / (BF / 4), f4 = idx % (BF / 4);
                *reinterpret_cast<float4*>(S + kk * LD + f4 * 4) = r[i];
            }
        }
    }
};

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NTH = WM * WN * 64;
    using LA = TileLoader<BM, BK, AK, NTH>;
    using LB = TileLoader<BN, BK, BKC, NTH>;
    constexpr int LDA = LA::LD, LDB = LB::LD;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BK][LDA]
    float* Bs = smem + 2 * ASZ;       // [2][BK][LDB]   (2*ASZ*4 bytes is a multiple of 16 for every instance)

    // ---- XCD-aware, bijective tile mapping -------------------------------------------------------
    const int nwg = p.nbm * p.nbn;
    const int id = blockIdx.x;
    const int q = nwg / 8, rr = nwg % 8, xcd = id % 8;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + id / 8;
    const int tile_m = swz / p.nbn, tile_n = swz % p.nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    LA la; LB lb;
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.load(p.A, p.lda, m0, p.M, kbeg, kend, p.rs, p.ldrs, p.rs_div);
        lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, nullptr, 0, 1);
        if (p.rs) la.apply_scale();
        la.store(As); lb.store(Bs);
    }
    __syncthreads();
    const int kl = lane >> 5, fl = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = kbeg + (kt + 1) * BK;
            la.load(p.A, p.lda, m0, p.M, k0, kend, p.rs, p.ldrs, p.rs_div);
            lb.load(p.B, p.ldb, n0, p.N, k0, kend, nullptr, 0, 1);
        }
        const float* Ac = As + cur * ASZ + wm0 + fl;
        const float* Bc = Bs + cur * BSZ + wn0 + fl;
        // fragment registers are double-buffered: the ds_reads of k-step kk+2 are issued before the MFMAs of kk
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = Ac[kl * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bc[kl * LDB + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int c = (kk >> 1) & 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[c ^ 1][i] = Ac[(kk + 2 + kl) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[c ^ 1][j] = Bc[(kk + 2 + kl) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            if (p.rs) la.apply_scale();
            la.store(As + (cur ^ 1) * ASZ);
            lb.store(Bs + (cur ^ 1) * BSZ);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + j * 32 + fl;
            if (col >= p.N) continue;
            const float bv = (p.bias != nullptr && p.splits == 1) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                if (row >= p.M) continue;
                float v = acc[i][j][e];
                if (p.splits > 1) {
                    p.partial[((size_t)split * p.M + row) * p.N + col] = v;
                } else {
                    v = act_fwd(v + bv, p.act);
                    if (p.dref) v *= act_bwd_from_out(p.dref[(size_t)row * p.ldr + col], p.dact);
                    float* c = p.C + (size_t)row * p.ldc + col;
                    *c = p.accumulate ? (*c + v) : v;
                }
            }
        }
}

// fixed-order reduction of split-K partials (+ the same epilogue)
__global__ __launch_bounds__(256) void gemm_splitk_reduce(GemmParams p) {
    const size_t n = (size_t)p.M * p.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.partial[(size_t)s * n + i];
        const int row = (int)(i / p.N), col = (int)(i % p.N);
        if (p.bias) v += p.bias[col];
        v = act_fwd(v, p.act);
        if (p.dref) v *= act_bwd_from_out(p.dref[(size_t)row * p.ldr + col], p.dact);
        float* c = p.C + (size_t)row * p.ldc + col;
        *c = p.accumulate ? (*c + v) : v;
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC>
static int launch_cfg(GemmParams& p, hipStream_t st) {
    using LA = TileLoader<BM, BK, AK, WM * WN * 64>;
    using LB = TileLoader<BN, BK, BKC, WM * WN * 64>;
    const size_t smem = (size_t)2 * BK * (LA::LD + LB::LD) * sizeof(float);
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.N + BN - 1) / BN;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, BK, AK, BKC>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return -CHAM_ERR_LAUNCH;
        attr_done = true;
    }
    dim3 grid(p.nbm * p.nbn, p.splits, 1);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, p);
    CHAM_CHECK_LAUNCH();
    if (p.splits > 1) {
        const size_t n = (size_t)p.M * p.N;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
        CHAM_CHECK_LAUNCH();
    }
    return CHAM_OK;
}

static int g_variant = -1;     // -1 = automatic
extern "C" void cham_gemm_set_variant(int v) { g_variant = v; }

template <bool AK, bool BKC>
static int launch_by_shape(GemmParams& p, hipStream_t st) {
    if (p.N > 64) {
        // default: 256x128 tile / 8 waves for large outputs (best on MI355X: 113-117 TFLOP/s on the CAR shapes),
        // 128x128 / 4 waves when the grid would otherwise be too small to fill 256 CUs
        const int v = g_variant >= 0 ? g_variant : (((long)p.M * p.N >= (1L << 20)) ? 2 : 0);
        switch (v) {
            case 1: return launch_cfg<128, 128, 2, 2, 32, AK, BKC>(p, st);
            case 2: return launch_cfg<256, 128, 4, 2, 16, AK, BKC>(p, st);
            case 3: return launch_cfg<128, 256, 2, 2, 16, AK, BKC>(p, st);
            case 4: return launch_cfg<256, 256, 4, 2, 16, AK, BKC>(p, st);
            case 5: return launch_cfg<256, 128, 2, 2, 16, AK, BKC>(p, st);
            default: return launch_cfg<128, 128, 2, 2, 16, AK, BKC>(p, st);
        }
    }
    if (p.N > 32) return launch_cfg<256, 64, 4, 1, 16, AK, BKC>(p, st);
    return launch_cfg<256, 32, 4, 1, 16, AK, BKC>(p, st);
}

extern "C" int cham_gemm_f32(const float* A, int lda, int transA, const float* B, int ldb, int transB,
                             float* C, int ldc, int M, int N, int K,
                             const float* bias, int act,
                             const float* dref, int ldr, int dact,
                             const float* rowscale, int ldrs, int rs_div,
                             int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                             void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K < 0) return -CHAM_ERR_ARG;
    if ((lda & 3) || (ldb & 3)) return -CHAM_ERR_ARG;
    // contiguous extents must be float4 multiples (model dims are padded on the host side)
    if (!transA && (K & 3)) return -CHAM_ERR_ARG;        // A[M,K] row-major, k contiguous
    if (transA && (M & 3)) return -CHAM_ERR_ARG;         // A stored [K,M]
    if (!transB && (N & 3)) return -CHAM_ERR_ARG;        // B[K,N] row-major
    if (transB && (K & 3)) return -CHAM_ERR_ARG;         // B stored [N,K]
    if (rowscale && ((ldrs & 3) || rs_div <= 0)) return -CHAM_ERR_ARG;
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.act = act; p.dref = dref; p.ldr = ldr; p.dact = dact;
    p.rs = rowscale; p.ldrs = ldrs; p.rs_div = rs_div > 0 ? rs_div : 1;
    p.accumulate = accumulate; p.partial = workspace;
    int splits = 1;
    if (splits_hint != 1 && workspace) {
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
    kchunk = ((kchunk + 15) / 16) * 16;                  // multiple of BK
    if (kchunk == 0) kchunk = 16;
    p.kchunk = kchunk;
    p.splits = (K + kchunk - 1) / kchunk;
    if (p.splits < 1) p.splits = 1;
    hipStream_t st = (hipStream_t)stream;
    if (!transA && !transB) return launch_by_shape<true, false>(p, st);
    if (!transA && transB) return launch_by_shape<true, true>(p, st);
    if (transA && !transB) return launch_by_shape<false, false>(p, st);
    return -CHAM_ERR_ARG;                                // TT never occurs on this path
}
