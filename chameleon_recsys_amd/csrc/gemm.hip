// fp32 MFMA GEMM with fused prologue/epilogue for the NAR step (gfx950, wave64).
//
//   C[M,N] (+)= epi( op(A)[M,K] * op(B)[K,N] )
//
// Replaces the tf.layers.Dense / tf.matmul clusters of nar_module/nar/nar_model.py:374-405 (CAR),
// :410-426 (session FCs), :447-500 (scorer), the RNN input projection (:1308-1361) and all of their
// autodiff twins (dgrad = NT, wgrad = TN + split-K).
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cyc/SIMD, 157.3 TFLOP/s chip peak).  Operand
// fragments are one f32 VGPR per lane: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31]
// (cdna_hip_programming.md section 3), so both LDS tiles are stored k-major with the free index contiguous:
// every ds_read_b32 of a fragment is 32 consecutive dwords per half-wave = conflict free.
//
//   * ALL global traffic goes through buffer descriptors (buffer_load/store_dword[x4] ... offen) whose base is
//     the workgroup's tile window (wave-uniform, advanced along K with scalar adds): ragged edges are handled
//     by steering the 32-bit offset of an out-of-range element to an out-of-window sentinel - the hardware
//     returns 0 / drops the store.  No divergent branch around any memory operation: the loads of K-tile t+1
//     stay in flight behind the MFMAs of tile t (with `if (ok) load` hipcc drained vmcnt(0) right after every
//     load and exposed the full L2/HBM latency once per K-tile).
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
//   * epilogue (compile-time specialised): bias + {none, leaky_relu(0.2), tanh}; or multiply by act'(saved output)
//     for dgrad; optional accumulate.  Split-K (grid.y) writes raw partials, reduced in fixed order (deterministic).
#include "gemm_shared.h"
#include <stdlib.h>

// One operand tile [BF x BK].  Offsets are tile-window-local bytes, computed once; per K tile only the k-validity
// select and the loads remain.
template <int BF, int BK, bool XK, int NTH>
struct TileLoader {
    static constexpr int NF4 = BF * BK / 4;               // float4 in the tile
    static constexpr int NV = (NF4 + NTH - 1) / NTH;      // float4 per thread
    static constexpr int LD = XK ? (BF + (BK == 16 ? 2 : 1)) : (BF + 4);
    u32x4 r[NV];
    u32x4 sc[NV];                                         // row-broadcast scale (multiplied at store time)
    unsigned off[NV];                                     // window-local byte offset (OOB_OFF when the free index is out of range)
    unsigned soff[NV];                                    // scale offset (XK only: affine in k0 -> soffset)
    int kidx[NV];                                         // tile-local k of this float4

    __device__ __forceinline__ void init(int ld, int limF, int f0, int ldrs, int rs_div, bool has_rs) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTH;
            unsigned o; int kk; bool fok;
            if (XK) {
                const int fr = idx / (BK / 4), kq = idx % (BK / 4);
                o = ((unsigned)fr * (unsigned)ld + (unsigned)kq * 4u) * 4u; kk = kq * 4; fok = fr < limF;
                soff[i] = has_rs ? ((unsigned)((f0 + fr) / rs_div) * (unsigned)ldrs + (unsigned)kq * 4u) * 4u : 0u;
            } else {
                const int k = idx / (BF / 4), f4 = idx % (BF / 4);
                o = ((unsigned)k * (unsigned)ld + (unsigned)f4 * 4u) * 4u; kk = k; fok = f4 * 4 < limF;
                soff[i] = 0u;
            }
            if (NF4 % NTH != 0 && idx >= NF4) fok = false;
            off[i] = fok ? o : OOB_OFF;
            kidx[i] = kk;
        }
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t win, int limK) {
        if (limK >= BK) {            // full K tile (wave-uniform): no per-load compare / select in the steady-state loop
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(win, off[i], 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const unsigned o = kidx[i] < limK ? off[i] : OOB_OFF;
                r[i] = __builtin_amdgcn_raw_buffer_load_b128(win, o, 0, 0);
            }
        }
    }
    // XK operand (A of NN): scale[(row / rs_div), k] - rows fixed per thread, k advances with the tile (soffset)
    __device__ __forceinline__ void load_scale_xk(__amdgpu_buffer_rsrc_t rsw, int k0, int limK) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned o = (kidx[i] < limK && off[i] != OOB_OFF) ? soff[i] : OOB_OFF;
            sc[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, o, k0 * 4, 0);
        }
    }
    // non-XK operand (A of TN, stored [K rows, M cols]): scale[((k0+kk) / rs_div), f0 + col]
    __device__ __forceinline__ void load_scale_fk(__amdgpu_buffer_rsrc_t rsw, int krow0, int f0, int ldrs, int rs_div, int limK) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTH;
            const int f4 = idx % (BF / 4);
            const unsigned o = ((unsigned)((krow0 + kidx[i]) / rs_div) * (unsigned)ldrs + (unsigned)(f0 + f4 * 4)) * 4u;
            sc[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (kidx[i] < limK && off[i] != OOB_OFF) ? o : OOB_OFF, 0, 0);
        }
    }
    __device__ __forceinline__ void apply_scale() {      // after the loads have landed, just before the LDS store
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i].x = __float_as_uint(__uint_as_float(r[i].x) * __uint_as_float(sc[i].x));
            r[i].y = __float_as_uint(__uint_as_float(r[i].y) * __uint_as_float(sc[i].y));
            r[i].z = __float_as_uint(__uint_as_float(r[i].z) * __uint_as_float(sc[i].z));
            r[i].w = __float_as_uint(__uint_as_float(r[i].w) * __uint_as_float(sc[i].w));
        }
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
                d[0] = __uint_as_float(r[i].x); d[LD] = __uint_as_float(r[i].y);
                d[2 * LD] = __uint_as_float(r[i].z); d[3 * LD] = __uint_as_float(r[i].w);
            } else {
                const int kk = idx / (BF / 4), f4 = idx % (BF / 4);
                *reinterpret_cast<float4*>(S + kk * LD + f4 * 4) = as_f4(r[i]);
            }
        }
    }
};

// compile-time switch of the ds_read / MFMA interleave hint: on for the tile configurations with one wave per SIMD
template <int BM, int BN, int WM, int WN>
__host__ __device__ constexpr bool g_sched_hint_static() { return WM * WN <= 4 && BM * BN >= 256 * 256; }

// EPI: see gemm_epilogue.  (The ablation bits and the non-pipelined K loop of the round-1 probes are gone: profiles/r01_notes.md items 3, 9.)
template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, int EPI, bool RS = false>
__device__ __forceinline__ void gemm_f32_body(const GemmParams& p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NTH = WM * WN * 64;
    using LA = TileLoader<BM, BK, AK, NTH>;
    using LB = TileLoader<BN, BK, BKC, NTH>;
    constexpr int LDA = LA::LD, LDB = LB::LD;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BK][LDA]
    float* Bs = smem + 2 * ASZ;       // [2][BK][LDB]   (2*ASZ*4 bytes is a multiple of 16 for every instance)

    // ---- XCD-aware, bijective tile mapping (all scalar) --------------------------------------------
    const int nwg = p.nbm * p.nbn;
    int tile_m, tile_n, split;
    if (p.xcd_split) {
        // split-K with a small output (wgrad): ALL tiles of one K-split on the same XCD, so that each K-panel of A and B is
        // fetched from HBM once and shared through that XCD's L2 (tile-major placement re-read every panel on ~3 XCDs: the W2
        // wgrad fetched 6.3 GB for 2.07 GB of operands).  Workgroups are dealt to the 8 XCDs round-robin in dispatch order.
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

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: SGPR tile offsets)
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // tile windows: byte pointers advanced along K by scalar adds
    const char* aw = reinterpret_cast<const char*>(p.A) + (AK ? ((size_t)m0 * p.lda + kbeg) : ((size_t)kbeg * p.lda + m0)) * 4;
    const char* bw = reinterpret_cast<const char*>(p.B) + (BKC ? ((size_t)n0 * p.ldb + kbeg) : ((size_t)kbeg * p.ldb + n0)) * 4;
    const size_t astep = (AK ? (size_t)BK : (size_t)BK * p.lda) * 4, bstep = (BKC ? (size_t)BK : (size_t)BK * p.ldb) * 4;
    constexpr bool has_rs = RS;          // compile-time: the scale product / select must not cost VALU slots in every GEMM
    const __amdgpu_buffer_rsrc_t rsw = make_window(p.rs);

    LA la; LB lb;
    la.init(p.lda, p.M - m0, m0, p.ldrs, p.rs_div, has_rs);
    lb.init(p.ldb, p.N - n0, n0, 0, 1, false);
    const int nk = (kend - kbeg + BK - 1) / BK;
    // Software pipeline, prefetch distance 2: during the MFMAs of K-tile t the wave (1) writes tile t+1 - loaded one full tile
    // ago, so its vmcnt wait is free - into the other LDS buffer and (2) issues the global loads of tile t+2 into the freed
    // registers.  Every memory operation of the loop is in flight behind matrix work; the barrier at the end of a tile only
    // collects stragglers (before: wait -> ds_write -> wait -> barrier -> ds_read sat exposed between two MFMA phases).
    auto load_tile = [&](int t) {
        const int k0 = kbeg + t * BK;
        la.load(make_window(aw), kend - k0);
        lb.load(make_window(bw), kend - k0);
        if (has_rs) {
            if (AK) la.load_scale_xk(rsw, k0, kend - k0);
            else la.load_scale_fk(rsw, k0, m0, p.ldrs, p.rs_div, kend - k0);
        }
        aw += astep; bw += bstep;
    };
    if (nk > 0) {
        load_tile(0);
        if (has_rs) la.apply_scale();
        la.store(As); lb.store(Bs);
        if (nk > 1) load_tile(1);
    }
    __syncthreads();
    const int kl = lane >> 5, fl = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // one base register per fragment (kept opaque so that the compiler addresses every k-step with the 16-bit immediate of
        // ds_read_b32 instead of re-basing a ds_read2_b32 pair with a v_add per k-step: VALU slots are MFMA slots)
        int ai[TM], bj[TN];          // indices into smem[] (the array keeps its LDS address space -> ds_read_b32 base + immediate)
#pragma unroll
        for (int i = 0; i < TM; ++i) { ai[i] = cur * ASZ + wm0 + fl + kl * LDA + i * 32; asm volatile("" : "+v"(ai[i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bj[j] = 2 * ASZ + cur * BSZ + wn0 + fl + kl * LDB + j * 32; asm volatile("" : "+v"(bj[j])); }
        // fragment registers are double-buffered: the ds_reads of k-step kk+2 are issued before the MFMAs of kk
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = smem[ai[i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = smem[bj[j]];
        if (kt + 1 < nk) {        // tile kt+1: registers -> the LDS buffer every wave finished reading at the last barrier
            if (has_rs) la.apply_scale();
            la.store(As + (cur ^ 1) * ASZ);
            lb.store(Bs + (cur ^ 1) * BSZ);
            if (kt + 2 < nk) load_tile(kt + 2);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int c = (kk >> 1) & 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[c ^ 1][i] = smem[ai[i] + (kk + 2) * LDA];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[c ^ 1][j] = smem[bj[j] + (kk + 2) * LDB];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            if (g_sched_hint_static<BM, BN, WM, WN>()) {     // interleave the next fragments' ds_reads between this step's MFMAs
#pragma unroll
                for (int r = 0; r < TM * TN; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (r < TM + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __syncthreads();
    }

    gemm_epilogue<EPI, TM, TN>(p, acc, m0, n0, wm0, wn0, split, kl, fl);
}

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, int EPI, bool RS = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmParams p) {
    gemm_f32_body<BM, BN, WM, WN, BK, AK, BKC, EPI, RS>(p);
}

// =====================================================================================================================
// bf16-input variant (BASELINE config 3: "bf16 compute, fp32 master weights + fp32 softmax / loss / Adam").
// Storage stays fp32 everywhere; operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they are staged into LDS and
// multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation - 16x the fp32 matrix rate, so these GEMMs become
// bound by moving the fp32 operands (HBM / L2), not by the matrix cores.  Same tiles, windows, swizzle and epilogues.
// LDS image: [row][k] bf16, k contiguous, row stride BK + 8 (80 B at BK = 32: 5 x 16-B slots, coprime with the 16 slots of
// a 256-B bank row -> every ds_read_b128 lane group hits 16 distinct slots); fragment = 8 consecutive k per lane:
// A[i = lane&31][k = 8*(lane>>5) .. +7], B[k = 8*(lane>>5) .. +7][j = lane&31].
template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, int EPI, bool RS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(GemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NTH = WM * WN * 64;
    using LA = TileLoaderBF<BM, BK, AK, NTH, RS>;
    using LB = TileLoaderBF<BN, BK, BKC, NTH, false>;
    constexpr int LDK = BK + 8;
    constexpr int ASZ = BM * LDK, BSZ = BN * LDK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);      // [2][BM][LDK]
    __bf16* Bs = As + 2 * ASZ;                         // [2][BN][LDK]

    const int nwg = p.nbm * p.nbn;
    int tile_m, tile_n, split;
    if (p.xcd_split) {
        // split-K with a small output (wgrad): ALL tiles of one K-split on the same XCD, so that each K-panel of A and B is
        // fetched from HBM once and shared through that XCD's L2 (tile-major placement re-read every panel on ~3 XCDs: the W2
        // wgrad fetched 6.3 GB for 2.07 GB of operands).  Workgroups are dealt to the 8 XCDs round-robin in dispatch order.
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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: SGPR tile offsets)
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const char* aw = reinterpret_cast<const char*>(p.A) + (AK ? ((size_t)m0 * p.lda + kbeg) : ((size_t)kbeg * p.lda + m0)) * 4;
    const char* bw = reinterpret_cast<const char*>(p.B) + (BKC ? ((size_t)n0 * p.ldb + kbeg) : ((size_t)kbeg * p.ldb + n0)) * 4;
    const size_t astep = (AK ? (size_t)BK : (size_t)BK * p.lda) * 4, bstep = (BKC ? (size_t)BK : (size_t)BK * p.ldb) * 4;
    constexpr bool has_rs = RS;
    const __amdgpu_buffer_rsrc_t rsw = make_window(p.rs);

    LA la; LB lb;
    la.init(p.lda, p.M - m0, m0, p.ldrs, p.rs_div, has_rs);
    lb.init(p.ldb, p.N - n0, n0, 0, 1, false);
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.load(make_window(aw), kend - kbeg);
        lb.load(make_window(bw), kend - kbeg);
        if constexpr (RS) {
            if (AK) la.load_scale_xk(rsw, kbeg, kend - kbeg);
            else la.load_scale_fk(rsw, kbeg, m0, p.ldrs, p.rs_div, kend - kbeg);
            la.apply_scale();
        }
        la.store(As); lb.store(Bs);
    }
    __syncthreads();
    const int kl = lane >> 5, fl = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = kbeg + (kt + 1) * BK;
            aw += astep; bw += bstep;
            la.load(make_window(aw), kend - k0);
            lb.load(make_window(bw), kend - k0);
            if constexpr (RS) {
                if (AK) la.load_scale_xk(rsw, k0, kend - k0);
                else la.load_scale_fk(rsw, k0, m0, p.ldrs, p.rs_div, kend - k0);
            }
        }
        const __bf16* Ac = As + cur * ASZ + (wm0 + fl) * LDK + 8 * kl;
        const __bf16* Bc = Bs + cur * BSZ + (wn0 + fl) * LDK + 8 * kl;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Ac + i * 32 * LDK + kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bc + j * 32 * LDK + kk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            if constexpr (RS) la.apply_scale();
            la.store(As + (cur ^ 1) * ASZ);
            lb.store(Bs + (cur ^ 1) * BSZ);
        }
        __syncthreads();
    }
    gemm_epilogue<EPI, TM, TN>(p, acc, m0, n0, wm0, wn0, split, kl, fl);
}

// Launch counters per tile instance (tests assert that a shape really ran on the instance it is meant to cover):
// [0] 128x128, [1] 256x128, [2] 256x256, [3] 256x64, [4] 256x32, [5] the small-output TN VALU kernel; +8 for the bf16 kernels; [14] = epilogue variant (EPI) and
// [15] = K-splits of the LAST launch (bench.py reconstructs the kernel symbol of a timed launch).  Not thread-safe (test / bench aid).
static long long g_tile_launches[16];
extern "C" void cham_gemm_launch_counts(long long* out16, int reset) {
    for (int i = 0; i < 16; ++i) { if (out16) out16[i] = g_tile_launches[i]; if (reset) g_tile_launches[i] = 0; }
}

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, int EPI, bool BF16>
static int launch_epi(GemmParams& p, hipStream_t st) {
    g_tile_launches[14] = EPI; g_tile_launches[15] = p.splits;
    size_t smem;
    const void* kern;
    if (BF16) {
        smem = (size_t)2 * (BM + BN) * (BK + 8) * 2;
        // the row-broadcast scale only ever accompanies the scorer's first layer (bias+leaky, NN) and its wgrad (TN; split-K, or
        // plain when the reduction is too short to split: a batch with a handful of valid positions)
        constexpr bool RSI = (EPI == 1 && AK && !BKC) || ((EPI == 6 || EPI == 0) && !AK && !BKC);
        if (p.rs != nullptr && !RSI) return -CHAM_ERR_ARG;
        if (RSI && p.rs != nullptr) {
            auto k = gemm_bf16_kernel<BM, BN, WM, WN, BK, AK, BKC, EPI, RSI>;
            CHAM_SET_DYNAMIC_LDS(k, (int)smem);
            hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, p.splits, 1), dim3(WM * WN * 64), smem, st, p);
            CHAM_CHECK_LAUNCH();
            return CHAM_OK;
        }
        kern = reinterpret_cast<const void*>(gemm_bf16_kernel<BM, BN, WM, WN, BK, AK, BKC, EPI, false>);
    } else {
        using LA = TileLoader<BM, BK, AK, WM * WN * 64>;
        using LB = TileLoader<BN, BK, BKC, WM * WN * 64>;
        smem = (size_t)2 * BK * (LA::LD + LB::LD) * sizeof(float);
        constexpr bool RSI = (EPI == 1 && AK && !BKC) || ((EPI == 6 || EPI == 0) && !AK && !BKC);
        if (p.rs != nullptr && !RSI) return -CHAM_ERR_ARG;
        if (RSI && p.rs != nullptr) {
            auto k = gemm_f32_kernel<BM, BN, WM, WN, BK, AK, BKC, EPI, RSI>;
            CHAM_SET_DYNAMIC_LDS(k, (int)smem);
            hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, p.splits, 1), dim3(WM * WN * 64), smem, st, p);
            CHAM_CHECK_LAUNCH();
            return CHAM_OK;
        }
        kern = reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN, WM, WN, BK, AK, BKC, EPI, false>);
    }
    CHAM_SET_DYNAMIC_LDS(kern, (int)smem);
    dim3 grid(p.nbm * p.nbn, p.splits, 1);
    if (BF16) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, BK, AK, BKC, EPI, false>), grid, dim3(WM * WN * 64), smem, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, AK, BKC, EPI, false>), grid, dim3(WM * WN * 64), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

template <int BM, int BN, int WM, int WN, int BK, bool AK, bool BKC, bool BF16 = false>
static int launch_cfg(GemmParams& p, hipStream_t st) {
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.N + BN - 1) / BN;
    // Epilogues are instantiated for the layouts that use them (a third of the library's code size otherwise):
    //   NN (forward)  plain | bias | bias + leaky | bias + tanh        NT (dgrad)  plain / accumulate | x leaky'(dref) | x tanh'(dref)
    //   TN (wgrad)    plain / accumulate | split-K partial
    // gemm_plan never plans K-splits for NN / NT; the other combinations return -EINVAL.
    constexpr bool NN = AK && !BKC, NT = AK && BKC, TN = !AK && !BKC;
    if (p.splits > 1) {
        if constexpr (TN) {
            p.xcd_split = (p.splits % 8 == 0) ? 1 : 0;
            const int rc = launch_epi<BM, BN, WM, WN, BK, AK, BKC, 6, BF16>(p, st);
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
            if (p.dact == ACT_LEAKY) return launch_epi<BM, BN, WM, WN, BK, AK, BKC, 3, BF16>(p, st);
            if (p.dact == ACT_TANH) return launch_epi<BM, BN, WM, WN, BK, AK, BKC, 4, BF16>(p, st);
        }
        return -CHAM_ERR_ARG;
    }
    if (p.bias || p.act != ACT_NONE) {
        if (!p.bias || p.accumulate) return -CHAM_ERR_ARG;       // every activated layer of the model has a bias
        if constexpr (NN) {
            if (p.act == ACT_LEAKY) return launch_epi<BM, BN, WM, WN, BK, AK, BKC, 1, BF16>(p, st);
            if (p.act == ACT_TANH) return launch_epi<BM, BN, WM, WN, BK, AK, BKC, 2, BF16>(p, st);
            return launch_epi<BM, BN, WM, WN, BK, AK, BKC, 5, BF16>(p, st);
        }
        return -CHAM_ERR_ARG;
    }
    return launch_epi<BM, BN, WM, WN, BK, AK, BKC, 0, BF16>(p, st);
}

static int g_variant = -1;     // -1 = automatic
extern "C" void cham_gemm_set_variant(int v) { g_variant = v; }
template <bool AK, bool BKC>
static int launch_by_shape(GemmParams& p, hipStream_t st) {
    if (p.N > 64) {
        // default: 256x128 tile / 8 waves for large outputs, 128x128 / 4 waves when the grid would otherwise be too
        // small to fill 256 CUs
        // measured on MI355X at the CAR shapes (profiles/r01_gemm_variants.md): NN 256x128 126-130 TFLOP/s; NT (dgrad) and
        // TN (wgrad, long K) prefer the 256x256 tile (123 / 133 TFLOP/s)
        // the largest tile whose grid still gives every one of the 256 CUs a workgroup (a 4 864-row GEMM on 256x256 tiles is 76
        // workgroups: 70 % of the chip idle)
        auto grid = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.splits; };
        int v = ((long)p.M * p.N >= (1L << 20)) ? 2 : 0;
        if (v == 2 && (!AK || BKC) && p.K >= 512 && p.M >= 1024 && p.N >= 512) v = 4;
        if (v == 4 && grid(256, 256) < 256) v = 2;       // ... demoted while the grid does not cover the chip
        if (v == 2 && grid(256, 128) < 256) v = 0;
        if (g_variant >= 0) v = g_variant;
        switch (v) {      // (other tile shapes were measured and dropped: profiles/r01_notes.md items 3 and 9)
            case 2: ++g_tile_launches[1]; return launch_cfg<256, 128, 4, 2, 16, AK, BKC>(p, st);
            case 4: ++g_tile_launches[2]; return launch_cfg<256, 256, 4, 2, 16, AK, BKC>(p, st);
            default: ++g_tile_launches[0]; return launch_cfg<128, 128, 2, 2, 16, AK, BKC>(p, st);
        }
    }
    if (p.N > 32) { ++g_tile_launches[3]; return launch_cfg<256, 64, 4, 1, 16, AK, BKC>(p, st); }
    ++g_tile_launches[4];
    return launch_cfg<256, 32, 4, 1, 16, AK, BKC>(p, st);
}

template <bool AK, bool BKC>
static int launch_by_shape_bf16(GemmParams& p, hipStream_t st) {
    if (p.N > 64) {
        bool big = (long)p.M * p.N >= (1L << 20) && (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * p.splits >= 256;
        if (g_variant >= 0) big = g_variant >= 2;
        if (big) { ++g_tile_launches[8 + 1]; return launch_cfg<256, 128, 4, 2, 32, AK, BKC, true>(p, st); }
        ++g_tile_launches[8 + 0];
        return launch_cfg<128, 128, 2, 2, 32, AK, BKC, true>(p, st);
    }
    if (p.N > 32) { ++g_tile_launches[8 + 3]; return launch_cfg<256, 64, 4, 1, 32, AK, BKC, true>(p, st); }
    ++g_tile_launches[8 + 4];
    return launch_cfg<256, 32, 4, 1, 32, AK, BKC, true>(p, st);
}

static int gemm_dispatch(int precision, const float* A, int lda, int transA, const float* B, int ldb, int transB,
                         float* C, int ldc, int M, int N, int K, const float* bias, int act, const float* dref, int ldr, int dact,
                         const float* rowscale, int ldrs, int rs_div, int accumulate, float* workspace, size_t workspace_bytes,
                         int splits_hint, void* stream);

extern "C" int cham_gemm_f32(const float* A, int lda, int transA, const float* B, int ldb, int transB,
                             float* C, int ldc, int M, int N, int K,
                             const float* bias, int act,
                             const float* dref, int ldr, int dact,
                             const float* rowscale, int ldrs, int rs_div,
                             int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                             void* stream) {
    return gemm_dispatch(0, A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, dref, ldr, dact, rowscale, ldrs, rs_div,
                         accumulate, workspace, workspace_bytes, splits_hint, stream);
}
// same contract, operands rounded to bf16 on the fly, fp32 accumulate / epilogue / output
extern "C" int cham_gemm_bf16(const float* A, int lda, int transA, const float* B, int ldb, int transB,
                              float* C, int ldc, int M, int N, int K,
                              const float* bias, int act,
                              const float* dref, int ldr, int dact,
                              const float* rowscale, int ldrs, int rs_div,
                              int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                              void* stream) {
    return gemm_dispatch(1, A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, dref, ldr, dact, rowscale, ldrs, rs_div,
                         accumulate, workspace, workspace_bytes, splits_hint, stream);
}

// ================================================================================================================================
// SMALL-OUTPUT weight gradients (round 6): C[M, N] = A[K, M]^T B[K, N] with M <= 128 and a long reduction - the scorer's layer-3 / layer-2
// weight gradients (64 x 32 and 128 x 64 over K = B T (1 + N) = 248 064 candidate rows, nar_model.py:452-468 under :718) and the
// user-context share of the PreCAR kernel's (72 x 1024 over K = B T).  On the MFMA tile kernels above these shapes waste three quarters of
// a 256-row tile and - worse - their workgroups need 32-110 KB of LDS: launched beside the W2 weight gradient (one workgroup per CU holding
// 128 KB of the CU's 160 KB) they do not become resident until that kernel's workgroups retire.  Measured in the step
// (profiles/r06_notes.md section 5): 0.75-0.89 ms for the 64 x 32 gradient (95 MB of operands: 25 us at HBM speed) and 0.47-0.54 ms for the
// 72 x 1024 one on the critical tail of the main lane.  Here: v_mfma_f32_32x32x2_f32 fed STRAIGHT FROM GLOBAL MEMORY (see the kernel), four
// waves per workgroup over quarters of its K chunk, their accumulators added in wave order through <= 24 KB of LDS at the end - no staging,
// no barrier in the loop, ~100 VGPRs: a workgroup fits beside anything.  (A first version - fp32 FMAs on the VALU over 16-row LDS stages -
// took 0.52 ms for the 64 x 32 gradient: one 6 KB stage in flight per workgroup is latency-, not bandwidth-bound.)  K-splits write partials
// in the shared layout ([split][M][N]) and the shared fixed-order reduction adds them: deterministic.
template <int TMT, int TNT>
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(GemmParams p) {
    // One wave = a (32 TMT) x (32 TNT) block of C over a quarter of the workgroup's K chunk, straight from global memory: the operands of
    // v_mfma_f32_32x32x2_f32 for a TN product are lane (k = l / 32, m or n = l % 32) - i.e. lanes 0-31 read 128 consecutive bytes of k-row
    // k0, lanes 32-63 of k-row k0 + 1: coalesced dword loads, no LDS staging, no transposition, no barrier in the loop.  U k-pairs of loads
    // are issued back to back (memory-level parallelism: (TMT + TNT) U dwords in flight per lane) before their MFMAs.
    constexpr int U = (TMT * TNT >= 8) ? 4 : 8;
    __shared__ float red[TMT * TNT * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kl = lane >> 5, cl = lane & 31;
    const int n0 = blockIdx.x * (32 * TNT), split = blockIdx.y, m0 = blockIdx.z * (32 * TMT);
    const int kbeg = split * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int quarter = (((kend - kbeg + 3) / 4) + 1) & ~1;          // rows per wave, even
    const int wbeg = kbeg + wave * quarter, wend = min(kend, wbeg + quarter);
    floatx16 acc[TMT][TNT];
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // buffer descriptors over the operands' K rows of this wave (32-bit offsets; a row at or beyond the wave's end, a column beyond the
    // matrix: out-of-range offset -> the load returns 0 without a branch)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (unsigned)min((size_t)p.K * p.lda * 4, (size_t)0x7FFFFFF0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (unsigned)min((size_t)p.K * p.ldb * 4, (size_t)0x7FFFFFF0u), 0x00020000);
    unsigned ca[TMT], cb[TNT];
#pragma unroll
    for (int i = 0; i < TMT; ++i) ca[i] = m0 + 32 * i + cl < p.M ? (unsigned)(m0 + 32 * i + cl) * 4u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < TNT; ++j) cb[j] = n0 + 32 * j + cl < p.N ? (unsigned)(n0 + 32 * j + cl) * 4u : 0x80000000u;
    const unsigned sa = (unsigned)p.lda * 4u, sb = (unsigned)p.ldb * 4u;
    for (int k0 = wbeg; k0 < wend; k0 += 2 * U) {
        float a[U][TMT], b[U][TNT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned row = (unsigned)(k0 + 2 * u + kl);
            const unsigned oob = (int)row < wend ? 0u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < TMT; ++i) a[u][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (row * sa + ca[i]) | oob, 0, 0));
#pragma unroll
            for (int j = 0; j < TNT; ++j) b[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (row * sb + cb[j]) | oob, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
    // waves 1, 2, 3 hand their accumulators to wave 0 one after the other through ONE buffer (<= 16 KB: the workgroup must fit into the
    // 32 KB of LDS that are free beside a plane GEMM's 128 KB): ascending wave order = a fixed summation order
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((i * TNT + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += red[((i * TNT + j) * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave > 0) return;
    // acc[i][j][e] = C(row m0 + 32 i + (e & 3) + 8 (e >> 2) + 4 kl, column n0 + 32 j + cl)
    const bool part = p.splits > 1;
    float* dst = part ? p.partial + (size_t)split * p.M * p.N : p.C;
    const int ld = part ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) {
            const int n = n0 + 32 * j + cl;
            if (n >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kl;
                if (m >= p.M) continue;
                float v = acc[i][j][e];
                if (!part && p.accumulate) v += dst[(size_t)m * ld + n];
                dst[(size_t)m * ld + n] = v;
            }
        }
}

// [5] of cham_gemm_launch_counts: launches of the small-output TN kernel.  CHAM_GEMM_TN_SMALL=0 (environment, read once): off (A/B arm).
static int tn_small_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("CHAM_GEMM_TN_SMALL"); on = (e && e[0] == '0') ? 0 : 1; }
    return on;
}
template <int TMT, int TNT>
static int launch_tn_small(GemmParams& p, hipStream_t st) {
    constexpr int BNt = 32 * TNT;
    ++g_tile_launches[5]; g_tile_launches[14] = p.splits > 1 ? 6 : 0; g_tile_launches[15] = p.splits;
    hipLaunchKernelGGL((gemm_tn_small_kernel<TMT, TNT>), dim3((p.N + BNt - 1) / BNt, p.splits, (p.M + 32 * TMT - 1) / (32 * TMT)), dim3(256), 0, st, p);
    if (p.splits > 1) launch_splitk_reduce(p, st);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

static int gemm_dispatch(int precision, const float* A, int lda, int transA, const float* B, int ldb, int transB,
                             float* C, int ldc, int M, int N, int K,
                             const float* bias, int act,
                             const float* dref, int ldr, int dact,
                             const float* rowscale, int ldrs, int rs_div,
                             int accumulate, float* workspace, size_t workspace_bytes, int splits_hint,
                             void* stream) {
    GemmParams p;
    // small-output weight gradient (see gemm_tn_small_kernel): plain TN, M <= 128, a reduction long enough to matter
    if (precision == 0 && transA && !transB && M <= 128 && K >= 512 && !bias && act == ACT_NONE && !dref && !rowscale && tn_small_enabled()) {
        const int rc0 = gemm_plan(p, A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, dref, ldr, dact, rowscale, ldrs, rs_div,
                                  accumulate, workspace, workspace_bytes, splits_hint);
        if (rc0 != CHAM_OK) return rc0;
        hipStream_t st0 = (hipStream_t)stream;
        // a wave owns at most 2 x 2 tiles of 32 x 32 (64 accumulator registers: the kernel stays under the ~160 VGPRs that are free beside the
        // plane GEMMs' two waves per SIMD); more row tiles go to blockIdx.z (their workgroups re-read the B block: a few MB through L2)
        if ((size_t)K * (lda > ldb ? lda : ldb) * 4 >= 0x7FFFFFF0ull) { /* 32-bit operand offsets: leave it to the tile kernels below */ }
        else {
            const int mt = (M + 31) / 32;
            if (N <= 32) return mt == 1 ? launch_tn_small<1, 1>(p, st0) : launch_tn_small<2, 1>(p, st0);
            return mt == 1 ? launch_tn_small<1, 2>(p, st0) : launch_tn_small<2, 2>(p, st0);
        }
    }
    const int rc = gemm_plan(p, A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, act, dref, ldr, dact, rowscale, ldrs, rs_div,
                             accumulate, workspace, workspace_bytes, splits_hint);
    if (rc != CHAM_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (precision == 1) {
        if (!transA && !transB) return launch_by_shape_bf16<true, false>(p, st);
        if (!transA && transB) return launch_by_shape_bf16<true, true>(p, st);
        if (transA && !transB) return launch_by_shape_bf16<false, false>(p, st);
        return -CHAM_ERR_ARG;
    }
    if (!transA && !transB) return launch_by_shape<true, false>(p, st);
    if (!transA && transB) return launch_by_shape<true, true>(p, st);
    if (transA && !transB) return launch_by_shape<false, false>(p, st);
    return -CHAM_ERR_ARG;                                // TT never occurs on this path
}

