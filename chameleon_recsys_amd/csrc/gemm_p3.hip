// fp32-grade GEMM over operands that ALREADY live in HBM as three bf16 planes (round 3; gfx950, wave64).
//
//   a = a_h + a_m + a_l  (split3 of gemm_shared.h: exact),   C = sum of six plane products, fp32 accumulate
//   (a_h b_h, a_h b_m, a_m b_h, a_h b_l, a_l b_h, a_m b_m - the arithmetic of gemm_x3.hip, tests/test_split3_cpu.py)
//
// gemm_x3.hip splits every fp32 element while it is staged: 3.7 VALU + 0.72 LDS + 0.17 VMEM instructions per MFMA, the matrix pipe
// busy 49 % of the time, and each of the two big matrices of the step (Z1 = PreCAR output, dZ2 = gradient at the CAR tanh) is split
// twice per step by every column tile that reads it.  Here the PRODUCER of such a matrix writes the planes once (k_combine_fwd_cand,
// k_mulpred_bwd: +0.5 GB of stores each) and the K loop contains nothing but
//     MFMAs | fragment ds_reads | LDS-DMA issue (buffer_load_dwordx4 ... lds: global -> LDS without passing through VGPRs)
// i.e. per 16-k step and wave 48 MFMAs, 18 ds_read_b128 (36 ds_read_b64_tr_b16 for TN), 6 DMA instructions, 1 barrier, ~10 SALU.
//
//   NT  C[M,N] = epi(A[M,K] B[N,K]^T)    planes of A / B k-contiguous: CAR forward (B = planes of W2^T), CAR dgrad (B = planes of W2)
//   TN  C[M,N] = A[K,M]^T B[K,N]         planes stored with the FREE index contiguous (W2 wgrad: Z1^T dZ2, K = candidate rows); split-K
//
// Tile 256 x 256 x 16, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 4 x 2 MFMA tiles of 32x32x16.  A stage = 6 slabs of 8 KB
// (A_h, A_m, A_l, B_h, B_m, B_l); ring of three stages = 144 KB of LDS, one workgroup per CU.
//   * NT slab = [256 rows][16 k] (32 B per row); 16-byte halves of a row swapped on rows with bit 3 set -> the four 16-lane groups of
//     a ds_read_b128 fragment read (MI355X_MICROARCH.md, LDS) each cover the 64 banks exactly once.
//   * TN slab = [16 k][256 m] (512 B per k); the 64-byte groups of k-row k are XOR-ed with k & 3 -> the 4 k-rows x 64 B a half-wave
//     touches in one ds_read_b64_tr_b16 lie on four different bank quarters.
//   LDS-DMA writes wave-uniform base + lane * 16, so both swizzles are applied to the SOURCE address of the lane that owns the
//   destination piece and again in the fragment read (cdna_hip_programming.md rule 21).
// Pipeline (one s_barrier per 16-k step; i = step, slot = i % 3):
//   top      : DMA of stage i + 2 into slot (i + 2) % 3 (its last reader was step i - 1, P0; separated by barrier i - 1)
//   P0..P2   : AL x BH | AM x BH | AH x BH;  P0 also reads the LATE fragments of stage i (A_h, B_m) from slot i % 3
//   mid      : s_waitcnt vmcnt(6) - this wave's DMA of stage i + 1 has landed, stage i + 2's six stay in flight - then s_barrier
//   P3..P5   : AH x BL | AM x BM | AH x BM;  the EARLY fragments of stage i + 1 (B_h, A_l | B_l | A_m) are read into the registers
//              whose last use has passed - no second fragment set: 128 accumulator + 72 fragment registers
//   A DMA has 1.5 steps (~4 600 cycles at two waves per SIMD) to land.  Ragged rows / K tails rely on the buffer descriptor's range
//   check (one descriptor per plane: out-of-range pieces arrive as zeros).
// Epilogues: plain, + bias -> tanh (CAR forward), x leaky'(h plane of the saved activation) (CAR dgrad), split-K partial (wgrad).
#include "gemm_shared.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct P3Params {
    const __bf16* A; const __bf16* B;          // plane 0 of each operand
    long long a_ps, b_ps;                      // plane stride, elements
    int lda, ldb;                              // row stride, elements
    float* C; int ldc;
    int M, N, K;
    const float* bias;
    const __bf16* dref; int ldr;               // h plane of the saved activation (dgrad)
    int kchunk, splits; float* partial;
    int nbm, nbn, xcd_split, accumulate;
};

#define P3_SLAB 8192
#define P3_STAGE (6 * P3_SLAB)
#define P3_RING 3

__device__ __forceinline__ u32x4 p3_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r.x = (unsigned)a; r.y = (unsigned)(a >> 32) & 0xFFFFu; r.z = bytes; r.w = 0x00020000u;
    return r;
}

// six LDS-DMA requests of one stage: 16 bytes per lane, LDS destination = M0 + lane * 16 (wave-uniform), source = descriptor base
// + voffset (per lane; the K advance is part of it, so that the descriptor's range check - which ignores an SGPR offset - sees it).
// M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void p3_dma_stage(unsigned lds0, unsigned va, unsigned vb, const u32x4& ra0, const u32x4& ra1, const u32x4& ra2,
                                             const u32x4& rb0, const u32x4& rb1, const u32x4& rb2) {
    unsigned keep;
    const unsigned l1 = lds0 + P3_SLAB, l2 = lds0 + 2 * P3_SLAB, l3 = lds0 + 3 * P3_SLAB, l4 = lds0 + 4 * P3_SLAB, l5 = lds0 + 5 * P3_SLAB;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %9, 0 offen lds\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %10, 0 offen lds\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %11, 0 offen lds\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %12, 0 offen lds\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %13, 0 offen lds\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %14, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0), "s"(l1), "s"(l2), "s"(l3), "s"(l4), "s"(l5), "v"(va), "v"(vb), "s"(ra0), "s"(ra1), "s"(ra2), "s"(rb0), "s"(rb1),
          "s"(rb2)
        : "memory");
}

template <bool TN>
__device__ __forceinline__ bf16x8 p3_frag(const unsigned char* __restrict__ s) {
    if constexpr (!TN) {
        return *reinterpret_cast<const bf16x8*>(s);
    } else {
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(s));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(s + 4 * 512));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8, v);
    }
}

// LDS-only barrier (gemm_x3.hip): builtins so that the wait-count pass sees the drain; vmcnt is handled by hand (the DMA requests are
// invisible to the compiler).
__device__ __forceinline__ void p3_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    __builtin_amdgcn_sched_barrier(0);
}

// epilogue of the plane kernels: acc[i][j][e] = element (row wm0 + 32 i + (e & 3) + 8 (e >> 2) + 4 kl, column wn0 + 32 j + fl) of the tile at (m0, n0)
template <int EPI, int TM, int TNN>
__device__ __forceinline__ void p3_epilogue(const P3Params& p, floatx16 (&acc)[TM][TNN], int m0, int n0, int wm0, int wn0, int split, int lane) {
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int kl = lane_e >> 5, fl = lane_e & 31;
    if constexpr (EPI == 3) {
        // dgrad: x leaky'(saved activation), the activation's sign read from its h plane (bf16 rounding keeps sign and zero)
        const int limM = p.M - m0, limN = p.N - n0;
        const __amdgpu_buffer_rsrc_t cw = make_window(p.C + (size_t)m0 * p.ldc + n0);
        const __amdgpu_buffer_rsrc_t dw = make_window(p.dref + (size_t)m0 * p.ldr + n0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
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
                    y[e] = __builtin_amdgcn_raw_buffer_load_b16(dw, ok ? ((unsigned)row * (unsigned)p.ldr + (unsigned)col) * 2u : OOB_OFF, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float yv = __uint_as_float((unsigned)y[e] << 16);
                    const float v = acc[i][j][e] * (yv > 0.f ? 1.f : 0.2f);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), cw, offs[e], 0, 0);
                }
            }
    } else {
        GemmParams g;
        g.A = nullptr; g.B = nullptr; g.C = p.C; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = 0; g.ldb = 0; g.ldc = p.ldc;
        g.bias = p.bias; g.act = ACT_NONE; g.dref = nullptr; g.ldr = 0; g.dact = ACT_NONE; g.rs = nullptr; g.ldrs = 0; g.rs_div = 1;
        g.accumulate = p.accumulate; g.kchunk = p.kchunk; g.splits = p.splits; g.partial = p.partial; g.nbm = p.nbm; g.nbn = p.nbn; g.xcd_split = p.xcd_split;
        gemm_epilogue<EPI, TM, TNN>(g, acc, m0, n0, wm0, wn0, split, kl, fl);
    }
}

// EPI: 0 plain, 2 bias + tanh, 3 x leaky'(dref h plane), 5 bias, 6 split-K partial
// (Round 3's A/B arms of this kernel - a staggered pipeline with one DMA request per MFMA pass, 2-5 % slower; the NT form on 256x128
// tiles with two workgroups per CU, 15 % slower; the last round of tiles as its own split-K launch, neutral - were measured and removed:
// profiles/r03_notes.md sections 1 and 5.)
template <bool TN, int EPI>
__global__ __launch_bounds__(512) void gemm_p3_kernel(P3Params p) {
    constexpr int BM = 256, BN = 256, BK = 16, TM = 4, TNN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char p3_smem[];

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
    u32x4 ra[3], rb[3];
    unsigned va, vb, stepa, stepb;
    if constexpr (!TN) {
        // (the window starts kbeg elements into the first row: it must end that much earlier, or the requests past the reduction range
        // of the LAST K-split - issued, never consumed - would leave the operand's allocation)
        const size_t aall = (size_t)max(p.M - m0, 0) * p.lda * 2, ball = (size_t)max(p.N - n0, 0) * p.ldb * 2, koff = (size_t)kbeg * 2;
        const size_t abytes = aall > koff ? aall - koff : 0, bbytes = ball > koff ? ball - koff : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ra[q] = p3_rsrc(p.A + q * p.a_ps + (size_t)m0 * p.lda + kbeg, (unsigned)min(abytes, (size_t)0xFFFFFFF0u));
            rb[q] = p3_rsrc(p.B + q * p.b_ps + (size_t)n0 * p.ldb + kbeg, (unsigned)min(bbytes, (size_t)0xFFFFFFF0u));
        }
        // lane l of wave w fills LDS piece (row 32 w + l / 2, half l & 1); that piece holds source half (l & 1) ^ bit 3 of the row
        const int row = 32 * wave + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
        va = ((unsigned)row * (unsigned)p.lda + 8u * half) * 2u;
        vb = ((unsigned)row * (unsigned)p.ldb + 8u * half) * 2u;
        stepa = stepb = BK * 2;
    } else {
        const size_t abytes = (size_t)max(kend - kbeg, 0) * p.lda * 2, bbytes = (size_t)max(kend - kbeg, 0) * p.ldb * 2;
#pragma unroll
        for (int q = 0; q < 3; ++q) {      // (M, N multiples of 256: the m / n extent of a tile never leaves its k-row)
            ra[q] = p3_rsrc(p.A + q * p.a_ps + (size_t)kbeg * p.lda + m0, (unsigned)min(abytes - (size_t)m0 * 2, (size_t)0xFFFFFFF0u));
            rb[q] = p3_rsrc(p.B + q * p.b_ps + (size_t)kbeg * p.ldb + n0, (unsigned)min(bbytes - (size_t)n0 * 2, (size_t)0xFFFFFFF0u));
        }
        // lane l of wave w fills LDS piece (k-row 2 w + l / 32, piece l & 31); its 64-byte group index is XOR-ed with k & 3
        const int k = 2 * wave + (lane >> 5), jp = lane & 31, j = ((((jp >> 2) ^ (k & 3)) << 2) | (jp & 3));
        va = ((unsigned)k * (unsigned)p.lda + 8u * j) * 2u;
        vb = ((unsigned)k * (unsigned)p.ldb + 8u * j) * 2u;
        stepa = (unsigned)BK * (unsigned)p.lda * 2u; stepb = (unsigned)BK * (unsigned)p.ldb * 2u;
    }

    // ---- per-lane fragment offsets inside a slab
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
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = (unsigned)kq * 512u + (unsigned)((((wm0 >> 5) + i) ^ (kq & 3))) * 64u + within;
#pragma unroll
        for (int j = 0; j < TNN; ++j) fb[j] = (unsigned)kq * 512u + (unsigned)((((wn0 >> 5) + j) ^ (kq & 3))) * 64u + within;
    }

    floatx16 acc[TM][TNN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_u8*)p3_smem;
    const unsigned wave_off = (unsigned)wave * 1024u;
    bf16x8 AH[TM], AM[TM], AL[TM], BH[TNN], BMf[TNN], BL[TNN];

    if (nk > 0) {          // all six requests at the top of a step, both wave groups in lockstep
        p3_dma_stage(lds_base + wave_off, va, vb, ra[0], ra[1], ra[2], rb[0], rb[1], rb[2]);
        va += stepa; vb += stepb;
        p3_dma_stage(lds_base + P3_STAGE + wave_off, va, vb, ra[0], ra[1], ra[2], rb[0], rb[1], rb[2]);
        va += stepa; vb += stepb;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        p3_barrier();
        {   // early fragments of stage 0
            const unsigned char* S = p3_smem;
#pragma unroll
            for (int j = 0; j < TNN; ++j) BH[j] = p3_frag<TN>(S + 3 * P3_SLAB + fb[j]);
#pragma unroll
            for (int i = 0; i < TM; ++i) AL[i] = p3_frag<TN>(S + 2 * P3_SLAB + fa[i]);
#pragma unroll
            for (int j = 0; j < TNN; ++j) BL[j] = p3_frag<TN>(S + 5 * P3_SLAB + fb[j]);
#pragma unroll
            for (int i = 0; i < TM; ++i) AM[i] = p3_frag<TN>(S + 1 * P3_SLAB + fa[i]);
        }
        int cur = 0;                                    // slot of stage i
        for (int i = 0; i < nk; ++i) {
            const int nxt = cur == P3_RING - 1 ? 0 : cur + 1, nx2 = nxt == P3_RING - 1 ? 0 : nxt + 1;
            const unsigned char* Sc = p3_smem + cur * P3_STAGE;
            const unsigned char* Sn = p3_smem + nxt * P3_STAGE;
            __builtin_amdgcn_sched_barrier(0);
            // top: stage i + 2 (a stage beyond the reduction range is requested all the same - it is never consumed, and the wait
            // counts stay the same in every step)
            p3_dma_stage(lds_base + (unsigned)nx2 * P3_STAGE + wave_off, va, vb, ra[0], ra[1], ra[2], rb[0], rb[1], rb[2]);
            va += stepa; vb += stepb;
            __builtin_amdgcn_sched_barrier(0);
            // P0: A_l x B_h; late fragments of this stage
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AH[ii] = p3_frag<TN>(Sc + 0 * P3_SLAB + fa[ii]);
#pragma unroll
            for (int j = 0; j < TNN; ++j) BMf[j] = p3_frag<TN>(Sc + 4 * P3_SLAB + fb[j]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL[ii], BH[j], acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // P1: A_m x B_h
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM[ii], BH[j], acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // P2: A_h x B_h
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[ii], BH[j], acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // mid: this wave's requests for stage i + 1 have landed (the six of stage i + 2 stay in flight); after the barrier every
            // wave's have
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            p3_barrier();
            // P3: A_h x B_l; early fragments of stage i + 1 into the registers that are dead
#pragma unroll
            for (int j = 0; j < TNN; ++j) BH[j] = p3_frag<TN>(Sn + 3 * P3_SLAB + fb[j]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AL[ii] = p3_frag<TN>(Sn + 2 * P3_SLAB + fa[ii]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[ii], BL[j], acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // P4: A_m x B_m
#pragma unroll
            for (int j = 0; j < TNN; ++j) BL[j] = p3_frag<TN>(Sn + 5 * P3_SLAB + fb[j]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM[ii], BMf[j], acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // P5: A_h x B_m
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) AM[ii] = p3_frag<TN>(Sn + 1 * P3_SLAB + fa[ii]);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[ii], BMf[j], acc[ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no request may outlive the workgroup's LDS allocation
    }

    p3_epilogue<EPI, TM, TNN>(p, acc, m0, n0, wm0, wn0, split, lane);
}

// ================================================================================================================================
// The bf16 configuration's CAR GEMMs on the same core: ONE bf16 plane per operand (the matrices the bf16 configuration keeps in HBM),
// the three A slabs / three B slabs of a stage holding three CONSECUTIVE 16-k chunks, so a stage advances K by 48 and its three
// passes are the "diagonal" products chunk q of A x chunk q of B: per stage and wave 24 MFMAs, 18 ds_read_b128, 6 DMA requests, one
// barrier - nothing else in the K loop (gemm_b16.hip stages through registers: 0.26-0.32 of the bf16 matrix peak).
//   step i:  top   DMA of stage i + 2
//            P0    A0 x B0 | fragment reads of chunk 1 (this stage)
//            P1    A1 x B1 | fragment reads of chunk 2
//            mid   s_waitcnt vmcnt(6) (stage i + 1 landed) + s_barrier
//            P2    A2 x B2 | fragment reads of chunk 0 of stage i + 1
//   K tail (K % 48 != 0): NT operands are k-contiguous, a chunk past K would read the row's neighbours - the requests of such a
//   chunk go out with a zero-sized descriptor (every lane out of range: zeros land in LDS); TN: the k-rows past the split's end are
//   beyond the descriptor's range anyway.
//   NT runs the MFMAs with swapped operands (accumulator = C^T: a lane owns four consecutive output columns of its row), so the
//   bf16 epilogues load / store 8 bytes per access (as gemm_b16.hip); TN keeps the shared split-K partial epilogue.
// EPI: NT 12 = + bias -> tanh -> bf16, 13 = x leaky'(saved bf16 activation) -> bf16, 10 = plain -> bf16;  TN 6 = split-K partial, 0 = fp32.
__device__ __forceinline__ float b1_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float b1_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ unsigned b1_pack(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;           // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

// The NT epilogue of the bf16-resident kernels (accumulators of SWAPPED products: acc[i][j][e] = row m0 + wm0 + 32 i + fl, column n0 + wn0 +
// 32 j + 8 (e >> 2) + 4 kl + (e & 3)); shared by gemm_b1_kernel<false, .> and gemm_b1w_kernel.  Needs the kernel's 128 KB of dynamic LDS.
template <int EPI>
__device__ __forceinline__ void b1_nt_epilogue(const P3Params& p, floatx16 (&acc)[4][2], int m0, int n0, int wm0, int wn0, int lane, int wave) {
    constexpr int TM = 4, TNN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char p3_smem[];
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int kl = lane_e >> 5, fl = lane_e & 31;
    {
        // acc[i][j][e]: row m = wm0 + 32 i + fl, column n = wn0 + 32 j + 8 (e >> 2) + 4 kl + (e & 3)
        const int limM = p.M - m0, limN = p.N - n0;
        __bf16* Cb = reinterpret_cast<__bf16*>(p.C);
        const __amdgpu_buffer_rsrc_t cw = make_window(Cb + (size_t)m0 * p.ldc + n0);
        const __amdgpu_buffer_rsrc_t dw = make_window(EPI == 13 ? (const void*)(p.dref + (size_t)m0 * p.ldr + n0) : (const void*)Cb);
        const __amdgpu_buffer_rsrc_t bw = make_window(EPI == 12 ? (const void*)(p.bias + n0) : (const void*)Cb);
        // ---- interior wave tiles: the epilogue goes THROUGH LDS (round 5).  With swapped operands a lane owns a ROW and four consecutive
        // columns: its 8-byte stores (and the dgrad's 8-byte loads of the saved activation) hit 32 different rows per instruction - 32
        // sixteen-byte fragments of 32 different lines.  Measured with the stores compiled out: 0.548 ms against 0.713 ms for the plain
        // NT product, and another 0.15 ms for the dgrad's loads (profiles/r05_notes.md section 6): the NT forms' gap to the TN form is this
        // access pattern.  Here a wave packs its 128 x 64 bf16 outputs into 16 KB of the (now free) ring - row pitch 128 bytes, 16-byte piece
        // q of row r at slot q ^ ((r >> 1) & 7), the two 8-byte halves of a piece swapped on rows with bit 4 set (ds_write_b64 of 32 rows x one
        // column group: 32 different bank pairs) - reads it back 16 bytes per lane (8 lanes = one whole 128-byte row segment; two rows per 16
        // lanes = all 64 banks) and stores / loads global memory in whole lines: 16 b128 accesses per lane instead of 32 b64.
        // The dgrad (EPI 13) works in two halves of 64 rows: [saved activation | outputs] share the wave's 16 KB.
        p3_barrier();                 // every wave is past its last fragment read; no DMA in flight (vmcnt(0) above): the ring is free
        if (limM >= wm0 + 128 && limN >= wn0 + 64) {          // wave-uniform
            unsigned char* Rg = p3_smem + (unsigned)wave * 16384u;
            const unsigned lr = (unsigned)(lane >> 3), ls = (unsigned)(lane & 7);
            auto slot_off = [](unsigned r, unsigned pp, unsigned half) -> unsigned {
                return r * 128u + (((pp ^ (r >> 1)) & 7u) << 4) + (((half ^ (r >> 4)) & 1u) << 3);
            };
            if constexpr (EPI != 13) {
#pragma unroll
                for (int j = 0; j < TNN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u32x4 bv;
                        if constexpr (EPI == 12) bv = __builtin_amdgcn_raw_buffer_load_b128(bw, (unsigned)(wn0 + j * 32 + 8 * q + 4 * kl) * 4u, 0, 0);
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                            if constexpr (EPI == 12) {
                                v[0] = cham_tanhf(v[0] + __uint_as_float(bv.x)); v[1] = cham_tanhf(v[1] + __uint_as_float(bv.y));
                                v[2] = cham_tanhf(v[2] + __uint_as_float(bv.z)); v[3] = cham_tanhf(v[3] + __uint_as_float(bv.w));
                            }
                            u32x2 w;
                            w.x = b1_pack(v[0], v[1]); w.y = b1_pack(v[2], v[3]);
                            *reinterpret_cast<u32x2*>(Rg + slot_off((unsigned)(32 * i + fl), (unsigned)(4 * j + q), (unsigned)kl)) = w;
                        }
                    }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const unsigned r = 8u * t + lr;
                    u32x4 v = *reinterpret_cast<const u32x4*>(Rg + r * 128u + ls * 16u);
                    if ((r >> 4) & 1u) { const unsigned a = v.x, b = v.y; v.x = v.z; v.y = v.w; v.z = a; v.w = b; }
                    const unsigned pp = (ls ^ (r >> 1)) & 7u;
                    __builtin_amdgcn_raw_buffer_store_b128(v, cw, ((unsigned)(wm0 + r) * (unsigned)p.ldc + (unsigned)(wn0 + 8 * pp)) * 2u, 0, 0);
                }
            } else {
                unsigned char* Dg = Rg;                     // saved activation, 64 rows
                unsigned char* Og = Rg + 8192;              // outputs, 64 rows
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const unsigned r = 8u * t + lr, pp = (ls ^ (r >> 1)) & 7u;
                        u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(dw, ((unsigned)(wm0 + 64 * h + r) * (unsigned)p.ldr + (unsigned)(wn0 + 8 * pp)) * 2u, 0, 0);
                        if ((r >> 4) & 1u) { const unsigned a = y.x, b = y.y; y.x = y.z; y.y = y.w; y.z = a; y.w = b; }
                        *reinterpret_cast<u32x4*>(Dg + r * 128u + ls * 16u) = y;
                    }
#pragma unroll
                    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
                        for (int j = 0; j < TNN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int i = 2 * h + ih;
                                const unsigned off = slot_off((unsigned)(32 * ih + fl), (unsigned)(4 * j + q), (unsigned)kl);
                                const u32x2 y = *reinterpret_cast<const u32x2*>(Dg + off);
                                float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                                v[0] *= b1_lo(y.x) > 0.f ? 1.f : 0.2f; v[1] *= b1_hi(y.x) > 0.f ? 1.f : 0.2f;
                                v[2] *= b1_lo(y.y) > 0.f ? 1.f : 0.2f; v[3] *= b1_hi(y.y) > 0.f ? 1.f : 0.2f;
                                u32x2 w;
                                w.x = b1_pack(v[0], v[1]); w.y = b1_pack(v[2], v[3]);
                                *reinterpret_cast<u32x2*>(Og + off) = w;
                            }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const unsigned r = 8u * t + lr;
                        u32x4 v = *reinterpret_cast<const u32x4*>(Og + r * 128u + ls * 16u);
                        if ((r >> 4) & 1u) { const unsigned a = v.x, b = v.y; v.x = v.z; v.y = v.w; v.z = a; v.w = b; }
                        const unsigned pp = (ls ^ (r >> 1)) & 7u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, cw, ((unsigned)(wm0 + 64 * h + r) * (unsigned)p.ldc + (unsigned)(wn0 + 8 * pp)) * 2u, 0, 0);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = wm0 + i * 32 + fl;
            const bool mok = m < limM;
#pragma unroll
            for (int j = 0; j < TNN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = wn0 + j * 32 + 8 * q + 4 * kl;
                    const bool ok = mok && n < limN;              // N % 4 == 0: a group of 4 columns is in or out as a whole
                    float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    if constexpr (EPI == 12) {
                        const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(bw, ok ? (unsigned)n * 4u : OOB_OFF, 0, 0);
                        v[0] = cham_tanhf(v[0] + __uint_as_float(bv.x)); v[1] = cham_tanhf(v[1] + __uint_as_float(bv.y));
                        v[2] = cham_tanhf(v[2] + __uint_as_float(bv.z)); v[3] = cham_tanhf(v[3] + __uint_as_float(bv.w));
                    }
                    if constexpr (EPI == 13) {
                        const u32x2 y = __builtin_amdgcn_raw_buffer_load_b64(dw, ok ? ((unsigned)m * (unsigned)p.ldr + (unsigned)n) * 2u : OOB_OFF, 0, 0);
                        v[0] *= b1_lo(y.x) > 0.f ? 1.f : 0.2f; v[1] *= b1_hi(y.x) > 0.f ? 1.f : 0.2f;
                        v[2] *= b1_lo(y.y) > 0.f ? 1.f : 0.2f; v[3] *= b1_hi(y.y) > 0.f ? 1.f : 0.2f;
                    }
                    u32x2 w;
                    w.x = b1_pack(v[0], v[1]); w.y = b1_pack(v[2], v[3]);
                    __builtin_amdgcn_raw_buffer_store_b64(w, cw, ok ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 2u : OOB_OFF, 0, 0);
                }
        }
    }
}

template <bool TN, int EPI>
__global__ __launch_bounds__(512) void gemm_b1_kernel(P3Params p) {
    constexpr int BM = 256, BN = 256, KS = 48, TM = 4, TNN = 2;
    constexpr bool SWAP = !TN;
    extern __shared__ __attribute__((aligned(1024))) unsigned char p3_smem[];

    const int nwg = p.nbm * p.nbn;
    int tile_m, tile_n, split;
    if (p.xcd_split) {
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
    const int klen = max(kend - kbeg, 0);
    const int nk = (klen + KS - 1) / KS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

    // ---- one descriptor per 16-k chunk of a stage
    u32x4 ra[3], rb[3];
    unsigned va, vb, stepa, stepb;
    if constexpr (!TN) {
        const size_t abytes = (size_t)max(p.M - m0, 0) * p.lda * 2, bbytes = (size_t)max(p.N - n0, 0) * p.ldb * 2;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ra[q] = p3_rsrc(p.A + (size_t)m0 * p.lda + kbeg + 16 * q, (unsigned)min(abytes > 32u * q ? abytes - 32u * q : (size_t)0, (size_t)0xFFFFFFF0u));
            rb[q] = p3_rsrc(p.B + (size_t)n0 * p.ldb + kbeg + 16 * q, (unsigned)min(bbytes > 32u * q ? bbytes - 32u * q : (size_t)0, (size_t)0xFFFFFFF0u));
        }
        const int row = 32 * wave + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
        va = ((unsigned)row * (unsigned)p.lda + 8u * half) * 2u;
        vb = ((unsigned)row * (unsigned)p.ldb + 8u * half) * 2u;
        stepa = stepb = KS * 2;
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const size_t rows = (size_t)max(klen - 16 * q, 0);
            const size_t abytes = rows * p.lda * 2, bbytes = rows * p.ldb * 2;
            ra[q] = p3_rsrc(p.A + (size_t)(kbeg + 16 * q) * p.lda + m0, (unsigned)min(abytes > (size_t)m0 * 2 ? abytes - (size_t)m0 * 2 : (size_t)0, (size_t)0xFFFFFFF0u));
            rb[q] = p3_rsrc(p.B + (size_t)(kbeg + 16 * q) * p.ldb + n0, (unsigned)min(bbytes > (size_t)n0 * 2 ? bbytes - (size_t)n0 * 2 : (size_t)0, (size_t)0xFFFFFFF0u));
        }
        const int k = 2 * wave + (lane >> 5), jp = lane & 31, j = ((((jp >> 2) ^ (k & 3)) << 2) | (jp & 3));
        va = ((unsigned)k * (unsigned)p.lda + 8u * j) * 2u;
        vb = ((unsigned)k * (unsigned)p.ldb + 8u * j) * 2u;
        stepa = (unsigned)KS * (unsigned)p.lda * 2u; stepb = (unsigned)KS * (unsigned)p.ldb * 2u;
    }

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
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = (unsigned)kq * 512u + (unsigned)((((wm0 >> 5) + i) ^ (kq & 3))) * 64u + within;
#pragma unroll
        for (int j = 0; j < TNN; ++j) fb[j] = (unsigned)kq * 512u + (unsigned)((((wn0 >> 5) + j) ^ (kq & 3))) * 64u + within;
    }

    floatx16 acc[TM][TNN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_u8*)p3_smem;
    const unsigned wave_off = (unsigned)wave * 1024u;
    bf16x8 A0[TM], A1[TM], A2[TM], B0[TNN], B1[TNN], B2[TNN];

    // requests of stage s: NT chunks that start at or past the end of the reduction range get a zero-sized descriptor
    auto dma = [&](int s, unsigned slot) {
        u32x4 a0 = ra[0], a1 = ra[1], a2 = ra[2], b0 = rb[0], b1 = rb[1], b2 = rb[2];
        if constexpr (!TN) {
            const int k0 = s * KS;
            const bool v0 = k0 < klen, v1 = k0 + 16 < klen, v2 = k0 + 32 < klen;
            a0.z = v0 ? a0.z : 0u; b0.z = v0 ? b0.z : 0u;
            a1.z = v1 ? a1.z : 0u; b1.z = v1 ? b1.z : 0u;
            a2.z = v2 ? a2.z : 0u; b2.z = v2 ? b2.z : 0u;
        }
        p3_dma_stage(lds_base + slot * P3_STAGE + wave_off, va, vb, a0, a1, a2, b0, b1, b2);
        va += stepa; vb += stepb;
    };
    // the pass's fragment reads are issued BEFORE its MFMAs (pinned: left to itself the scheduler sinks them to the end of the pass and
    // the next pass starts by waiting for them)
    auto mma = [&](const bf16x8 (&X)[TM], const bf16x8 (&Y)[TNN]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
            for (int j = 0; j < TNN; ++j) {
                if constexpr (SWAP) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Y[j], X[ii], acc[ii][j], 0, 0, 0);
                else acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[ii], Y[j], acc[ii][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {
        dma(0, 0u);
        dma(1, 1u);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        p3_barrier();
#pragma unroll
        for (int ii = 0; ii < TM; ++ii) A0[ii] = p3_frag<TN>(p3_smem + 0 * P3_SLAB + fa[ii]);
#pragma unroll
        for (int j = 0; j < TNN; ++j) B0[j] = p3_frag<TN>(p3_smem + 3 * P3_SLAB + fb[j]);
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            const int nxt = cur == P3_RING - 1 ? 0 : cur + 1, nx2 = nxt == P3_RING - 1 ? 0 : nxt + 1;
            const unsigned char* Sc = p3_smem + cur * P3_STAGE;
            const unsigned char* Sn = p3_smem + nxt * P3_STAGE;
            __builtin_amdgcn_sched_barrier(0);
            dma(i + 2, (unsigned)nx2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) A1[ii] = p3_frag<TN>(Sc + 1 * P3_SLAB + fa[ii]);
#pragma unroll
            for (int j = 0; j < TNN; ++j) B1[j] = p3_frag<TN>(Sc + 4 * P3_SLAB + fb[j]);
            mma(A0, B0);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) A2[ii] = p3_frag<TN>(Sc + 2 * P3_SLAB + fa[ii]);
#pragma unroll
            for (int j = 0; j < TNN; ++j) B2[j] = p3_frag<TN>(Sc + 5 * P3_SLAB + fb[j]);
            mma(A1, B1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            p3_barrier();
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) A0[ii] = p3_frag<TN>(Sn + 0 * P3_SLAB + fa[ii]);
#pragma unroll
            for (int j = 0; j < TNN; ++j) B0[j] = p3_frag<TN>(Sn + 3 * P3_SLAB + fb[j]);
            mma(A2, B2);
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no request may outlive the workgroup's LDS allocation
    }

    if constexpr (SWAP) {
        b1_nt_epilogue<EPI>(p, acc, m0, n0, wm0, wn0, lane, wave);
    } else {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int kl = lane_e >> 5, fl = lane_e & 31;
        GemmParams g;
        g.A = nullptr; g.B = nullptr; g.C = p.C; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = 0; g.ldb = 0; g.ldc = p.ldc;
        g.bias = nullptr; g.act = ACT_NONE; g.dref = nullptr; g.ldr = 0; g.dact = ACT_NONE; g.rs = nullptr; g.ldrs = 0; g.rs_div = 1;
        g.accumulate = p.accumulate; g.kchunk = p.kchunk; g.splits = p.splits; g.partial = p.partial; g.nbm = p.nbm; g.nbn = p.nbn; g.xcd_split = p.xcd_split;
        gemm_epilogue<EPI, TM, TNN>(g, acc, m0, n0, wm0, wn0, split, kl, fl);
    }
}

// ================================================================================================================================
// NT with 64-BYTE SOURCE PIECES for the bf16-resident operands (round 6; what gemm_h2w_kernel is to gemm_h2_kernel).  gemm_b1_kernel's NT
// requests fetch 32 rows x 32 bytes - a quarter of each 128-byte line, the next quarter a stage later when the vector L1 has long dropped
// the line: per workgroup and 64 k, 4 096 cycles of L1 line fills (64 B / clk) against 2 048 cycles of MFMA work - which is why the NT
// forms took 0.84 / 0.79 ms where the TN form of the same FLOPs takes 0.46 ms (profiles/r06_bf16_kernel_stats.csv).  Here a request
// fetches 16 rows x 64 bytes; the ring holds TWO buffers of 64 k = four 16-k chunks:
//   buffer = [A k 0..31 | A k 32..63 | B k 0..31 | B k 32..63], each slab [256 rows][64 bytes] = 16 KB (gemm_h2w_kernel's slab: LDS piece q of
//   row r holds source piece q ^ ((r >> 2) & 3); the k-half of a 32-k slab: fragment offset ^ 32)
//   per buffer D_j and wave: top  the B slabs of D_{j+1} (4 requests: the L2-resident operand)
//                            chunks 0, 1, 2 - fragment reads of the next chunk, then 8 MFMAs
//                            s_waitcnt vmcnt(0) + barrier (D_{j+1} has landed, every wave is past its last fragment read of D_j)
//                            the A slabs of D_{j+2} into D_j's buffer (4 requests: the HBM-streamed operand, a whole buffer time to land)
//                            chunk 3's MFMAs | fragment reads of chunk 0 of D_{j+1}
// One barrier per 64 k, 32 MFMAs, 8 DMA requests.  K % 64 == 0 (the caller keeps gemm_b1_kernel otherwise).  Same products in another
// grouping of the K loop: NOT bit-identical to gemm_b1_kernel (48-k stages: another summation order) - fp32 accumulation either way.
#define B1W_SLAB 16384
#define B1W_BUF (4 * B1W_SLAB)

// four requests of one operand: the two 16-row halves of a wave's 32 rows, in the k 0..31 slab (descriptor r0) and the k 32..63 slab (r1 =
// the same rows 64 bytes further)
__device__ __forceinline__ void b1w_dma4(unsigned l0, unsigned l1, unsigned v0, unsigned v1, const u32x4& r0, const u32x4& r1) {
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

// EPI: 12 = + bias -> tanh -> bf16, 13 = x leaky'(saved bf16 activation) -> bf16, 10 = plain -> bf16
template <int EPI>
__global__ __launch_bounds__(512) void gemm_b1w_kernel(P3Params p) {
    constexpr int BM = 256, BN = 256, TM = 4, TNN = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char p3_smem[];
    const int nwg = p.nbm * p.nbn;
    const int id = blockIdx.x;
    const int q8 = nwg / 8, rr = nwg % 8, xcd = id % 8;       // XCD-aware bijective swizzle: the column tiles of an A panel share an L2
    const int swz = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + id / 8;
    const int tile_m = swz / p.nbn, tile_n = swz % p.nbn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nd = p.K >> 6;                                   // 64-k buffers
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;

    // descriptors: [0] = the rows from k = 0, [1] = the same rows 32 k (64 bytes) further; rows beyond the operand arrive as zeros
    u32x4 ra[2], rb[2];
    const size_t aall = (size_t)max(p.M - m0, 0) * p.lda * 2, ball = (size_t)max(p.N - n0, 0) * p.ldb * 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ra[q] = p3_rsrc(p.A + (size_t)m0 * p.lda + 32 * q, (unsigned)min(aall > 64u * q ? aall - 64u * q : (size_t)0, (size_t)0xFFFFFFF0u));
        rb[q] = p3_rsrc(p.B + (size_t)n0 * p.ldb + 32 * q, (unsigned)min(ball > 64u * q ? ball - 64u * q : (size_t)0, (size_t)0xFFFFFFF0u));
    }
    // lane l of wave w, request half r: LDS piece (row 32 w + 16 r + l / 4, piece l & 3) <- source piece (l & 3) ^ ((row >> 2) & 3)
    const unsigned row = 32u * (unsigned)wave + (unsigned)(lane >> 2), sp = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    unsigned va0 = (row * (unsigned)p.lda + 8u * sp) * 2u, va1 = va0 + 16u * (unsigned)p.lda * 2u;
    unsigned vb0 = (row * (unsigned)p.ldb + 8u * sp) * 2u, vb1 = vb0 + 16u * (unsigned)p.ldb * 2u;

    unsigned fa0[TM], fb0[TNN], fa1[TM], fb1[TNN];             // fragment offsets inside a slab: k-half 0, k-half 1 (^ 32)
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
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_u8*)p3_smem;
    const unsigned wave_off = (unsigned)wave * 2048u;
    bf16x8 A0[TM], A1[TM], B0[TNN], B1[TNN];

    auto mma = [&](const bf16x8 (&X)[TM], const bf16x8 (&Y)[TNN]) {          // swapped operands: accumulator = C^T (as gemm_b1_kernel's NT form)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
            for (int j = 0; j < TNN; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Y[j], X[ii], acc[ii][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto rd = [&](bf16x8 (&X)[TM], bf16x8 (&Y)[TNN], const unsigned char* S, int slab, const unsigned (&fa)[TM], const unsigned (&fb)[TNN]) {
#pragma unroll
        for (int ii = 0; ii < TM; ++ii) X[ii] = *reinterpret_cast<const bf16x8*>(S + slab * B1W_SLAB + fa[ii]);
#pragma unroll
        for (int j = 0; j < TNN; ++j) Y[j] = *reinterpret_cast<const bf16x8*>(S + (2 + slab) * B1W_SLAB + fb[j]);
    };
    auto dma_a = [&](unsigned buf) {
        b1w_dma4(lds_base + buf + 0 * B1W_SLAB + wave_off, lds_base + buf + 1 * B1W_SLAB + wave_off, va0, va1, ra[0], ra[1]);
        va0 += 128u; va1 += 128u;
    };
    auto dma_b = [&](unsigned buf) {
        b1w_dma4(lds_base + buf + 2 * B1W_SLAB + wave_off, lds_base + buf + 3 * B1W_SLAB + wave_off, vb0, vb1, rb[0], rb[1]);
        vb0 += 128u; vb1 += 128u;
    };

    if (nd > 0) {
        dma_a(0u); dma_b(0u);
        if (nd > 1) {
            dma_a((unsigned)B1W_BUF);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        p3_barrier();
        rd(A0, B0, p3_smem, 0, fa0, fb0);                        // chunk 0 of D_0
        for (int j = 0; j < nd; ++j) {
            const unsigned b0 = (unsigned)(j & 1) * (unsigned)B1W_BUF, b1 = (unsigned)B1W_BUF - b0;
            const unsigned char* S0 = p3_smem + b0;
            const unsigned char* S1 = p3_smem + b1;
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < nd) dma_b(b1);                           // (its buffer's last B reads are behind the barrier of iteration j - 1)
            __builtin_amdgcn_sched_barrier(0);
            rd(A1, B1, S0, 0, fa1, fb1);                         // chunk 1: slab 0, k-half 1
            mma(A0, B0);
            rd(A0, B0, S0, 1, fa0, fb0);                         // chunk 2: slab 1, k-half 0
            mma(A1, B1);
            rd(A1, B1, S0, 1, fa1, fb1);                         // chunk 3
            mma(A0, B0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // D_{j+1}: this wave's requests have landed; after the barrier every wave's have,
            p3_barrier();                                         // and every wave is past its last fragment read of D_j
            if (j + 2 < nd) dma_a(b0);                           // the A slabs of D_{j+2} into D_j's buffer
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < nd) rd(A0, B0, S1, 0, fa0, fb0);         // chunk 0 of D_{j+1}
            mma(A1, B1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    b1_nt_epilogue<EPI>(p, acc, m0, n0, wm0, wn0, lane, wave);
}

// ---- split3 of an fp32 matrix into planes (weights, once per step; test helper for whole operands)
// dst[q][r][c] (plane stride ps) = plane q of X[r][c]; dstT[q][c][r] likewise for the transposed matrix (either may be NULL)
__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ X, int R, int Cc, int ld, __bf16* __restrict__ dst, long long ps,
                                                int ldd, __bf16* __restrict__ dstT, long long psT, int lddT) {
    const size_t n = (size_t)R * Cc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / Cc), c = (int)(i % Cc);
        __bf16 h, m, l;
        split3(X[(size_t)r * ld + c], h, m, l);
        if (dst) { __bf16* d = dst + (size_t)r * ldd + c; d[0] = h; d[ps] = m; d[2 * ps] = l; }
        if (dstT) { __bf16* d = dstT + (size_t)c * lddT + r; d[0] = h; d[psT] = m; d[2 * psT] = l; }
    }
}

extern "C" int cham_split3(const float* X, int R, int Cc, int ld, void* dst, long long plane_stride, int ldd, void* dstT,
                           long long plane_strideT, int lddT, void* stream) {
    if (!X || R <= 0 || Cc <= 0 || (!dst && !dstT)) return -CHAM_ERR_ARG;
    const size_t n = (size_t)R * Cc;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_split3, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, R, Cc, ld, reinterpret_cast<__bf16*>(dst), plane_stride,
                       ldd, reinterpret_cast<__bf16*>(dstT), plane_strideT, lddT);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// launch counters: [0] NT launches, [1] TN launches, [2] / [3] NT / TN launches of the one-plane bf16 form, [6] epilogue and [7] K-splits of the last launch
static long long g_p3_launches[8];
extern "C" void cham_gemm_p3_launch_counts(long long* out8, int reset) {
    for (int i = 0; i < 8; ++i) { if (out8) out8[i] = g_p3_launches[i]; if (reset) g_p3_launches[i] = 0; }
}

template <bool TN, int EPI>
static int p3_launch(P3Params& p, hipStream_t st) {
    g_p3_launches[6] = EPI; g_p3_launches[7] = p.splits;
    constexpr int smem = P3_RING * P3_STAGE;
    auto k = gemm_p3_kernel<TN, EPI>;
    CHAM_SET_DYNAMIC_LDS(k, smem);
    hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, p.splits, 1), dim3(512), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// C[M,N] = epi(sum of six plane products) - see the header.  A, B: plane 0 (bf16), planes `*_plane_stride` elements apart.
//   tn = 0 (NT): A [M, lda], B [N, ldb], k contiguous; K % 16 == 0.  bias + act (CHAM_ACT_NONE / CHAM_ACT_TANH), or dref_h + dact =
//     CHAM_ACT_LEAKY: x leaky'(saved activation) with dref_h the h plane [M, ldr] of that activation.
//   tn = 1 (TN): A stored [K, lda >= M], B stored [K, ldb >= N]; M % 256 == 0, N % 256 == 0, any K; split-K through `workspace`
//     (splits_hint: 1 none, 0 automatic, n at most n; fixed-order reduction), accumulate adds to C.
// Returns -CHAM_ERR_ARG for shapes it does not take (the caller keeps cham_gemm_f32x3 for those).
extern "C" int cham_gemm_p3(const void* A, long long a_plane_stride, int lda, const void* B, long long b_plane_stride, int ldb, int tn,
                            float* C, int ldc, int M, int N, int K, const float* bias, int act, const void* dref_h, int ldr, int dact,
                            int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return -CHAM_ERR_ARG;
    if ((lda & 7) || (ldb & 7) || (a_plane_stride & 7) || (b_plane_stride & 7) || (N & 3) || (ldc & 3)) return -CHAM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return -CHAM_ERR_ARG;
    if ((size_t)ldc * 4 * 256 >= WINDOW_BYTES || (size_t)ldr * 2 * 256 >= WINDOW_BYTES) return -CHAM_ERR_ARG;
    P3Params p;
    p.A = reinterpret_cast<const __bf16*>(A); p.B = reinterpret_cast<const __bf16*>(B); p.a_ps = a_plane_stride; p.b_ps = b_plane_stride;
    p.lda = lda; p.ldb = ldb; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.dref = reinterpret_cast<const __bf16*>(dref_h); p.ldr = ldr; p.partial = workspace; p.xcd_split = 0; p.accumulate = 0;
    p.nbm = (M + 255) / 256; p.nbn = (N + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (!tn) {
        if ((K & 15) || accumulate) return -CHAM_ERR_ARG;
        if ((size_t)256 * lda * 2 >= (1ull << 31) || (size_t)256 * ldb * 2 >= (1ull << 31)) return -CHAM_ERR_ARG;
        p.kchunk = K; p.splits = 1;
        ++g_p3_launches[0];
        if (dref_h && (bias || act != ACT_NONE || dact != ACT_LEAKY)) return -CHAM_ERR_ARG;
        if (act != ACT_NONE && !(bias && act == ACT_TANH)) return -CHAM_ERR_ARG;
        if (dref_h) return p3_launch<false, 3>(p, st);
        if (bias) {
            if (act == ACT_TANH) return p3_launch<false, 2>(p, st);
            if (act == ACT_NONE) return p3_launch<false, 5>(p, st);
            return -CHAM_ERR_ARG;
        }
        if (act != ACT_NONE) return -CHAM_ERR_ARG;
        return p3_launch<false, 0>(p, st);
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
        if (splits_hint <= 0 && want >= 8) want = want / 8 * 8;      // an explicit count is taken as given (e.g. 14 x 16 tiles = 224 workgroups: one round that leaves 32 CUs to the kernels of the other lane)
        if (want > 1) splits = (int)want;
    }
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + 15) / 16) * 16;
    p.kchunk = kchunk;
    p.splits = (K + kchunk - 1) / kchunk;
    if ((size_t)kchunk * (lda > ldb ? lda : ldb) * 2 >= 0xFFFFFFF0ull) return -CHAM_ERR_ARG;
    ++g_p3_launches[1];
    if (p.splits > 1) {
        p.xcd_split = (p.splits % 8 == 0) ? 1 : 0;
        const int rc = p3_launch<true, 6>(p, st);
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
    return p3_launch<true, 0>(p, st);
}

// ---- bf16 configuration: one bf16 plane per operand on the LDS-DMA core (gemm_b1_kernel above)
//   tn = 0 (NT): A [M, lda], B [N, ldb] bf16, k contiguous, K % 16 == 0; C bf16 [M, ldc] = bf16(tanh(A B^T + bias)) (bias fp32 + act =
//     CHAM_ACT_TANH), bf16(A B^T x leaky'(dref)) (dref = saved bf16 activation [M, ldr], dact = CHAM_ACT_LEAKY) or bf16(A B^T).
//   tn = 1 (TN): A stored [K, lda >= M], B stored [K, ldb >= N] bf16; M % 256 == 0, N % 256 == 0; C fp32 [M, ldc] (+= with accumulate),
//     split-K through `workspace` (splits_hint as cham_gemm_p3; fixed-order reduction).
// Returns -CHAM_ERR_ARG for shapes it does not take (the caller keeps cham_gemm_b16 for those).  Launch counters: [2] NT, [3] TN.
template <bool TN, int EPI>
static int b1_launch(P3Params& p, hipStream_t st) {
    g_p3_launches[6] = EPI; g_p3_launches[7] = p.splits;
    constexpr int smem = P3_RING * P3_STAGE;
    auto k = gemm_b1_kernel<TN, EPI>;
    CHAM_SET_DYNAMIC_LDS(k, smem);
    hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, p.splits, 1), dim3(512), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// NT launches take the 64-byte-piece kernel (gemm_b1w_kernel) when K % 64 == 0, unless switched off (A/B arm, tests): launch counter [4]
static int g_b1_nt_wide = 1;
extern "C" int cham_gemm_b16_dma_set_nt_wide(int on) { const int was = g_b1_nt_wide; g_b1_nt_wide = on ? 1 : 0; return was; }
template <int EPI>
static int b1w_launch(P3Params& p, hipStream_t st) {
    g_p3_launches[6] = EPI; g_p3_launches[7] = 1; ++g_p3_launches[4];
    constexpr int smem = 2 * B1W_BUF;
    auto k = gemm_b1w_kernel<EPI>;
    CHAM_SET_DYNAMIC_LDS(k, smem);
    hipLaunchKernelGGL(k, dim3(p.nbm * p.nbn, 1, 1), dim3(512), smem, st, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_gemm_b16_dma(const void* A, int lda, const void* B, int ldb, int tn, void* C, int ldc, int M, int N, int K,
                                 const float* bias, int act, const void* dref, int ldr, int dact, int accumulate, float* workspace,
                                 size_t workspace_bytes, int splits_hint, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return -CHAM_ERR_ARG;
    if ((lda & 7) || (ldb & 7) || (N & 3) || (ldc & 3) || (dref && (ldr & 3))) return -CHAM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)dref | (uintptr_t)bias) & 15) return -CHAM_ERR_ARG;
    if ((size_t)ldc * 4 * 256 >= WINDOW_BYTES || (size_t)ldr * 2 * 256 >= WINDOW_BYTES) return -CHAM_ERR_ARG;
    P3Params p;
    p.A = reinterpret_cast<const __bf16*>(A); p.B = reinterpret_cast<const __bf16*>(B); p.a_ps = 0; p.b_ps = 0;
    p.lda = lda; p.ldb = ldb; p.C = reinterpret_cast<float*>(C); p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.dref = reinterpret_cast<const __bf16*>(dref); p.ldr = ldr; p.partial = workspace; p.xcd_split = 0; p.accumulate = 0;
    p.nbm = (M + 255) / 256; p.nbn = (N + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (!tn) {
        if ((K & 15) || accumulate) return -CHAM_ERR_ARG;
        if ((size_t)256 * lda * 2 >= (1ull << 31) || (size_t)256 * ldb * 2 >= (1ull << 31)) return -CHAM_ERR_ARG;
        p.kchunk = K; p.splits = 1;
        ++g_p3_launches[2];
        const bool wide = g_b1_nt_wide && (K & 63) == 0;
        if (dref) {
            if (bias || act != ACT_NONE || dact != ACT_LEAKY) return -CHAM_ERR_ARG;
            return wide ? b1w_launch<13>(p, st) : b1_launch<false, 13>(p, st);
        }
        if (bias) {
            if (act != ACT_TANH) return -CHAM_ERR_ARG;
            return wide ? b1w_launch<12>(p, st) : b1_launch<false, 12>(p, st);
        }
        if (act != ACT_NONE) return -CHAM_ERR_ARG;
        return wide ? b1w_launch<10>(p, st) : b1_launch<false, 10>(p, st);
    }
    if ((M & 255) || (N & 255) || bias || act != ACT_NONE || dref) return -CHAM_ERR_ARG;
    if ((size_t)48 * lda * 2 >= (1ull << 31) || (size_t)48 * ldb * 2 >= (1ull << 31)) return -CHAM_ERR_ARG;
    const long tiles = (long)p.nbm * p.nbn;
    int splits = 1;
    if (splits_hint != 1 && workspace) {
        long want = splits_hint > 1 ? splits_hint : (tiles >= 192 ? 1 : (256 + tiles - 1) / tiles);
        const long maxk = (K + 1535) / 1536;
        if (want > maxk) want = maxk;
        const long maxw = (long)(workspace_bytes / ((size_t)M * N * sizeof(float)));
        if (want > maxw) want = maxw;
        if (splits_hint <= 0 && want >= 8) want = want / 8 * 8;
        if (want > 1) splits = (int)want;
    }
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + 47) / 48) * 48;
    p.kchunk = kchunk;
    p.splits = (K + kchunk - 1) / kchunk;
    if ((size_t)kchunk * (lda > ldb ? lda : ldb) * 2 >= 0xFFFFFFF0ull) return -CHAM_ERR_ARG;
    ++g_p3_launches[3];
    if (p.splits > 1) {
        p.xcd_split = (p.splits % 8 == 0) ? 1 : 0;
        const int rc = b1_launch<true, 6>(p, st);
        if (rc != CHAM_OK) return rc;
        GemmParams g;
        g.A = nullptr; g.B = nullptr; g.C = reinterpret_cast<float*>(C); g.M = M; g.N = N; g.K = K; g.lda = 0; g.ldb = 0; g.ldc = ldc; g.bias = nullptr;
        g.act = ACT_NONE; g.dref = nullptr; g.ldr = 0; g.dact = ACT_NONE; g.rs = nullptr; g.ldrs = 0; g.rs_div = 1; g.accumulate = accumulate;
        g.kchunk = kchunk; g.splits = p.splits; g.partial = workspace; g.nbm = p.nbm; g.nbn = p.nbn; g.xcd_split = p.xcd_split;
        launch_splitk_reduce(g, st);
        CHAM_CHECK_LAUNCH();
        return CHAM_OK;
    }
    p.accumulate = accumulate;
    return b1_launch<true, 0>(p, st);
}
