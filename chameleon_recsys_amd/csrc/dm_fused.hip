// Scorer layer-1 dgrad fused with the `cand (.) pred` backward (round 3; gfx950, wave64).
//
//   dM[r, k]   = sum_j dS1[r, j] Ws1[k, j]                        (autodiff of nar_model.py:447-473, matching_dense_layer_1; K = 128)
//   dZ2[r, k]  = dM[r, k] * pred[p(r), k] * (1 - Z2c[r, k]^2)     (autodiff of cand (.) pred and the CAR tanh, :478-495, :384-388)
//   dpred[p,k] = (sum_{r in p} dM[r, k] Z2c[r, k]) * (1 - pred[p, k]^2)
//   b2part[p,k]= sum_{r in p} dZ2[r, k]                            (the position's share of the CAR bias gradient)
//
// Unfused (cham_gemm_f32x3 + cham_mulpred_bwd_p3) the [B*T*(1+N), C] matrix dM is written as fp32 (1 GB at the G1 shape) by a K = 128
// GEMM that is all epilogue, and read straight back: 1.19 + 0.73 ms of a 12.3 ms step, on the critical path between the scorer backward
// and the CAR dgrad (profiles/r03_notes.md).  Here dM exists only in the MFMA accumulators; HBM sees dS1 (0.13 GB), Z2c (1 GB) in
// and the three bf16 planes of dZ2 (1.5 GB) out.
//
// One workgroup = PW = floor(256 / (1+N)) whole positions (their PW * (1+N) <= 256 candidate rows; (1+N) >= 32) x ALL C columns:
//   * 8 waves, wave w owns rows [32 w, 32 w + 32): its dS1 fragments (32 rows x K = 128, three bf16 planes: 96 VGPRs) are loaded and split
//     once and stay in registers;
//   * the columns are walked in tiles of 64 = two 32x32 MFMA tiles per wave whose B fragments are PERMUTED - fragment row n of tile j is
//     Ws1 row 2 n + j - so lane n ends up with two CONSECUTIVE columns (8-byte Z2c loads, 4-byte plane stores, whole 256- / 128-byte row
//     segments per half-wave) while keeping 16 rows of those columns in one lane: the per-position column sums are in-lane adds + one
//     half-wave exchange, then a fixed-order sum over the waves through LDS.  No float atomics.
//   * Ws1's planes (written once per step by cham_split3) stream through LDS by LDS-DMA in half-stages of 64 columns x 64 k x 3 planes
//     = 24 KB, two buffers; six plane products as in gemm_x3.hip / gemm_p3.hip.
// Per half-stage and wave: 48 MFMAs, 24 ds_read_b128, 3 DMA requests, one barrier; every second half-stage the epilogue of a column tile.
#include "gemm_shared.h"
#include <type_traits>

#define DMF_NT 2                              // 32-column MFMA tiles per wave and column tile (4: 128 columns per tile spills the register file)
#define DMF_COLS (32 * DMF_NT)                // columns per column tile
#define DMF_PLANE (DMF_COLS * 128)            // one plane of a half-stage: DMF_COLS slots x 64 k x 2 B
#define DMF_HALF (3 * DMF_PLANE)              // one half-stage: three planes
#define DMF_RED_OFF (2 * DMF_HALF)            // [8 waves][2 slots][2 quantities][DMF_COLS columns] fp32
#define DMF_RED_BYTES (8 * 4 * DMF_COLS * 4)

static_assert(DMF_NT == 2, "the epilogue packs exactly two tiles' columns per lane");

struct DmfParams {
    const float* dS1; int lds1;                 // [Rc, 128] (fp32; MODE 2: bf16, lds1 in elements)
    const __bf16* W; long long w_ps;            // planes of Ws1 [C, 128] as stored, plane stride in elements (MODE 2: the one bf16 shadow)
    const float* Z2c; const float* pred;        // [Rc, C] (MODE 2: bf16), [BT, C]
    __bf16* out; long long out_ps;              // planes of dZ2 [Rc, C]: three bf16 planes, or (H2) two fp16 planes x the scale of osc
    const H2Scale* osc;
    const H2Scale* sa; const H2Scale* sw;       // MODE 3: the scales of dS1 and of Ws1's two fp16 planes
    float* dpred; float* b2part;                // [BT, C]
    int C, BT, NC, PW;
    int out_blocked;                            // H2: the planes of dZ2 are written tile-blocked (common.h h2b_index)
};

__device__ __forceinline__ void dmf_dma_one(unsigned lds, unsigned voff, const u32x4& r) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "s"(lds), "v"(voff), "s"(r) : "memory");
}

// MODE 0: three bf16 planes out (csrc/gemm_p3.hip).  MODE 1: the output as two fp16 planes x a power-of-two scale (csrc/gemm_h2.hip); the
// kernel's own products (dS1 x Ws1, K = 128: 12 % of its time) stay six bf16 plane products - its time is the 1 GB of Z2 in and the planes
// out.  MODE 2: the bf16 configuration (BASELINE configs[2]) - dS1, Ws1 (its bf16 shadow), Z2c and the output are single bf16 matrices, ONE
// product; dM is rounded to bf16 where the unfused pair (cham_gemm_b16 + cham_mulpred_bwd_b16) stores it, so the result is that pair's up
// to the summation order of the per-position sums; col_part sums the ROUNDED gradient (what cham_colsum_b16 would read back).
// MODE 3 (round 5): MODE 1 with the kernel's own products on TWO fp16 planes too - dS1 split with the scale of p.sa (its max row norm), Ws1's
// planes (cham_split2h with the scale of p.sw) - THREE v_mfma_f32_32x32x16_f16 products instead of six, dM scaled back by 1 / (s_a s_w).
template <int MODE>
__global__ __launch_bounds__(512) void k_dm_mulpred_fused(DmfParams p) {
    constexpr bool H2 = MODE == 1 || MODE == 3, B16 = MODE == 2, F16P = MODE == 3;
    constexpr int NPL = B16 ? 1 : (F16P ? 2 : 3);                         // planes of each operand
    extern __shared__ __attribute__((aligned(1024))) unsigned char dmf_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int NC = p.NC, C = p.C;
    const int pos0 = blockIdx.x * p.PW;                                   // first position of this workgroup
    const int npos = min(p.PW, p.BT - pos0);
    const int rows_valid = npos * NC;                                     // <= 256
    const size_t row0 = (size_t)pos0 * NC;                                // first candidate row
    const int wr0 = 32 * wave;                                            // this wave's first row inside the workgroup

    // ---- A: this wave's 32 rows of dS1, K = 128, split into planes; lane (row l31, k-half hh) holds 8 consecutive k per 16-k step
    // (every global access of this kernel goes through a buffer descriptor sized to the workgroup's valid rows: rows beyond them read
    // zeros / drop their stores without a branch - a branch around a load makes hipcc wait for each load in turn, 16 dependent HBM round
    // trips per column tile in the first version of this epilogue: 1.25 ms for the kernel, profiles/r03_notes.md)
    bf16x8 AH[8], AM[(B16 || F16P) ? 1 : 8], AL[B16 ? 1 : 8];          // (F16P: fp16 bit patterns in the same 16-bit containers)
    float ginv_a = 1.f, ginv_w = 1.f;      // the two inverse power-of-two scales, applied ONE AFTER THE OTHER: their product can leave the fp32 normal range (each exponent is clamped to +-110)
    if constexpr (B16) {
        const __bf16* a16 = reinterpret_cast<const __bf16*>(p.dS1);
        const __amdgpu_buffer_rsrc_t aw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a16 + row0 * (size_t)p.lds1), 0,
                                                                            (unsigned)((size_t)rows_valid * p.lds1 * 2), 0x00020000);
        const unsigned ao = ((unsigned)(wr0 + l31) * (unsigned)p.lds1 + 8u * hh) * 2u;
#pragma unroll
        for (int s = 0; s < 8; ++s) AH[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(aw, ao + 32u * s, 0, 0));
    } else {
        const __amdgpu_buffer_rsrc_t aw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dS1 + row0 * (size_t)p.lds1), 0,
                                                                            (unsigned)((size_t)rows_valid * p.lds1 * 4), 0x00020000);
        const unsigned ao = ((unsigned)(wr0 + l31) * (unsigned)p.lds1 + 8u * hh) * 4u;
        u32x4 x[16];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x[2 * s] = __builtin_amdgcn_raw_buffer_load_b128(aw, ao + 64u * s, 0, 0);
            x[2 * s + 1] = __builtin_amdgcn_raw_buffer_load_b128(aw, ao + 64u * s + 16u, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float v[8] = {__uint_as_float(x[2 * s].x), __uint_as_float(x[2 * s].y), __uint_as_float(x[2 * s].z), __uint_as_float(x[2 * s].w),
                                __uint_as_float(x[2 * s + 1].x), __uint_as_float(x[2 * s + 1].y), __uint_as_float(x[2 * s + 1].z),
                                __uint_as_float(x[2 * s + 1].w)};
            if constexpr (F16P) {
                const float sa = p.sa->scale;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 a, c;
                    split2h(v[e] * sa, a, c);
                    AH[s][e] = __builtin_bit_cast(__bf16, a); AL[s][e] = __builtin_bit_cast(__bf16, c);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { __bf16 a, b, c; split3(v[e], a, b, c); AH[s][e] = a; AM[s][e] = b; AL[s][e] = c; }
            }
        }
    }
    const __amdgpu_buffer_rsrc_t zw = B16 ?
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(reinterpret_cast<const __bf16*>(p.Z2c) + row0 * (size_t)C), 0,
                                          (unsigned)((size_t)rows_valid * C * 2), 0x00020000) :
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Z2c + row0 * (size_t)C), 0, (unsigned)((size_t)rows_valid * C * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t ow[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)        // (H2: two planes of 16-bit elements, B16: one - the other descriptors are never used)
        ow[q] = __builtin_amdgcn_make_buffer_rsrc(p.out + ((H2 && q == 2) || B16 ? 0 : q) * p.out_ps + row0 * (size_t)C, 0, (unsigned)((size_t)rows_valid * C * 2), 0x00020000);
    // tile-blocked output (H2 only): the workgroup's <= 256 rows start `rin` rows into row tile row0 / 256 and reach at most into the next
    // one; the windows start at that tile and span two (the allocation's last tile is followed by at least the other plane or the pad the
    // caller allocates); rows beyond the valid ones get an out-of-range offset (their stores are dropped)
    const bool oblk = H2 && p.out_blocked != 0;
    const unsigned rin = (unsigned)(row0 & 255), tile_bytes = ((unsigned)C >> 5) * (unsigned)(H2B_BLOCK * 2);
    if (oblk) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            ow[q] = __builtin_amdgcn_make_buffer_rsrc(p.out + q * p.out_ps + (row0 >> 8) * (size_t)((unsigned)C >> 5) * H2B_BLOCK, 0, 2u * tile_bytes, 0x00020000);
    }
    float osc = 1.f;
    if constexpr (H2) osc = p.osc->scale;
    if constexpr (F16P) { ginv_a = p.sa->inv; ginv_w = p.sw->inv; }

    // ---- B: LDS-DMA of Ws1's planes.  Half-stage image per plane: [slot 0..127][64 k] (128 B per slot), slot = j * 32 + n holds Ws1 row
    // n0 + 4 n + j; the eight 16-byte pieces of a slot are XOR-ed with (slot >> 1) & 7 (conflict-free ds_read_b128 fragments).  One
    // request fills 8 slots: lane l -> slot 8 c + l / 8, LDS piece l & 7, i.e. source piece (l & 7) ^ ((slot >> 1) & 7).
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_u8*)dmf_smem;
    u32x4 rw[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const unsigned long long a = (unsigned long long)(p.W + q * p.w_ps);
        rw[q].x = (unsigned)a; rw[q].y = (unsigned)(a >> 32) & 0xFFFFu; rw[q].z = (unsigned)((size_t)C * 128 * 2); rw[q].w = 0x00020000u;
    }
    // one request per plane and half-stage fills this wave's 8 slots
    unsigned voff;
    {
        const int slot = 8 * wave + (lane >> 3);
        const int piece = (lane & 7) ^ ((slot >> 1) & 7);
        const int col = DMF_NT * (slot & 31) + (slot >> 5);               // column inside the column tile
        voff = (unsigned)(col * 128 + piece * 8) * 2u;
    }
    auto dma_half = [&](int h) {                                          // half-stage h: column tile h / 2, k half h & 1 -> buffer h & 1
        const unsigned base = lds_base + (unsigned)(h & 1) * DMF_HALF + (unsigned)wave * 1024u;
        const unsigned add = (unsigned)(((h >> 1) * DMF_COLS) * 128 + (h & 1) * 64) * 2u;
#pragma unroll
        for (int q = 0; q < NPL; ++q) dmf_dma_one(base + (unsigned)q * DMF_PLANE, voff + add, rw[q]);
    };
    // fragment read offsets: tile j, lane (n = l31, hh), 16-k step t of the half: slot = 32 j + n, piece (2 t + hh) ^ ((slot >> 1) & 7)
    // ((slot >> 1) & 7 does not depend on j: tile j is the same lane offset + j * 4096)
    unsigned fo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) fo[t] = (unsigned)(l31 * 128 + (((2 * t + hh) ^ ((l31 >> 1) & 7)) * 16));

    // ---- rows of this lane: e -> local row wr0 + (e & 3) + 8 (e >> 2) + 4 hh; positions: the wave's 32 rows span at most two (NC >= 32)
    const int pa = wr0 / NC;                                              // local position of the wave's first row
    const int bnd = (pa + 1) * NC - wr0;                                  // rows (relative to wr0) >= bnd belong to position pa + 1
    const int ntiles = C / DMF_COLS, nhalf = 2 * ntiles;

    dma_half(0);
    dma_half(1);
    floatx16 acc[DMF_NT];
#pragma unroll
    for (int j = 0; j < DMF_NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    if constexpr (B16) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // half-stage 0 landed (this wave's requests: NPL per half-stage)
    else if constexpr (F16P) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");

    // the MFMAs of one half-stage; PAR = k half (compile-time: the A fragments are register arrays and must be indexed statically)
    auto mfma_half = [&](auto PARC) {
        constexpr int PAR = decltype(PARC)::value, s0 = PAR * 4;
        const unsigned char* S = dmf_smem + PAR * DMF_HALF;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x8 bh[DMF_NT], bm[DMF_NT], bl[DMF_NT];
#pragma unroll
            for (int j = 0; j < DMF_NT; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(S + fo[t] + j * 4096);
                if constexpr (F16P) {
                    bl[j] = *reinterpret_cast<const bf16x8*>(S + DMF_PLANE + fo[t] + j * 4096);
                } else if constexpr (!B16) {
                    bm[j] = *reinterpret_cast<const bf16x8*>(S + DMF_PLANE + fo[t] + j * 4096);
                    bl[j] = *reinterpret_cast<const bf16x8*>(S + 2 * DMF_PLANE + fo[t] + j * 4096);
                }
            }
            if constexpr (F16P) {
                typedef _Float16 dmf_half8 __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dmf_half8, AL[s0 + t]), __builtin_bit_cast(dmf_half8, bh[j]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dmf_half8, AH[s0 + t]), __builtin_bit_cast(dmf_half8, bl[j]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dmf_half8, AH[s0 + t]), __builtin_bit_cast(dmf_half8, bh[j]), acc[j], 0, 0, 0);
            } else if constexpr (B16) {
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[s0 + t], bh[j], acc[j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL[s0 + t], bh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[s0 + t], bl[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM[s0 + t], bm[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM[s0 + t], bh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[s0 + t], bm[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < DMF_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[s0 + t], bh[j], acc[j], 0, 0, 0);
            }
        }
    };
    for (int h = 0; h < nhalf; ++h) {
        if (h & 1) mfma_half(std::integral_constant<int, 1>{});
        else mfma_half(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        // this wave's requests for half-stage h + 1 (issued a half-stage ago) have landed; nothing else of its is in flight except
        // the previous tile's stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (h & 1) {
            // ---- epilogue of column tile h / 2: lane n holds columns n0 + 2 n, n0 + 2 n + 1 of its 16 rows
            const int col = (h >> 1) * DMF_COLS + DMF_NT * l31;
            float2 pra = make_float2(0.f, 0.f), prb = pra;
            if (pa < npos) pra = *reinterpret_cast<const float2*>(p.pred + (size_t)(pos0 + pa) * C + col);
            if (pa + 1 < npos) prb = *reinterpret_cast<const float2*>(p.pred + (size_t)(pos0 + pa + 1) * C + col);
            float2 sa = make_float2(0.f, 0.f), sb = sa, ca = sa, cb = sa;
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            u32x2_t zz[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {      // all sixteen row loads in flight at once (rows beyond the valid ones: zeros)
                const unsigned r = (unsigned)(wr0 + (e & 3) + 8 * (e >> 2) + 4 * hh);
                if constexpr (B16) {            // two bf16 -> two fp32 bit patterns (exact)
                    const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(zw, (r * (unsigned)C + (unsigned)col) * 2u, 0, 0);
                    zz[e].x = u << 16; zz[e].y = u & 0xFFFF0000u;
                } else {
                    zz[e] = __builtin_amdgcn_raw_buffer_load_b64(zw, (r * (unsigned)C + (unsigned)col) * 4u, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * hh;           // row relative to the wave's first row
                const unsigned r = (unsigned)(wr0 + rr);
                const float2 z = make_float2(__uint_as_float(zz[e].x), __uint_as_float(zz[e].y));
                const bool second = rr >= bnd;
                const float2 pr = second ? prb : pra;
                float2 g = make_float2(acc[0][e], acc[1][e]);      // (zero on rows beyond the valid ones: their dS1 rows read as zeros)
                if constexpr (F16P) { g.x = (g.x * ginv_a) * ginv_w; g.y = (g.y * ginv_a) * ginv_w; }  // back to true units (powers of two)
                if constexpr (B16) { g.x = (float)(__bf16)g.x; g.y = (float)(__bf16)g.y; }      // dM as the unfused pair stores it
                float2 o;
                o.x = g.x * pr.x * (1.f - z.x * z.x); o.y = g.y * pr.y * (1.f - z.y * z.y);
                if constexpr (B16) {
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    bf16x2_t ph; ph[0] = (__bf16)o.x; ph[1] = (__bf16)o.y;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ph), ow[0], (r * (unsigned)C + (unsigned)col) * 2u, 0, 0);
                    o.x = (float)ph[0]; o.y = (float)ph[1];                  // the b2 partial sums add what is stored
                } else if constexpr (H2) {
                    _Float16 h0, l0, h1, l1;
                    split2h(o.x * osc, h0, l0); split2h(o.y * osc, h1, l1);
                    const unsigned ph = (unsigned)h2_keep_sign(h0, o.x) | ((unsigned)h2_keep_sign(h1, o.y) << 16);
                    const unsigned pl = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                    unsigned oo = (r * (unsigned)C + (unsigned)col) * 2u;
                    if (oblk) {
                        const unsigned lr = rin + r;              // row counted from the first row of the window's first tile
                        oo = (int)r < rows_valid ? (lr >> 8) * tile_bytes + ((unsigned)col >> 5) * (unsigned)(H2B_BLOCK * 2) + (lr & 255u) * 64u + ((unsigned)col & 31u) * 2u
                                                 : 0x80000000u;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(ph, ow[0], oo, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(pl, ow[1], oo, 0, 0);
                } else {
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    bf16x2_t ph, pm, pl;
                    __bf16 a, b, c;
                    split3(o.x, a, b, c); ph[0] = a; pm[0] = b; pl[0] = c;
                    split3(o.y, a, b, c); ph[1] = a; pm[1] = b; pl[1] = c;
                    const unsigned oo = (r * (unsigned)C + (unsigned)col) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ph), ow[0], oo, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pm), ow[1], oo, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pl), ow[2], oo, 0, 0);
                }
                const float2 gz = make_float2(g.x * z.x, g.y * z.y);
                sb.x += second ? gz.x : 0.f; sb.y += second ? gz.y : 0.f; cb.x += second ? o.x : 0.f; cb.y += second ? o.y : 0.f;
                sa.x += second ? 0.f : gz.x; sa.y += second ? 0.f : gz.y; ca.x += second ? 0.f : o.x; ca.y += second ? 0.f : o.y;
                acc[0][e] = 0.f; acc[1][e] = 0.f;
            }
            // the other half-wave holds the other 16 rows of the same columns: lower half + upper half, in that order
            float* red = reinterpret_cast<float*>(dmf_smem + DMF_RED_OFF) + (size_t)wave * (4 * DMF_COLS);       // [slot 2][quantity 2][DMF_COLS]
            float2 q[4] = {sa, ca, sb, cb};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float2 o;
                o.x = __shfl_xor(q[i].x, 32, 64); o.y = __shfl_xor(q[i].y, 32, 64);
                if (hh == 0) *reinterpret_cast<float2*>(red + i * DMF_COLS + DMF_NT * l31) = make_float2(q[i].x + o.x, q[i].y + o.y);
            }
        }
        // every wave: done reading buffer h & 1, its requests for h + 1 landed (and, on odd h, its partial sums are in LDS)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        if (h & 1) {
            // ---- per-position column sums of this column tile: thread (position, column) adds the waves in ascending order
            const float* redw = reinterpret_cast<const float*>(dmf_smem + DMF_RED_OFF);
            for (int idx = threadIdx.x; idx < npos * DMF_COLS; idx += 512) {
                const int pl = idx / DMF_COLS, c = idx % DMF_COLS;
                float s = 0.f, cs = 0.f;
                const int w_lo = (pl * NC) >> 5, w_hi = min(7, ((pl + 1) * NC - 1) >> 5);
                for (int w = w_lo; w <= w_hi; ++w) {
                    const int slot = (pl == (32 * w) / NC) ? 0 : 1;
                    s += redw[(size_t)w * (4 * DMF_COLS) + slot * (2 * DMF_COLS) + c];
                    cs += redw[(size_t)w * (4 * DMF_COLS) + slot * (2 * DMF_COLS) + DMF_COLS + c];
                }
                const int colg = (h >> 1) * DMF_COLS + c;
                const float pr = p.pred[(size_t)(pos0 + pl) * C + colg];
                p.dpred[(size_t)(pos0 + pl) * C + colg] = s * (1.f - pr * pr);
                if (p.b2part) p.b2part[(size_t)(pos0 + pl) * C + colg] = cs;
            }
            // (the next epilogue writes `red` again only after the next two barriers)
        }
        // (requested only now: the loads of the block above are visible to the compiler, whose wait for them would also wait for
        // requests issued before them)
        if (h + 2 < nhalf) dma_half(h + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// dS1 [BT * (1 + N), 128] (row stride lds1), planes of Ws1 [C, 128] (cham_split3 of the weight as stored), Z2c [BT * (1 + N), C], pred
// [BT, C] -> planes of dZ2 (plane stride out_plane_stride elements), dpred_pre [BT, C], col_part [BT, C] (may be NULL).
// Takes C % 64 == 0, 32 <= 1 + N <= 256 and K = 128 (the reference's matching_dense_layer_1 width); -EINVAL otherwise (the caller
// keeps cham_gemm_f32x3 + cham_mulpred_bwd_p3 / _h2).
template <int MODE>
static int dm_mulpred_launch(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const float* Z2c,
                             const float* pred, int C, int BT, int N, void* dZ2p, long long out_plane_stride, const void* out_scale_rec,
                             float* dpred_pre, float* col_part, void* stream, const void* a_scale_rec = nullptr, const void* w_scale_rec = nullptr,
                             int out_blocked = 0) {
    if (!dS1 || !Wp || !Z2c || !pred || !dZ2p || !dpred_pre || BT < 0 || N < 0 || ((MODE == 1 || MODE == 3) && !out_scale_rec)) return -CHAM_ERR_ARG;
    if (MODE == 3 && (!a_scale_rec || !w_scale_rec || (((uintptr_t)a_scale_rec | (uintptr_t)w_scale_rec) & 3))) return -CHAM_ERR_ARG;
    const int NC = N + 1;
    if (K != 128 || (C % DMF_COLS) || C <= 0 || NC < 32 || NC > 256 || (lds1 & 3) || lds1 < K || (out_plane_stride & 3) || (w_plane_stride & 7))
        return -CHAM_ERR_ARG;
    if (((uintptr_t)dS1 | (uintptr_t)Wp | (uintptr_t)Z2c | (uintptr_t)pred | (uintptr_t)dZ2p) & 15) return -CHAM_ERR_ARG;
    if (out_blocked && ((MODE != 1 && MODE != 3) || (C & 31) || out_plane_stride < (long long)(((size_t)BT * NC + 255) / 256) * (C >> 5) * H2B_BLOCK)) return -CHAM_ERR_ARG;
    if (BT == 0) return CHAM_OK;
    DmfParams p;
    p.out_blocked = out_blocked ? 1 : 0;
    p.dS1 = dS1; p.lds1 = lds1; p.W = reinterpret_cast<const __bf16*>(Wp); p.w_ps = w_plane_stride; p.Z2c = Z2c; p.pred = pred;
    p.out = reinterpret_cast<__bf16*>(dZ2p); p.out_ps = out_plane_stride; p.osc = reinterpret_cast<const H2Scale*>(out_scale_rec);
    p.sa = reinterpret_cast<const H2Scale*>(a_scale_rec); p.sw = reinterpret_cast<const H2Scale*>(w_scale_rec);
    p.dpred = dpred_pre; p.b2part = col_part;
    p.C = C; p.BT = BT; p.NC = NC; p.PW = 256 / NC;
    constexpr int smem = 2 * DMF_HALF + DMF_RED_BYTES;
    auto k = k_dm_mulpred_fused<MODE>;
    CHAM_SET_DYNAMIC_LDS(k, smem);
    hipLaunchKernelGGL(k, dim3((BT + p.PW - 1) / p.PW), dim3(512), smem, (hipStream_t)stream, p);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_dm_mulpred_p3(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const float* Z2c,
                                  const float* pred, int C, int BT, int N, void* dZ2p, long long out_plane_stride, float* dpred_pre,
                                  float* col_part, void* stream) {
    return dm_mulpred_launch<0>(dS1, lds1, K, Wp, w_plane_stride, Z2c, pred, C, BT, N, dZ2p, out_plane_stride, nullptr, dpred_pre, col_part, stream);
}

// as cham_dm_mulpred_p3 with dZ2 written as TWO fp16 planes x the scale of `out_scale_rec` (an H2Scale record holding a bound of
// max |dS1 Ws1^T|: cham_h2_scale_rownorm)
extern "C" int cham_dm_mulpred_h2(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const float* Z2c,
                                  const float* pred, int C, int BT, int N, void* dZ2p, long long out_plane_stride, const void* out_scale_rec,
                                  float* dpred_pre, float* col_part, void* stream) {
    return dm_mulpred_launch<1>(dS1, lds1, K, Wp, w_plane_stride, Z2c, pred, C, BT, N, dZ2p, out_plane_stride, out_scale_rec, dpred_pre, col_part, stream);
}

// ... and with the kernel's own products on two fp16 planes as well (MODE 3): Wh = plane 0 of cham_split2h(Ws1 [C, K]) under the scale of
// w_scale_rec (planes w_plane_stride elements apart), ds1_scale_rec = a record whose scale covers max |dS1| (cham_h2_scale_rownorm2's second record)
extern "C" int cham_dm_mulpred_h2h(const float* dS1, int lds1, int K, const void* Wh, long long w_plane_stride, const void* ds1_scale_rec,
                                   const void* w_scale_rec, const float* Z2c, const float* pred, int C, int BT, int N, void* dZ2p,
                                   long long out_plane_stride, const void* out_scale_rec, float* dpred_pre, float* col_part, void* stream) {
    return dm_mulpred_launch<3>(dS1, lds1, K, Wh, w_plane_stride, Z2c, pred, C, BT, N, dZ2p, out_plane_stride, out_scale_rec, dpred_pre, col_part, stream,
                                ds1_scale_rec, w_scale_rec);
}

// the two H2 forms with the planes of dZ2 written TILE-BLOCKED (common.h h2b_index; round 6): ds1_scale_rec / w_scale_rec NULL = cham_dm_mulpred_h2's
// arithmetic (six bf16 products inside, Wp = three bf16 planes), both given = cham_dm_mulpred_h2h's.  The caller allocates
// ceil(BT (1 + N) / 256) row tiles per plane (+ the planes follow each other or a pad of one tile: a workgroup's window spans two tiles),
// zero-initialised; rows beyond BT (1 + N) are never written.
extern "C" int cham_dm_mulpred_h2_blk(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const void* ds1_scale_rec,
                                      const void* w_scale_rec, const float* Z2c, const float* pred, int C, int BT, int N, void* dZ2p,
                                      long long out_plane_stride, const void* out_scale_rec, float* dpred_pre, float* col_part, void* stream) {
    if ((ds1_scale_rec == nullptr) != (w_scale_rec == nullptr)) return -CHAM_ERR_ARG;
    if (ds1_scale_rec)
        return dm_mulpred_launch<3>(dS1, lds1, K, Wp, w_plane_stride, Z2c, pred, C, BT, N, dZ2p, out_plane_stride, out_scale_rec, dpred_pre, col_part, stream,
                                    ds1_scale_rec, w_scale_rec, 1);
    return dm_mulpred_launch<1>(dS1, lds1, K, Wp, w_plane_stride, Z2c, pred, C, BT, N, dZ2p, out_plane_stride, out_scale_rec, dpred_pre, col_part, stream,
                                nullptr, nullptr, 1);
}

// The bf16 configuration's twin (BASELINE configs[2]): dS1 [BT*(1+N), K = 128] bf16 (row stride lds1 elements), Ws1b = the bf16 shadow of Ws1
// [C, K] as stored, Z2c [BT*(1+N), C] bf16, pred [BT, C] fp32 -> dZ2c bf16 [BT*(1+N), C], dpred_pre, col_part (sums of the stored, i.e.
// bf16-rounded, gradient rows).  Replaces cham_gemm_b16(dS1, Ws1) + cham_mulpred_bwd_b16 (+ the cham_colsum_b16 pass over dZ2c for b2).
extern "C" int cham_dm_mulpred_b16(const void* dS1, int lds1, int K, const void* Ws1b, const void* Z2c, const float* pred, int C, int BT, int N,
                                   void* dZ2c, float* dpred_pre, float* col_part, void* stream) {
    if ((lds1 & 7)) return -CHAM_ERR_ARG;
    return dm_mulpred_launch<2>(reinterpret_cast<const float*>(dS1), lds1, K, Ws1b, 0, reinterpret_cast<const float*>(Z2c), pred, C, BT, N, dZ2c, 0,
                                nullptr, dpred_pre, col_part, stream);
}
