// Candidate combine, scoring tail, sampled-softmax cross-entropy and their backward passes.
//
// Replaces (nar_module/nar/nar_model.py):
//   * :356-405  the PreCAR layer over [B,T,N,F] negatives.  PreCAR(x) = leaky(W1^T(gamma*[u;i]+beta)+b1)
//               = leaky(U[b,t] + V[item]) with U = X_ctx*W1_ctx + b1 (per click) and V = X_item*W1_item (per
//               unique item row, features.hip) - exact up to fp32 re-association, valid for keep_prob==1.
//   * :478-517  cand (.) pred  (fused into the scorer's first GEMM as a row-broadcast prologue),
//               the last scorer layer (32 -> 1), softmax(logits / tau) over [positive, N negatives]
//   * :639-667  masked negative log-likelihood and its gradient.
//
// Row layout of every "CAR row" matrix: rows [0,BT) = clicked inputs, rows [BT, BT + BT*(1+N)) =
// candidates ordered (b,t,c) with c = 0 the positive, c = 1..N the negatives.
// V row set: [0,BT) inputs, [BT,2BT) positives, [2BT, 2BT+pmax] pool slots (last = zero-padding item).
#include "common.h"

__device__ __forceinline__ int cand_vrow(int bt, int c, int BT, int N, int pmax, const int* __restrict__ neg_slot) {
    if (c == 0) return BT + bt;
    int s = neg_slot[(size_t)bt * N + (c - 1)];
    if (s < 0) s = pmax;                      // masked click: any finite row (its gradient is exactly 0)
    return 2 * BT + s;
}

// Z1[row,:] = leaky(U[u,:] + V[v,:])
__global__ __launch_bounds__(256) void k_combine_fwd(const float* __restrict__ U, const float* __restrict__ V, int C,
                                                     int BT, int N, int pmax, const int* __restrict__ neg_slot,
                                                     float* __restrict__ Z1, int row_begin) {
    const int row = row_begin + blockIdx.x;
    int u, v;
    if (row < BT) { u = row; v = row; }
    else {
        const int i = row - BT, bt = i / (N + 1), c = i % (N + 1);
        u = bt; v = cand_vrow(bt, c, BT, N, pmax, neg_slot);
    }
    const float4* pu = reinterpret_cast<const float4*>(U + (size_t)u * C);
    const float4* pv = reinterpret_cast<const float4*>(V + (size_t)v * C);
    float4* po = reinterpret_cast<float4*>(Z1 + (size_t)row * C);
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 a = pu[k], b = pv[k];
        float4 o;
        o.x = act_fwd(a.x + b.x, ACT_LEAKY); o.y = act_fwd(a.y + b.y, ACT_LEAKY);
        o.z = act_fwd(a.z + b.z, ACT_LEAKY); o.w = act_fwd(a.w + b.w, ACT_LEAKY);
        po[k] = o;
    }
}

// The candidate rows of one position share U[bt]: one workgroup per position keeps its U row in registers and streams the 1+N
// candidate rows (4 independent V-row loads in flight) - the row-per-workgroup form above pays a dependent slot -> V -> store
// chain and a workgroup launch per 4 KB written.
template <typename T>
__global__ __launch_bounds__(256) void k_combine_fwd_cand(const float* __restrict__ U, const float* __restrict__ V, int C,
                                                          int BT, int N, int pmax, const int* __restrict__ neg_slot,
                                                          T* __restrict__ Z1c /* candidate rows only */) {
    const int bt = blockIdx.x, NC = N + 1;
    const float4* pu = reinterpret_cast<const float4*>(U + (size_t)bt * C);
    T* po = Z1c + (size_t)bt * NC * C;
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 a = pu[k];
        for (int c0 = 0; c0 < NC; c0 += 4) {
            float4 b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i < NC ? c0 + i : NC - 1;
                b[i] = reinterpret_cast<const float4*>(V + (size_t)cand_vrow(bt, c, BT, N, pmax, neg_slot) * C)[k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i >= NC) break;
                float4 o;
                o.x = act_fwd(a.x + b[i].x, ACT_LEAKY); o.y = act_fwd(a.y + b[i].y, ACT_LEAKY);
                o.z = act_fwd(a.z + b[i].z, ACT_LEAKY); o.w = act_fwd(a.w + b[i].w, ACT_LEAKY);
                st4(po + (size_t)(c0 + i) * C + 4 * k, o);
            }
        }
    }
}

// The same rows as THREE bf16 PLANES (csrc/gemm_p3.hip): the split that gemm_x3.hip repeats in every column tile of every GEMM that
// reads Z1 is done here once, where the value is in a register anyway; 6 instead of 4 bytes stored per element, no fp32 copy.
__global__ __launch_bounds__(256) void k_combine_fwd_cand_p3(const float* __restrict__ U, const float* __restrict__ V, int C,
                                                             int BT, int N, int pmax, const int* __restrict__ neg_slot,
                                                             __bf16* __restrict__ Z1p /* plane 0, candidate rows */, long long ps) {
    const int bt = blockIdx.x, NC = N + 1;
    const float4* pu = reinterpret_cast<const float4*>(U + (size_t)bt * C);
    __bf16* po = Z1p + (size_t)bt * NC * C;
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 a = pu[k];
        for (int c0 = 0; c0 < NC; c0 += 4) {
            float4 b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i < NC ? c0 + i : NC - 1;
                b[i] = reinterpret_cast<const float4*>(V + (size_t)cand_vrow(bt, c, BT, N, pmax, neg_slot) * C)[k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i >= NC) break;
                float4 o;
                o.x = act_fwd(a.x + b[i].x, ACT_LEAKY); o.y = act_fwd(a.y + b[i].y, ACT_LEAKY);
                o.z = act_fwd(a.z + b[i].z, ACT_LEAKY); o.w = act_fwd(a.w + b[i].w, ACT_LEAKY);
                st4_planes(po + (size_t)(c0 + i) * C + 4 * k, ps, o);
            }
        }
    }
}

// The same rows as TWO fp16 PLANES x the scale of `rec` (csrc/gemm_h2.hip; the record was filled from max |U| + max |V| by
// cham_h2_scale_absmax): 4 bytes stored per element, no fp32 copy.
__global__ __launch_bounds__(256) void k_combine_fwd_cand_h2(const float* __restrict__ U, const float* __restrict__ V, int C,
                                                             int BT, int N, int pmax, const int* __restrict__ neg_slot,
                                                             _Float16* __restrict__ Z1p /* plane 0, candidate rows */, long long ps,
                                                             const H2Scale* __restrict__ rec, int blocked) {
    const int bt = blockIdx.x, NC = N + 1;
    const float sc = rec->scale;
    const float4* pu = reinterpret_cast<const float4*>(U + (size_t)bt * C);
    _Float16* po = Z1p + (size_t)bt * NC * C;
    const size_t grow0 = (size_t)bt * NC;              // first candidate row of this position
    const unsigned ncb = (unsigned)C >> 5;
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 a = pu[k];
        for (int c0 = 0; c0 < NC; c0 += 4) {
            float4 b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i < NC ? c0 + i : NC - 1;
                b[i] = reinterpret_cast<const float4*>(V + (size_t)cand_vrow(bt, c, BT, N, pmax, neg_slot) * C)[k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i >= NC) break;
                float4 o;
                o.x = act_fwd(a.x + b[i].x, ACT_LEAKY); o.y = act_fwd(a.y + b[i].y, ACT_LEAKY);
                o.z = act_fwd(a.z + b[i].z, ACT_LEAKY); o.w = act_fwd(a.w + b[i].w, ACT_LEAKY);
                // row-major [rows, C], or tile-blocked (common.h h2b_index: four consecutive columns never leave a 32-column block)
                st4_planes_h2(blocked ? Z1p + h2b_index(grow0 + (size_t)(c0 + i), 4u * (unsigned)k, ncb) : po + (size_t)(c0 + i) * C + 4 * k, ps, o, sc);
            }
        }
    }
}

// dU[bt] = dpre[input bt] + sum_c dpre[cand (bt,c)];  dV_in[bt] = dpre[input bt];  dV_pos[bt] = dpre[cand (bt,0)]
// (T = element type of the candidate rows: fp32, or bf16 in the bf16 configuration; the clicked-input rows are fp32 in both)
template <typename T>
__global__ __launch_bounds__(256) void k_combine_bwd_u(const float* __restrict__ dpre_in, const T* __restrict__ dpre_cand, int C, int BT,
                                                       int N, float* __restrict__ dU, float* __restrict__ dV) {
    const int bt = blockIdx.x;
    const float4* pin = reinterpret_cast<const float4*>(dpre_in + (size_t)bt * C);
    const T* pc = dpre_cand + (size_t)bt * (N + 1) * C;
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 a = pin[k];
        const float4 p0 = ld4(pc + 4 * k);
        float4 s = make_float4(a.x + p0.x, a.y + p0.y, a.z + p0.z, a.w + p0.w);
        // eight candidate rows in flight per thread (round 6: left to the compiler the loop issued one 8- / 16-byte load at a time - 0.44 ms for the
        // 0.5 GB of the bf16 configuration, 1.1 TB/s, on the main lane's tail); the adds stay in ascending row order: same sums, bit for bit
        for (int c = 1; c <= N; c += 8) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = ld4(pc + (size_t)min(c + u, N) * C + 4 * k);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + u <= N) { s.x += x[u].x; s.y += x[u].y; s.z += x[u].z; s.w += x[u].w; }
        }
        reinterpret_cast<float4*>(dU + (size_t)bt * C)[k] = s;
        reinterpret_cast<float4*>(dV + (size_t)bt * C)[k] = a;
        reinterpret_cast<float4*>(dV + ((size_t)BT + bt) * C)[k] = p0;
    }
}

// The same from the GROUP SUMS the CAR dgrad's epilogue left (cham_gemm_h2_dgrad_gs, csrc/gemm_h2.hip H2Params::gsum): the candidate rows
// of click bt are rows [bt G, (bt + 1) G), G = N + 1 >= 32, of the dgrad's output; 128-row chunk q holds piece k = bt - (128 q) / G of
// that group at gsum[(q * (127 / G + 2) + k) * C + column].  dU[bt] = dpre[input bt] + the group's pieces in chunk order (<= 3 of them
// at the G1 shape): the 1 GB of candidate rows is not read again (only row (bt, 0) for dV_pos).
__global__ __launch_bounds__(256) void k_combine_bwd_u_gs(const float* __restrict__ dpre_in, const float* __restrict__ dpre_cand, int C, int BT,
                                                          int N, const float* __restrict__ gsum, float* __restrict__ dU, float* __restrict__ dV) {
    const int bt = blockIdx.x, G = N + 1, gsk = 127 / G + 2;
    const long r0 = (long)bt * G, r1 = r0 + G - 1;
    const int q0 = (int)(r0 >> 7), q1 = (int)(r1 >> 7);
    const float4* pin = reinterpret_cast<const float4*>(dpre_in + (size_t)bt * C);
    const float4* pc = reinterpret_cast<const float4*>(dpre_cand + (size_t)r0 * C);
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 a = pin[k], p0 = pc[k];
        float4 s = a;
        for (int q = q0; q <= q1; ++q) {
            const int piece = bt - (int)(((long)q << 7) / G);
            const float4 x = reinterpret_cast<const float4*>(gsum + ((size_t)q * gsk + piece) * C)[k];
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        reinterpret_cast<float4*>(dU + (size_t)bt * C)[k] = s;
        reinterpret_cast<float4*>(dV + (size_t)bt * C)[k] = a;
        reinterpret_cast<float4*>(dV + ((size_t)BT + bt) * C)[k] = p0;
    }
}

// dV[2BT + s] = sum over the candidate rows that reference pool slot s - deterministic (rows are summed in ascending position
// order, no float atomics) and without a size assumption (round-1 version: every (slot, row-chunk) workgroup scanned its 64-KB
// chunk of the slot table - 1 GB of L2 reads per step, hot slots serialised, and a fixed-size match list that silently
// truncated the zero-padding slot).
//   1. k_slot_bitmap: one bit per (slot, position): a click holds a pool slot at most once (k_click_select keeps one key per
//      canonical slot), so "which positions reference s" is a BT-bit set; integer atomicOr = order independent.
//   2. k_slot_reduce: workgroup (s, position chunk) walks its bitmap words, lists the set positions in ascending order
//      (popcount prefix), finds the slot's column inside each position's N entries and sums those dpre rows, 8 row loads in
//      flight.  Cold slots (<= SLOT_HOT rows: almost all) are finished by chunk 0 alone straight into dV; hot slots
//      (popularity-sampled negatives: a few slots sit in nearly every click) are split over position chunks -> partials,
//      added in chunk order by k_slot_final.
//   3. the zero-padding slot (s == pmax: a click holds it N - #candidates times) keeps the scan form, in chunks of SLOT_LIST
//      entries so that its match list cannot overflow.
#define SLOT_HOT 512            // rows a single workgroup sums; also the position-chunk size of hot slots (SLOT_HOT / 32 words)
#define SLOT_LIST 2048
__global__ __launch_bounds__(256) void k_slot_bitmap(const int* __restrict__ neg_slot, size_t n, int N, int pmax, int W,
                                                     unsigned* __restrict__ bitmap /*[pmax][W], zeroed*/) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = neg_slot[i];
    if (s < 0 || s >= pmax) return;
    const unsigned p = (unsigned)(i / N);
    atomicOr(bitmap + (size_t)s * W + (p >> 5), 1u << (p & 31));
}

__device__ __forceinline__ int block_sum_int(int v, int* red /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void k_slot_reduce(const T* __restrict__ cand /* candidate rows of dpre */, int C, int BT, int N,
                                                     const int* __restrict__ neg_slot, const unsigned* __restrict__ bitmap, int W,
                                                     int nslots, float* __restrict__ partial /*[nchunk][nslots][C]*/,
                                                     float* __restrict__ dV) {
    __shared__ int list[SLOT_HOT];
    __shared__ int red[4];
    const int s = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned* bm = bitmap + (size_t)s * W;
    int c = 0;
    for (int w = tid; w < W; w += 256) c += __popc(bm[w]);
    const int total = block_sum_int(c, red);
    const bool hot = total > SLOT_HOT;
    if (!hot && chunk != 0) return;
    constexpr int PCW = SLOT_HOT / 32;
    const int w0 = hot ? chunk * PCW : 0, w1 = hot ? min(W, w0 + PCW) : W;
    // ordered list of the set positions in [w0, w1): 256 words per round
    int base = 0;
    for (int t0 = w0; t0 < w1; t0 += 256) {
        const int w = t0 + tid;
        unsigned bits = w < w1 ? bm[w] : 0u;
        const int mycnt = __popc(bits);
        int incl = mycnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        __syncthreads();
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        int off = base + incl - mycnt;
        for (int q = 0; q < wave; ++q) off += red[q];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            if (off < SLOT_HOT) list[off] = w * 32 + b;       // (cannot overflow: <= SLOT_HOT rows by construction)
            ++off;
        }
        base += red[0] + red[1] + red[2] + red[3];
    }
    __syncthreads();
    const int cnt = min(base, SLOT_HOT);
    // position -> candidate row (relative to the first candidate row): find the slot's column among the position's N entries
    for (int j = tid; j < cnt; j += 256) {
        const int p = list[j];
        const int* r = neg_slot + (size_t)p * N;
        int n = 0;
        for (int q = 0; q < N; ++q) n = (r[q] == s) ? q : n;
        list[j] = p * (N + 1) + 1 + n;
    }
    __syncthreads();
    float* out = hot ? partial + ((size_t)chunk * nslots + s) * C : dV + ((size_t)2 * BT + s) * C;
    for (int k = tid; k < C / 4; k += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = ld4(cand + (size_t)list[t + u] * C + 4 * k);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        for (; t < cnt; ++t) {
            const float4 x = ld4(cand + (size_t)list[t] * C + 4 * k);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        reinterpret_cast<float4*>(out)[k] = acc;
    }
}
// hot slots only: dV[2BT + s] = sum of the position chunks' partials, in chunk order
__global__ __launch_bounds__(256) void k_slot_final(const float* __restrict__ partial, int C, int nslots, int nchunk, int BT,
                                                    const unsigned* __restrict__ bitmap, int W, float* __restrict__ dV) {
    __shared__ int red[4];
    const int s = blockIdx.x;
    const unsigned* bm = bitmap + (size_t)s * W;
    int c = 0;
    for (int w = threadIdx.x; w < W; w += 256) c += __popc(bm[w]);
    if (block_sum_int(c, red) <= SLOT_HOT) return;
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ch = 0; ch < nchunk; ++ch) {
            const float4 x = reinterpret_cast<const float4*>(partial + ((size_t)ch * nslots + s) * C)[k];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        reinterpret_cast<float4*>(dV + ((size_t)2 * BT + s) * C)[k] = acc;
    }
}

// The zero-padding slot: rows of one chunk of the slot table that hold `slot`, compacted IN ORDER into an LDS list (ballot-free
// per-thread bit masks + popcount prefix) and summed; chunk_len <= SLOT_LIST, so the list holds every match.
template <typename T>
__global__ __launch_bounds__(256) void k_combine_bwd_slots_partial(const T* __restrict__ cand /* candidate rows of dpre */, int C, int BT, int N,
                                                                   const int* __restrict__ neg_slot, int slot,
                                                                   float* __restrict__ partial /*[nchunk][C]*/,
                                                                   int chunk_len /* multiple of 1024, <= SLOT_LIST */) {
    __shared__ int list[SLOT_LIST];
    __shared__ int wave_tot[4];
    const int s = slot, chunk = blockIdx.x;
    const size_t n = (size_t)BT * N;
    const size_t i0 = (size_t)chunk * chunk_len;
    const size_t i1 = i0 + chunk_len < n ? i0 + chunk_len : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each thread owns a contiguous segment of chunk_len/256 (<= 8) entries: count, block prefix, then fill in order
    const int SEG = chunk_len / 256;
    const size_t sb = i0 + (size_t)threadIdx.x * SEG;
    unsigned mbits = 0u;
    for (int e = 0; e < SEG; ++e)
        if (sb + e < i1 && neg_slot[sb + e] == s) mbits |= 1u << e;
    const int mycnt = __popc(mbits);
    int incl = mycnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = incl - mycnt;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    const int cnt = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];        // <= chunk_len <= SLOT_LIST
    while (mbits) {
        const int bit = __ffs((int)mbits) - 1;
        mbits &= mbits - 1;
        list[off++] = threadIdx.x * SEG + bit;
    }
    __syncthreads();
    const int nk = C / 4;
    for (int k = threadIdx.x; k < nk; k += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t i = i0 + list[t + u];
                const size_t row = (i / N) * (N + 1) + 1 + (i % N);
                x[u] = ld4(cand + row * C + 4 * k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        for (; t < cnt; ++t) {
            const size_t i = i0 + list[t];
            const size_t row = (i / N) * (N + 1) + 1 + (i % N);
            const float4 x = ld4(cand + row * C + 4 * k);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        reinterpret_cast<float4*>(partial + (size_t)chunk * C)[k] = acc;
    }
}
__global__ __launch_bounds__(256) void k_combine_bwd_pad_final(const float* __restrict__ partial, int C, int nchunk,
                                                               float* __restrict__ dv_row) {
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // ONE workgroup adds the ~120 chunk partials of the pad slot: sixteen loads in flight per thread (round 6: one at a time this was a
        // chain of ~120 dependent L2 / HBM latencies - 0.11-0.14 ms on the critical chain of the backward tail); ascending chunk order kept
        for (int c = 0; c < nchunk; c += 16) {
            float4 x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = reinterpret_cast<const float4*>(partial + (size_t)min(c + u, nchunk - 1) * C)[k];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c + u < nchunk) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
        reinterpret_cast<float4*>(dv_row)[k] = acc;
    }
}

// in place: dM[row] <- dM[row] * pred[bt] * (1 - Z2c[row]^2)   (gradient w.r.t. the CAR tanh pre-activation)
// dpred_pre[bt]    = (sum_c dM[row] * Z2c[row]) * (1 - pred[bt]^2) (gradient w.r.t. the FC2 tanh pre-activation)
template <typename T>
__global__ __launch_bounds__(256) void k_mulpred_bwd(T* __restrict__ dM, const T* __restrict__ Z2c,
                                                     const float* __restrict__ pred, int C, int N, float* __restrict__ dpred_pre) {
    const int bt = blockIdx.x;
    const float4* pp = reinterpret_cast<const float4*>(pred + (size_t)bt * C);
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 p = pp[k];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c <= N; ++c) {
            const size_t off = ((size_t)bt * (N + 1) + c) * C + 4 * k;
            const float4 g = ld4(dM + off);
            const float4 z = ld4(Z2c + off);
            acc.x += g.x * z.x; acc.y += g.y * z.y; acc.z += g.z * z.z; acc.w += g.w * z.w;
            float4 o;
            o.x = g.x * p.x * (1.f - z.x * z.x); o.y = g.y * p.y * (1.f - z.y * z.y);
            o.z = g.z * p.z * (1.f - z.z * z.z); o.w = g.w * p.w * (1.f - z.w * z.w);
            st4(dM + off, o);
        }
        float4 o;
        o.x = acc.x * (1.f - p.x * p.x); o.y = acc.y * (1.f - p.y * p.y);
        o.z = acc.z * (1.f - p.z * p.z); o.w = acc.w * (1.f - p.w * p.w);
        reinterpret_cast<float4*>(dpred_pre + (size_t)bt * C)[k] = o;
    }
}

// The plane-resident form (csrc/gemm_p3.hip): reads dM (the scorer layer-1 dgrad, fp32), writes the gradient at the CAR tanh as three
// bf16 planes - the operand of the CAR dgrad and of the W2 weight gradient - and, while the values are in registers, the position's
// share of the b2 bias gradient (col_part[bt] = sum_c of its rows; one column sum over BT rows replaces one over BT * (1 + N)).
__global__ __launch_bounds__(256) void k_mulpred_bwd_p3(const float* __restrict__ dM, const float* __restrict__ Z2c,
                                                        const float* __restrict__ pred, int C, int N, float* __restrict__ dpred_pre,
                                                        __bf16* __restrict__ dZ2p, long long ps, float* __restrict__ col_part) {
    const int bt = blockIdx.x;
    const float4* pp = reinterpret_cast<const float4*>(pred + (size_t)bt * C);
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 p = pp[k];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c <= N; ++c) {
            const size_t off = ((size_t)bt * (N + 1) + c) * C + 4 * k;
            const float4 g = ld4(dM + off);
            const float4 z = ld4(Z2c + off);
            acc.x += g.x * z.x; acc.y += g.y * z.y; acc.z += g.z * z.z; acc.w += g.w * z.w;
            float4 o;
            o.x = g.x * p.x * (1.f - z.x * z.x); o.y = g.y * p.y * (1.f - z.y * z.y);
            o.z = g.z * p.z * (1.f - z.z * z.z); o.w = g.w * p.w * (1.f - z.w * z.w);
            cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
            st4_planes(dZ2p + off, ps, o);
        }
        float4 o;
        o.x = acc.x * (1.f - p.x * p.x); o.y = acc.y * (1.f - p.y * p.y);
        o.z = acc.z * (1.f - p.z * p.z); o.w = acc.w * (1.f - p.w * p.w);
        reinterpret_cast<float4*>(dpred_pre + (size_t)bt * C)[k] = o;
        if (col_part) reinterpret_cast<float4*>(col_part + (size_t)bt * C)[k] = cs;
    }
}

// The two-fp16-plane form (csrc/gemm_h2.hip): as k_mulpred_bwd_p3, the gradient at the CAR tanh stored as (h, l) planes x the scale of
// `rec` (a bound of max |dM|: |pred| <= 1 and 1 - z^2 <= 1 only shrink it).
__global__ __launch_bounds__(256) void k_mulpred_bwd_h2(const float* __restrict__ dM, const float* __restrict__ Z2c,
                                                        const float* __restrict__ pred, int C, int N, float* __restrict__ dpred_pre,
                                                        _Float16* __restrict__ dZ2p, long long ps, float* __restrict__ col_part,
                                                        const H2Scale* __restrict__ rec) {
    const int bt = blockIdx.x;
    const float sc = rec->scale;
    const float4* pp = reinterpret_cast<const float4*>(pred + (size_t)bt * C);
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 p = pp[k];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c <= N; ++c) {
            const size_t off = ((size_t)bt * (N + 1) + c) * C + 4 * k;
            const float4 g = ld4(dM + off);
            const float4 z = ld4(Z2c + off);
            acc.x += g.x * z.x; acc.y += g.y * z.y; acc.z += g.z * z.z; acc.w += g.w * z.w;
            float4 o;
            o.x = g.x * p.x * (1.f - z.x * z.x); o.y = g.y * p.y * (1.f - z.y * z.y);
            o.z = g.z * p.z * (1.f - z.z * z.z); o.w = g.w * p.w * (1.f - z.w * z.w);
            cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
            st4_planes_h2(dZ2p + off, ps, o, sc);
        }
        float4 o;
        o.x = acc.x * (1.f - p.x * p.x); o.y = acc.y * (1.f - p.y * p.y);
        o.z = acc.z * (1.f - p.z * p.z); o.w = acc.w * (1.f - p.w * p.w);
        reinterpret_cast<float4*>(dpred_pre + (size_t)bt * C)[k] = o;
        if (col_part) reinterpret_cast<float4*>(col_part + (size_t)bt * C)[k] = cs;
    }
}

// bf16 configuration: Mc[row] = bf16(Z2c[row] * pred[bt]) - the `cand (.) pred` product of nar_model.py:478-495 as the bf16 operand
// of the scorer's first layer and of its weight gradient (the fp32 path fuses it into the GEMM's staging as a row scale)
__global__ __launch_bounds__(256) void k_mul_rows_b16(const __bf16* __restrict__ Z2c, const float* __restrict__ pred, int C, int NC,
                                                      __bf16* __restrict__ Mc) {
    const int bt = blockIdx.x;
    const float4* pp = reinterpret_cast<const float4*>(pred + (size_t)bt * C);
    for (int k = threadIdx.x; k < C / 4; k += 256) {
        const float4 p = pp[k];
        for (int c0 = 0; c0 < NC; c0 += 4) {
            float4 z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = ld4(Z2c + ((size_t)bt * NC + min(c0 + i, NC - 1)) * C + 4 * k);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i >= NC) break;
                st4(Mc + ((size_t)bt * NC + c0 + i) * C + 4 * k, make_float4(z[i].x * p.x, z[i].y * p.y, z[i].z * p.z, z[i].w * p.w));
            }
        }
    }
}

// last scorer layer (K3 -> 1) + softmax(logits/tau) + masked NLL.  One wave per click (b,t); lanes over the
// 1+N candidates; max / sum by wavefront shuffles.
template <int K3, typename T>
__global__ __launch_bounds__(256) void k_score_softmax_fwd(const T* __restrict__ S3, const float* __restrict__ w4,
                                                           const float* __restrict__ b4, int BT, int N, float inv_tau,
                                                           const unsigned char* __restrict__ mask,
                                                           float* __restrict__ logits, float* __restrict__ probs,
                                                           float* __restrict__ nll, float nov_factor,
                                                           const int64_t* __restrict__ neg_ids, const float* __restrict__ pop_norm,
                                                           float* __restrict__ nov_aux /*[BT,3]*/, float inv_log2_base) {
    const int lane = threadIdx.x & 63, bt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= BT) return;
    const int NC = N + 1;
    float w[K3];
#pragma unroll
    for (int k = 0; k < K3; ++k) w[k] = w4[k];
    const float bias = b4[0];
    float mx = -INFINITY;
    for (int c = lane; c < NC; c += 64) {
        const T* r = S3 + ((size_t)bt * NC + c) * K3;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K3 / 4; ++k) { const float4 x = ld4(r + 4 * k); s += x.x * w[4 * k] + x.y * w[4 * k + 1] + x.z * w[4 * k + 2] + x.w * w[4 * k + 3]; }
        s += bias;
        logits[(size_t)bt * NC + c] = s;
        mx = fmaxf(mx, s * inv_tau);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < NC; c += 64) {
        const float e = expf(logits[(size_t)bt * NC + c] * inv_tau - mx);   // same lane wrote it
        probs[(size_t)bt * NC + c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < NC; c += 64) probs[(size_t)bt * NC + c] *= inv;
    // optional novelty regulariser (nar_model.py:673-683, 544): - factor * sum_n softmax(s_neg / tau)_n * (-log_base pop_norm[neg_n]),
    // base = popularity_smooth_log_base (2 in every shipped script; -log2(p) * inv_log2_base, inv_log2_base exactly 1 for base 2);
    // the softmax over the negatives alone (:517) has its own max / normaliser; {max, sum, q.nov} are kept for the backward
    float novterm = 0.f;
    if (nov_factor > 0.f) {
        float mxn = -INFINITY;
        for (int c = lane; c < NC; c += 64)
            if (c > 0) mxn = fmaxf(mxn, logits[(size_t)bt * NC + c] * inv_tau);
        mxn = wave_max(mxn);
        float sn = 0.f, acc = 0.f;
        for (int c = lane; c < NC; c += 64)
            if (c > 0) {
                const float e = expf(logits[(size_t)bt * NC + c] * inv_tau - mxn);
                sn += e;
                acc += e * (-log2f(pop_norm[neg_ids[(size_t)bt * N + (c - 1)]]) * inv_log2_base);
            }
        sn = wave_sum(sn); acc = wave_sum(acc);
        novterm = acc / sn;
        if (lane == 0) { nov_aux[(size_t)bt * 3] = mxn; nov_aux[(size_t)bt * 3 + 1] = sn; nov_aux[(size_t)bt * 3 + 2] = novterm; }
    }
    if (lane == 0) {
        // -log softmax_0 (log-softmax form of nar_model.py:660; identical wherever the reference is finite)
        const float z0 = logits[(size_t)bt * NC] * inv_tau;
        nll[bt] = mask[bt] ? -((z0 - mx) - logf(sum)) - nov_factor * novterm : 0.f;
    }
}

// ds[row] = (p - [c==0]) * mask / (tau * sum(mask));  dS3pre[row,k] = ds * w4[k] * leaky'(S3[row,k])
template <int K3, typename T>
__global__ __launch_bounds__(256) void k_score_softmax_bwd(const T* __restrict__ S3, const float* __restrict__ w4,
                                                           const float* __restrict__ probs, const unsigned char* __restrict__ mask,
                                                           int BT, int N, float scale /* 1/(tau*sum_mask) */,
                                                           float* __restrict__ ds, T* __restrict__ dS3, float nov_factor,
                                                           const int64_t* __restrict__ neg_ids, const float* __restrict__ pop_norm,
                                                           const float* __restrict__ logits, float inv_tau,
                                                           const float* __restrict__ nov_aux, float inv_log2_base,
                                                           const ChamStepScalars* __restrict__ sc, float tau) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int NC = N + 1;
    if (row >= (size_t)BT * NC) return;
    if (sc) scale = 1.0f / (tau * sc->sum_mask);       // (the by-value entry points compute the same fp32 expression on the host)
    const int bt = (int)(row / NC), c = (int)(row % NC);
    float g = mask[bt] ? (probs[row] - (c == 0 ? 1.f : 0.f)) * scale : 0.f;
    if (nov_factor > 0.f && c > 0 && mask[bt]) {
        // d/ds_c of -factor * sum_n q_n nov_n with q = softmax(s_neg / tau): -factor/tau * q_c * (nov_c - q.nov)
        const float q = expf(logits[row] * inv_tau - nov_aux[(size_t)bt * 3]) / nov_aux[(size_t)bt * 3 + 1];
        const float nov_c = -log2f(pop_norm[neg_ids[(size_t)bt * N + (c - 1)]]) * inv_log2_base;
        g -= nov_factor * scale * q * (nov_c - nov_aux[(size_t)bt * 3 + 2]);
    }
    ds[row] = g;
    const T* r = S3 + row * K3;
    T* o = dS3 + row * K3;
#pragma unroll
    for (int k = 0; k < K3 / 4; ++k) {
        const float4 x = ld4(r + 4 * k);
        float4 y;
        y.x = g * w4[4 * k] * act_bwd_from_out(x.x, ACT_LEAKY);
        y.y = g * w4[4 * k + 1] * act_bwd_from_out(x.y, ACT_LEAKY);
        y.z = g * w4[4 * k + 2] * act_bwd_from_out(x.z, ACT_LEAKY);
        y.w = g * w4[4 * k + 3] * act_bwd_from_out(x.w, ACT_LEAKY);
        st4(o + 4 * k, y);
    }
}


// rank_items_by_predicted_prob (nar_model.py:777-794): tf.nn.top_k over the 1+N candidates = descending stable sort
// (lowest index wins ties).  One wave per click; rank by counting; also emits the rank of the positive (c = 0).
__global__ __launch_bounds__(256) void k_rank_items(const float* __restrict__ probs, const int64_t* __restrict__ label_next,
                                                    const int64_t* __restrict__ neg_ids, const unsigned char* __restrict__ mask,
                                                    int BT, int N, int64_t* __restrict__ pred_ids, float* __restrict__ pred_probs,
                                                    int32_t* __restrict__ label_rank) {
    extern __shared__ float sp[];                  // [4][NC]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, bt = blockIdx.x * 4 + w;
    const int NC = N + 1;
    float* p = sp + (size_t)w * NC;
    if (bt < BT)
        for (int c = lane; c < NC; c += 64) p[c] = probs[(size_t)bt * NC + c];
    __syncthreads();
    if (bt >= BT) return;
    for (int c = lane; c < NC; c += 64) {
        const float pc = p[c];
        int r = 0;
        for (int j = 0; j < NC; ++j) {
            const float pj = p[j];
            r += (pj > pc) || (pj == pc && j < c);
        }
        pred_ids[(size_t)bt * NC + r] = c == 0 ? label_next[bt] : neg_ids[(size_t)bt * N + (c - 1)];
        pred_probs[(size_t)bt * NC + r] = pc;
        if (c == 0) label_rank[bt] = mask[bt] ? r : -1;
    }
}

// ---------------------------------------------------------------------------------------------------
extern "C" int cham_combine_fwd(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot,
                                float* Z1, long row_begin, long row_count, void* stream) {
    if (!U || !V || !neg_slot || !Z1 || (C & 3) || BT <= 0 || N <= 0) return -CHAM_ERR_ARG;
    const long rows = (long)BT + (long)BT * (N + 1);
    if (row_begin < 0 || row_count <= 0 || row_begin + row_count > rows) return -CHAM_ERR_ARG;
    if (row_begin == BT && row_count == (long)BT * (N + 1))         // all candidate rows: one workgroup per position
        hipLaunchKernelGGL(k_combine_fwd_cand<float>, dim3((unsigned)BT), dim3(256), 0, (hipStream_t)stream, U, V, C, BT, N, pmax, neg_slot,
                           Z1 + (size_t)BT * C);
    else
        hipLaunchKernelGGL(k_combine_fwd, dim3((unsigned)row_count), dim3(256), 0, (hipStream_t)stream, U, V, C, BT, N, pmax,
                           neg_slot, Z1, (int)row_begin);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
// bf16 configuration: the BT*(1+N) candidate rows as bf16 (the clicked-input rows stay fp32: cham_combine_fwd on rows [0, BT))
extern "C" int cham_combine_fwd_b16(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot,
                                    void* Z1c, void* stream) {
    if (!U || !V || !neg_slot || !Z1c || (C & 3) || BT <= 0 || N <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_combine_fwd_cand<__bf16>, dim3((unsigned)BT), dim3(256), 0, (hipStream_t)stream, U, V, C, BT, N, pmax, neg_slot,
                       reinterpret_cast<__bf16*>(Z1c));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct SlotWs { size_t bitmap_bytes, partial_bytes, pad_bytes; int W, nchunk, nchunk_pad; };
static SlotWs slot_ws(int C, int BT, int N, int pmax) {
    SlotWs w;
    w.W = (BT + 31) / 32;
    w.nchunk = (w.W + SLOT_HOT / 32 - 1) / (SLOT_HOT / 32);
    w.nchunk_pad = (int)(((size_t)BT * N + SLOT_LIST - 1) / SLOT_LIST);
    w.bitmap_bytes = al256((size_t)pmax * w.W * sizeof(unsigned));
    w.partial_bytes = al256((size_t)w.nchunk * pmax * C * sizeof(float));
    w.pad_bytes = al256((size_t)w.nchunk_pad * C * sizeof(float));
    return w;
}
extern "C" size_t cham_combine_bwd_workspace_bytes(int C, int BT, int N, int pmax) {
    if (C <= 0 || BT <= 0 || N <= 0 || pmax <= 0) return 0;
    const SlotWs w = slot_ws(C, BT, N, pmax);
    return w.bitmap_bytes + w.partial_bytes + w.pad_bytes;
}

template <typename T>
static int combine_bwd_impl(const float* dpre_in, const T* dpre_cand, int C, int BT, int N, int pmax, const int32_t* neg_slot,
                            float* dU, float* dV, float* workspace, size_t workspace_bytes, void* stream, const float* gsum = nullptr) {
    if (!dpre_in || !dpre_cand || !neg_slot || !dU || !dV || !workspace || (C & 3) || BT <= 0 || N <= 0 || pmax <= 0) return -CHAM_ERR_ARG;
    if (workspace_bytes < cham_combine_bwd_workspace_bytes(C, BT, N, pmax) || ((uintptr_t)workspace & 15)) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const SlotWs w = slot_ws(C, BT, N, pmax);
    unsigned* bitmap = reinterpret_cast<unsigned*>(workspace);
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + w.bitmap_bytes);
    float* pad_partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + w.bitmap_bytes + w.partial_bytes);
    const size_t n = (size_t)BT * N;
    if constexpr (sizeof(T) == 4) {
        if (gsum) hipLaunchKernelGGL(k_combine_bwd_u_gs, dim3(BT), dim3(256), 0, st, dpre_in, reinterpret_cast<const float*>(dpre_cand), C, BT, N, gsum, dU, dV);
        else hipLaunchKernelGGL(k_combine_bwd_u<T>, dim3(BT), dim3(256), 0, st, dpre_in, dpre_cand, C, BT, N, dU, dV);
    } else {
        hipLaunchKernelGGL(k_combine_bwd_u<T>, dim3(BT), dim3(256), 0, st, dpre_in, dpre_cand, C, BT, N, dU, dV);
    }
    if (hipMemsetAsync(bitmap, 0, (size_t)pmax * w.W * sizeof(unsigned), st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_slot_bitmap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, neg_slot, n, N, pmax, w.W, bitmap);
    hipLaunchKernelGGL(k_slot_reduce<T>, dim3(pmax, w.nchunk), dim3(256), 0, st, dpre_cand, C, BT, N, neg_slot, bitmap, w.W, pmax, partial, dV);
    hipLaunchKernelGGL(k_slot_final, dim3(pmax), dim3(256), 0, st, partial, C, pmax, w.nchunk, BT, bitmap, w.W, dV);
    hipLaunchKernelGGL(k_combine_bwd_slots_partial<T>, dim3(w.nchunk_pad), dim3(256), 0, st, dpre_cand, C, BT, N, neg_slot, pmax,
                       pad_partial, SLOT_LIST);
    hipLaunchKernelGGL(k_combine_bwd_pad_final, dim3(1), dim3(256), 0, st, pad_partial, C, w.nchunk_pad,
                       dV + ((size_t)2 * BT + pmax) * C);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_combine_bwd(const float* dpre, int C, int BT, int N, int pmax, const int32_t* neg_slot,
                                float* dU, float* dV, float* workspace, size_t workspace_bytes, void* stream) {
    if (!dpre || BT <= 0 || C <= 0) return -CHAM_ERR_ARG;
    return combine_bwd_impl<float>(dpre, dpre + (size_t)BT * C, C, BT, N, pmax, neg_slot, dU, dV, workspace, workspace_bytes, stream);
}
// The per-click sums from the CAR dgrad's group sums (cham_gemm_h2_dgrad_gs with group_rows = N + 1 >= 32 over M = BT (N + 1) rows and C
// columns; groupsum_bytes >= cham_gemm_h2_groupsum_bytes(M, C, N + 1)): same outputs as cham_combine_bwd, the candidate rows are read once (by the
// slot sums) instead of twice.  Another - equally fixed - summation order for dU than cham_combine_bwd's.
extern "C" int cham_combine_bwd_gs(const float* dpre, int C, int BT, int N, int pmax, const int32_t* neg_slot, float* dU, float* dV,
                                   float* workspace, size_t workspace_bytes, const float* groupsum, size_t groupsum_bytes, void* stream) {
    if (!dpre || BT <= 0 || C <= 0 || !groupsum || N + 1 < 32 || ((uintptr_t)groupsum & 15)) return -CHAM_ERR_ARG;
    const size_t rows = (size_t)BT * (N + 1);
    if (groupsum_bytes < (size_t)2 * ((rows + 255) / 256) * (size_t)(127 / (N + 1) + 2) * (size_t)C * sizeof(float)) return -CHAM_ERR_ARG;
    return combine_bwd_impl<float>(dpre, dpre + (size_t)BT * C, C, BT, N, pmax, neg_slot, dU, dV, workspace, workspace_bytes, stream, groupsum);
}
// bf16 configuration: clicked-input rows fp32 [BT, C], candidate rows bf16 [BT*(1+N), C]
extern "C" int cham_combine_bwd_b16(const float* dpre_in, const void* dpre_cand, int C, int BT, int N, int pmax, const int32_t* neg_slot,
                                    float* dU, float* dV, float* workspace, size_t workspace_bytes, void* stream) {
    return combine_bwd_impl<__bf16>(dpre_in, reinterpret_cast<const __bf16*>(dpre_cand), C, BT, N, pmax, neg_slot, dU, dV, workspace,
                                    workspace_bytes, stream);
}

extern "C" int cham_mulpred_bwd(float* dM, const float* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre,
                                void* stream) {
    if (!dM || !Z2c || !pred || !dpred_pre || (C & 3) || BT <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_mulpred_bwd<float>, dim3(BT), dim3(256), 0, (hipStream_t)stream, dM, Z2c, pred, C, N, dpred_pre);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_mulpred_bwd_b16(void* dM, const void* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* stream) {
    if (!dM || !Z2c || !pred || !dpred_pre || (C & 3) || BT <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_mulpred_bwd<__bf16>, dim3(BT), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<__bf16*>(dM),
                       reinterpret_cast<const __bf16*>(Z2c), pred, C, N, dpred_pre);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_combine_fwd_p3(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1p,
                                   long long plane_stride, void* stream) {
    if (!U || !V || !neg_slot || !Z1p || C <= 0 || (C & 3) || BT < 0 || N < 0 || (plane_stride & 3)) return -CHAM_ERR_ARG;
    if (BT == 0) return CHAM_OK;
    hipLaunchKernelGGL(k_combine_fwd_cand_p3, dim3(BT), dim3(256), 0, (hipStream_t)stream, U, V, C, BT, N, pmax, neg_slot,
                       reinterpret_cast<__bf16*>(Z1p), plane_stride);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_mulpred_bwd_p3(const float* dM, const float* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* dZ2p,
                                   long long plane_stride, float* col_part, void* stream) {
    if (!dM || !Z2c || !pred || !dpred_pre || !dZ2p || C <= 0 || (C & 3) || BT < 0 || N < 0 || (plane_stride & 3)) return -CHAM_ERR_ARG;
    if (BT == 0) return CHAM_OK;
    hipLaunchKernelGGL(k_mulpred_bwd_p3, dim3(BT), dim3(256), 0, (hipStream_t)stream, dM, Z2c, pred, C, N, dpred_pre,
                       reinterpret_cast<__bf16*>(dZ2p), plane_stride, col_part);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// blocked != 0: the planes are written TILE-BLOCKED (common.h h2b_index; C % 32 == 0; the caller allocates ceil(rows / 256) row tiles per
// plane, zero-initialised: rows beyond BT * (1 + N) are never written)
extern "C" int cham_combine_fwd_h2b(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1p,
                                    long long plane_stride, const void* scale_rec, int blocked, void* stream) {
    if (!U || !V || !neg_slot || !Z1p || !scale_rec || C <= 0 || (C & 3) || BT < 0 || N < 0 || (plane_stride & 3)) return -CHAM_ERR_ARG;
    if (blocked && ((C & 31) || plane_stride < (long long)(((size_t)BT * (N + 1) + 255) / 256) * (C >> 5) * H2B_BLOCK)) return -CHAM_ERR_ARG;
    if (BT == 0) return CHAM_OK;
    hipLaunchKernelGGL(k_combine_fwd_cand_h2, dim3(BT), dim3(256), 0, (hipStream_t)stream, U, V, C, BT, N, pmax, neg_slot,
                       reinterpret_cast<_Float16*>(Z1p), plane_stride, reinterpret_cast<const H2Scale*>(scale_rec), blocked ? 1 : 0);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_combine_fwd_h2(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1p,
                                   long long plane_stride, const void* scale_rec, void* stream) {
    return cham_combine_fwd_h2b(U, V, C, BT, N, pmax, neg_slot, Z1p, plane_stride, scale_rec, 0, stream);
}

extern "C" int cham_mulpred_bwd_h2(const float* dM, const float* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* dZ2p,
                                   long long plane_stride, float* col_part, const void* scale_rec, void* stream) {
    if (!dM || !Z2c || !pred || !dpred_pre || !dZ2p || !scale_rec || C <= 0 || (C & 3) || BT < 0 || N < 0 || (plane_stride & 3)) return -CHAM_ERR_ARG;
    if (BT == 0) return CHAM_OK;
    hipLaunchKernelGGL(k_mulpred_bwd_h2, dim3(BT), dim3(256), 0, (hipStream_t)stream, dM, Z2c, pred, C, N, dpred_pre,
                       reinterpret_cast<_Float16*>(dZ2p), plane_stride, col_part, reinterpret_cast<const H2Scale*>(scale_rec));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_mul_rows_b16(const void* Z2c, const float* pred, int C, int BT, int NC, void* Mc, void* stream) {
    if (!Z2c || !pred || !Mc || (C & 3) || BT <= 0 || NC <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_mul_rows_b16, dim3(BT), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(Z2c), pred, C, NC,
                       reinterpret_cast<__bf16*>(Mc));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

template <typename T>
static int score_softmax_fwd_impl(const T* S3, int K3, const float* w4, const float* b4, int BT, int N, float tau,
                                  const uint8_t* mask, float* logits, float* probs, float* nll, float novelty_reg_factor,
                                  const int64_t* neg_ids, const float* pop_norm, float* nov_aux, void* stream) {
    if (!S3 || !w4 || !b4 || !mask || !logits || !probs || !nll || K3 != 32 || BT <= 0 || N <= 0) return -CHAM_ERR_ARG;
    if (novelty_reg_factor > 0.f && (!neg_ids || !pop_norm || !nov_aux)) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL((k_score_softmax_fwd<32, T>), dim3((BT + 3) / 4), dim3(256), 0, (hipStream_t)stream, S3, w4, b4, BT, N,
                       1.0f / tau, mask, logits, probs, nll, novelty_reg_factor, neg_ids, pop_norm, nov_aux, g_cham_inv_log2_pop_base);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_score_softmax_fwd(const float* S3, int K3, const float* w4, const float* b4, int BT, int N, float tau,
                                      const uint8_t* mask, float* logits, float* probs, float* nll, float novelty_reg_factor,
                                      const int64_t* neg_ids, const float* pop_norm, float* nov_aux, void* stream) {
    return score_softmax_fwd_impl<float>(S3, K3, w4, b4, BT, N, tau, mask, logits, probs, nll, novelty_reg_factor, neg_ids, pop_norm,
                                         nov_aux, stream);
}
extern "C" int cham_score_softmax_fwd_b16(const void* S3, int K3, const float* w4, const float* b4, int BT, int N, float tau,
                                          const uint8_t* mask, float* logits, float* probs, float* nll, float novelty_reg_factor,
                                          const int64_t* neg_ids, const float* pop_norm, float* nov_aux, void* stream) {
    return score_softmax_fwd_impl<__bf16>(reinterpret_cast<const __bf16*>(S3), K3, w4, b4, BT, N, tau, mask, logits, probs, nll,
                                          novelty_reg_factor, neg_ids, pop_norm, nov_aux, stream);
}

template <typename T>
static int score_softmax_bwd_impl(const T* S3, int K3, const float* w4, const float* probs, const uint8_t* mask,
                                  int BT, int N, float tau, float sum_mask, float* ds, T* dS3, float novelty_reg_factor,
                                  const int64_t* neg_ids, const float* pop_norm, const float* logits, const float* nov_aux,
                                  void* stream, const void* scalars = nullptr) {
    if (!S3 || !w4 || !probs || !mask || !ds || !dS3 || K3 != 32 || BT <= 0 || (!scalars && sum_mask <= 0.f)) return -CHAM_ERR_ARG;
    if (novelty_reg_factor > 0.f && (!neg_ids || !pop_norm || !logits || !nov_aux)) return -CHAM_ERR_ARG;
    const size_t rows = (size_t)BT * (N + 1);
    hipLaunchKernelGGL((k_score_softmax_bwd<32, T>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S3, w4,
                       probs, mask, BT, N, scalars ? 0.f : 1.0f / (tau * sum_mask), ds, dS3, novelty_reg_factor, neg_ids, pop_norm, logits,
                       1.0f / tau, nov_aux, g_cham_inv_log2_pop_base, reinterpret_cast<const ChamStepScalars*>(scalars), tau);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
// sum(mask) from the ChamStepScalars record (device; common.h)
extern "C" int cham_score_softmax_bwd_dev(const float* S3, int K3, const float* w4, const float* probs, const uint8_t* mask,
                                          int BT, int N, float tau, const void* scalars, float* ds, float* dS3, float novelty_reg_factor,
                                          const int64_t* neg_ids, const float* pop_norm, const float* logits, const float* nov_aux,
                                          void* stream) {
    if (!scalars) return -CHAM_ERR_ARG;
    return score_softmax_bwd_impl<float>(S3, K3, w4, probs, mask, BT, N, tau, 0.f, ds, dS3, novelty_reg_factor, neg_ids, pop_norm,
                                         logits, nov_aux, stream, scalars);
}
extern "C" int cham_score_softmax_bwd_b16_dev(const void* S3, int K3, const float* w4, const float* probs, const uint8_t* mask,
                                              int BT, int N, float tau, const void* scalars, float* ds, void* dS3, float novelty_reg_factor,
                                              const int64_t* neg_ids, const float* pop_norm, const float* logits, const float* nov_aux,
                                              void* stream) {
    if (!scalars) return -CHAM_ERR_ARG;
    return score_softmax_bwd_impl<__bf16>(reinterpret_cast<const __bf16*>(S3), K3, w4, probs, mask, BT, N, tau, 0.f, ds,
                                          reinterpret_cast<__bf16*>(dS3), novelty_reg_factor, neg_ids, pop_norm, logits, nov_aux, stream, scalars);
}
extern "C" int cham_score_softmax_bwd(const float* S3, int K3, const float* w4, const float* probs, const uint8_t* mask,
                                      int BT, int N, float tau, float sum_mask, float* ds, float* dS3, float novelty_reg_factor,
                                      const int64_t* neg_ids, const float* pop_norm, const float* logits, const float* nov_aux,
                                      void* stream) {
    return score_softmax_bwd_impl<float>(S3, K3, w4, probs, mask, BT, N, tau, sum_mask, ds, dS3, novelty_reg_factor, neg_ids, pop_norm,
                                         logits, nov_aux, stream);
}
extern "C" int cham_score_softmax_bwd_b16(const void* S3, int K3, const float* w4, const float* probs, const uint8_t* mask,
                                          int BT, int N, float tau, float sum_mask, float* ds, void* dS3, float novelty_reg_factor,
                                          const int64_t* neg_ids, const float* pop_norm, const float* logits, const float* nov_aux,
                                          void* stream) {
    return score_softmax_bwd_impl<__bf16>(reinterpret_cast<const __bf16*>(S3), K3, w4, probs, mask, BT, N, tau, sum_mask, ds,
                                          reinterpret_cast<__bf16*>(dS3), novelty_reg_factor, neg_ids, pop_norm, logits, nov_aux, stream);
}

extern "C" int cham_rank_items(const float* probs, const int64_t* label_next, const int64_t* neg_ids, const uint8_t* mask, int BT,
                               int N, int64_t* pred_ids, float* pred_probs, int32_t* label_rank, void* stream) {
    if (!probs || !label_next || !neg_ids || !mask || !pred_ids || !pred_probs || !label_rank || BT <= 0 || N <= 0 || N > 8191)
        return -CHAM_ERR_ARG;
    const size_t smem = (size_t)4 * (N + 1) * sizeof(float);
    hipLaunchKernelGGL(k_rank_items, dim3((BT + 3) / 4), dim3(256), smem, (hipStream_t)stream, probs, label_next, neg_ids, mask, BT,
                       N, pred_ids, pred_probs, label_rank);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
