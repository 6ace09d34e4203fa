// Feature gather / assemble (+ backward) for the NAR step.
//
// Replaces nar_module/nar/nar_model.py:730-773 (get_features: one-hot / embedding / numeric),
// :921-994 (get_item_features: metadata embeddings, ACE row, trainable item embedding),
// :1055-1131 (recency), :1134-1193 (novelty), :996-1039 (normalize_values / min_max_normalization)
// and :887-907 (scale_center_features, x*gamma+beta).
//
// MI355X design: item features depend only on the item id (positives / negatives are computed against
// the scalar max_event_timestamp, nar_model.py:343,356) and negatives are drawn from a pool of <= 20*N
// ids, so features are assembled ONCE per row of a small "item row set"
//     [ clicked inputs (B*T, per-click reference timestamp) ; positives (B*T) ; pool slots (20*N + 1) ]
// instead of once per [B,T,N] occurrence (409 MB -> ~10 MB of gathers at G1 shape).  The PreCAR layer is
// then applied to these rows (V = X_item * W1_item) and combined per candidate in scorer.hip.
//
// HBM-bound; rows are <= ~16k so one thread per (row, column) with column descriptors is enough:
// consecutive threads walk consecutive columns of one row -> the 1000-byte ACE row and the embedding
// rows are read as contiguous, coalesced segments.
#include "common.h"

enum { COL_ZERO = 0, COL_OHE = 1, COL_EMB = 2, COL_NUM = 3, COL_ACE = 4, COL_ITEMEMB = 5, COL_RECENCY = 6, COL_NOVELTY = 7 };
// descriptor = 5 x int64: kind, feat, sub, dim, param_offset
#define DESC_W 5

float g_cham_ln_elapsed_base = logf(1.3f), g_cham_ln_pop_base = logf(2.0f), g_cham_inv_log2_pop_base = 1.0f;
extern "C" int cham_set_log_bases(float elapsed_days_smooth_log_base, float popularity_smooth_log_base) {
    if (!(elapsed_days_smooth_log_base > 0.f) || elapsed_days_smooth_log_base == 1.f || !(popularity_smooth_log_base > 0.f) ||
        popularity_smooth_log_base == 1.f)
        return -CHAM_ERR_ARG;
    g_cham_ln_elapsed_base = logf(elapsed_days_smooth_log_base);
    g_cham_ln_pop_base = logf(popularity_smooth_log_base);
    g_cham_inv_log2_pop_base = popularity_smooth_log_base == 2.0f ? 1.0f : (float)(1.0 / log2((double)popularity_smooth_log_base));
    return CHAM_OK;
}

__device__ __forceinline__ float recency_raw(int64_t ref_ts, int64_t created, float ln_base) {
    // nar_model.py:1055-1060: int64 -> float32 BEFORE the subtraction; then log_{1.3}(1+x) (:33-34, 1074)
    const float d = ((float)ref_ts - (float)created) / 86400000.0f;
    return logf(fmaxf(d, 0.f) + 1.0f) / ln_base;
}
__device__ __forceinline__ float novelty_raw(float pop_norm, float ln_base) {
    return -(logf(pop_norm) / ln_base);      // nar_model.py:1147-1148
}
// a numerical article-metadata column (nar_model.py:755-757 'numerical' -> expand_dims): integer-valued features are stored as such in
// the int64 metadata table, float-valued ones as their float32 bit pattern (descriptor sub-field 1; nar/layout.py)
__device__ __forceinline__ float meta_num(int64_t m, int is_float_bits) {
    return is_float_bits ? __int_as_float((int)m) : (float)m;
}
__device__ __forceinline__ float norm_apply(float x, const float* st) {
    // st = {mean, sd, zmin, zmax}  (nar_model.py:1031-1037, 1007-1008)
    const float z = (x - st[0]) / st[1];
    const float scaled = (z - st[2] + 1e-24f) / fmaxf(st[3] - st[2], 2e-24f);
    return scaled * 2.0f - 1.0f;
}

// ---------------------------------------------------------------------------------------------------
// user-context rows  (nar_model.py:315-317)
__global__ __launch_bounds__(256) void k_ctx_assemble(const int64_t* __restrict__ cat, const float* __restrict__ num, int R,
                                                      const int64_t* __restrict__ desc, int F,
                                                      const float* __restrict__ params, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ xraw, float* __restrict__ xs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * F) return;
    const int r = (int)(i / F), c = (int)(i % F);
    const int64_t* d = desc + (size_t)c * DESC_W;
    const int kind = (int)d[0], feat = (int)d[1], sub = (int)d[2], dim = (int)d[3];
    float v = 0.f;
    if (kind == COL_OHE) v = (cat[(size_t)feat * R + r] == sub) ? 1.f : 0.f;
    else if (kind == COL_EMB) v = params[d[4] + cat[(size_t)feat * R + r] * dim + sub];
    else if (kind == COL_NUM) v = num[(size_t)feat * R + r];
    xraw[i] = v;
    xs[i] = v * gamma[c] + beta[c];
}

// raw (un-normalised) recency / novelty per item row
__global__ __launch_bounds__(256) void k_item_dynamic_raw(const int64_t* __restrict__ ids, const int64_t* __restrict__ ref_ts, int R,
                                                          const int64_t* __restrict__ created, const float* __restrict__ pop_norm,
                                                          float* __restrict__ rec_raw, float* __restrict__ nov_raw, float ln_e, float ln_p) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int64_t id = ids[r];
    rec_raw[r] = recency_raw(ref_ts[r], created[id], ln_e);
    nov_raw[r] = novelty_raw(pop_norm[id], ln_p);
}

// same, for the "last N recent clicks" used as normalisation population (nar_model.py:1066-1071, 1156-1158)
__global__ __launch_bounds__(256) void k_last_dynamic_raw(const int64_t* __restrict__ last_ids, int n, int64_t max_ts,
                                                          const int64_t* __restrict__ created, const float* __restrict__ pop_norm,
                                                          float* __restrict__ rec_raw, float* __restrict__ nov_raw, float ln_e, float ln_p) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int64_t id = last_ids[r];
    rec_raw[r] = recency_raw(max_ts, created[id], ln_e);
    nov_raw[r] = novelty_raw(pop_norm[id], ln_p);
}

// weighted population stats -> {mean, sd, zmin, zmax}; single workgroup, fixed reduction tree.
// (tf.nn.moments population variance, sd = sqrt(var + 1e-24); nar_model.py:1014-1025, 1001-1002)
__global__ __launch_bounds__(1024) void k_norm_stats(const float* __restrict__ vals, const float* __restrict__ wts, int n,
                                                     float* __restrict__ out, int n_copies) {
    __shared__ float red[16];
    float sw = 0.f, swx = 0.f, mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float w = wts ? wts[i] : 1.f;
        if (w > 0.f) { const float x = vals[i]; sw += w; swx += w * x; mn = fminf(mn, x); mx = fmaxf(mx, x); }
    }
    sw = block_sum(sw, red);
    swx = block_sum(swx, red);
    const float mean = swx / sw;
    float sv = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float w = wts ? wts[i] : 1.f;
        if (w > 0.f) { const float d = vals[i] - mean; sv += w * d * d; }
    }
    sv = block_sum(sv, red);
    mn = wave_min(mn); mx = wave_max(mx);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ float rmn[16], rmx[16];
    if (lane == 0) { rmn[w] = mn; rmx[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) { mn = fminf(mn, rmn[i]); mx = fmaxf(mx, rmx[i]); }
        mn = fminf(mn, rmn[0]); mx = fmaxf(mx, rmx[0]);
        const float sd = sqrtf(sv / sw + 1e-24f);
        for (int c = 0; c < n_copies; ++c) {
            out[c * 8 + 0] = mean; out[c * 8 + 1] = sd;
            out[c * 8 + 2] = (mn - mean) / sd; out[c * 8 + 3] = (mx - mean) / sd;
        }
    }
}

// item rows: concat(metadata feats, ACE, item embedding, recency, novelty) * gamma + beta  (nar_model.py:921-994)
// stats: [3 groups][8] = {rec mean, sd, zmin, zmax, nov mean, sd, zmin, zmax}; group by row range.
__global__ __launch_bounds__(256) void k_item_assemble(const int64_t* __restrict__ ids, int R, int g1_begin, int g2_begin,
                                                       const int64_t* __restrict__ meta_cat, int n_items,
                                                       const float* __restrict__ ace, int ld_ace,
                                                       const float* __restrict__ rec_raw, const float* __restrict__ nov_raw,
                                                       const float* __restrict__ stats,
                                                       const int64_t* __restrict__ desc, int F,
                                                       const float* __restrict__ params, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       float* __restrict__ xraw, float* __restrict__ xs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * F) return;
    const int r = (int)(i / F), c = (int)(i % F);
    const int64_t* d = desc + (size_t)c * DESC_W;
    const int kind = (int)d[0], feat = (int)d[1], sub = (int)d[2], dim = (int)d[3];
    const int64_t id = ids[r];
    const int g = r < g1_begin ? 0 : (r < g2_begin ? 1 : 2);
    float v = 0.f;
    if (kind == COL_OHE) v = (meta_cat[(size_t)feat * n_items + id] == sub) ? 1.f : 0.f;
    else if (kind == COL_EMB) v = params[d[4] + meta_cat[(size_t)feat * n_items + id] * dim + sub];
    else if (kind == COL_NUM) v = meta_num(meta_cat[(size_t)feat * n_items + id], sub);
    else if (kind == COL_ACE) v = ace[(size_t)id * ld_ace + sub];
    else if (kind == COL_ITEMEMB) v = params[d[4] + id * dim + sub];
    else if (kind == COL_RECENCY) v = norm_apply(rec_raw[r], stats + g * 8);
    else if (kind == COL_NOVELTY) v = norm_apply(nov_raw[r], stats + g * 8 + 4);
    xraw[i] = v;
    xs[i] = v * gamma[c] + beta[c];
}

// LDS-staged form of the item-row gather (north_star: "coalesced HBM reads of the article-content-embedding table into LDS tiles"):
// a row of the item matrix is a few CONTIGUOUS source segments - the article's ACE row, its trainable embedding row, one row per
// metadata embedding - plus a handful of scalar columns (one-hot bits, recency, novelty).  One workgroup stages ITEM_TILE rows:
// each wave copies whole segments of its rows into an LDS image with lane-contiguous loads (a wave reads 256 consecutive bytes
// of a table row per instruction; no per-element descriptor decode), then all threads stream the finished rows out with 16-byte
// stores - raw and gamma/beta-scaled - in one pass.  Replaces the one-thread-per-element k_item_assemble (5 descriptor words +
// 1 id per 4 output bytes) on the large-catalog path (5 M articles x 200 negatives: 160 k item rows x 2.2 KB per step).
#define ITEM_TILE 8
#define SEG_W 6          // segment = 6 x int64: kind (0 ACE, 1 item embedding, 2 metadata embedding), dst column, length, source offset, pitch, feat
__global__ __launch_bounds__(256) void k_item_assemble_lds(const int64_t* __restrict__ ids, int R, int g1_begin, int g2_begin,
                                                           const int64_t* __restrict__ meta_cat, int n_items,
                                                           const float* __restrict__ ace, int ld_ace,
                                                           const float* __restrict__ rec_raw, const float* __restrict__ nov_raw,
                                                           const float* __restrict__ stats, const int64_t* __restrict__ desc, int F,
                                                           const int64_t* __restrict__ segs, int n_segs,
                                                           const int32_t* __restrict__ singles, int n_singles,
                                                           const float* __restrict__ params, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ xraw, float* __restrict__ xs) {
    extern __shared__ __attribute__((aligned(16))) float tile[];            // [ITEM_TILE][F]
    const int r0 = blockIdx.x * ITEM_TILE, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int rr = w; rr < ITEM_TILE; rr += 4) {
        const int r = r0 + rr;
        if (r >= R) break;
        const int64_t id = ids[r];
        float* row = tile + rr * F;
        for (int sgi = 0; sgi < n_segs; ++sgi) {
            const int64_t* sg = segs + sgi * SEG_W;
            const int kind = (int)sg[0], dst = (int)sg[1], len = (int)sg[2];
            const float* src;
            if (kind == 0) src = ace + (size_t)id * ld_ace;
            else if (kind == 1) src = params + sg[3] + id * sg[4];
            else src = params + sg[3] + meta_cat[(size_t)sg[5] * n_items + id] * sg[4];
            for (int c = lane; c < len; c += 64) row[dst + c] = src[c];
        }
        const int g = r < g1_begin ? 0 : (r < g2_begin ? 1 : 2);
        for (int k = lane; k < n_singles; k += 64) {
            const int c = singles[k];
            const int64_t* d = desc + (size_t)c * DESC_W;
            const int kind = (int)d[0], feat = (int)d[1], sub = (int)d[2];
            float v = 0.f;
            if (kind == COL_OHE) v = (meta_cat[(size_t)feat * n_items + id] == sub) ? 1.f : 0.f;
            else if (kind == COL_NUM) v = meta_num(meta_cat[(size_t)feat * n_items + id], sub);
            else if (kind == COL_RECENCY) v = norm_apply(rec_raw[r], stats + g * 8);
            else if (kind == COL_NOVELTY) v = norm_apply(nov_raw[r], stats + g * 8 + 4);
            row[c] = v;
        }
    }
    __syncthreads();
    const int rows = min(ITEM_TILE, R - r0), f4 = F / 4;
    for (int i = threadIdx.x; i < rows * f4; i += 256) {
        const int rr = i / f4, c4 = i % f4;
        const float4 v = reinterpret_cast<const float4*>(tile + rr * F)[c4];
        const float4 gm = reinterpret_cast<const float4*>(gamma)[c4], bt = reinterpret_cast<const float4*>(beta)[c4];
        const size_t o = ((size_t)(r0 + rr) * F) / 4 + c4;
        reinterpret_cast<float4*>(xraw)[o] = v;
        reinterpret_cast<float4*>(xs)[o] = make_float4(v.x * gm.x + bt.x, v.y * gm.y + bt.y, v.z * gm.z + bt.z, v.w * gm.w + bt.w);
    }
}

// ---------------------------------------------------------------------------------------------------
// backward: dgamma / dbeta column sums (deterministic, one workgroup per column) ...
__global__ __launch_bounds__(256) void k_scale_bwd_cols(const float* __restrict__ dxs, const float* __restrict__ xraw, int R, int F,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float sg = 0.f, sb = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float g = dxs[(size_t)r * F + c];
        sg += g * xraw[(size_t)r * F + c]; sb += g;
    }
    sg = block_sum(sg, red);
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) { dgamma[c] = sg; dbeta[c] = sb; }
}
// ... and the embedding-table gradients (the IndexedSlices of tf.nn.embedding_lookup, nar_model.py:741, 918): the rows of dxs that
// looked up the same table row are SUMMED IN A FIXED ORDER - no float atomics, the step is bit-reproducible (round 1 scattered
// with atomicAdd: duplicate ids within a batch made two runs differ in the last bit, and with them every later step).
//
// (a) small tables (context / article-metadata embeddings: 12 ... ~1000 rows): one workgroup per TABLE row scans the source rows'
//     keys 64 at a time (ballot), 16 waves take interleaved 64-row stripes, matching rows are added in ascending order per wave,
//     the 16 wave sums in wave order.
__global__ __launch_bounds__(1024) void k_emb_grad_scan(const float* __restrict__ dxs, int R, int F, int c0, int dim,
                                                        const float* __restrict__ gamma, const int64_t* __restrict__ keysrc,
                                                        const int64_t* __restrict__ ids /* null: key = keysrc[r]; else keysrc[ids[r]] */,
                                                        float* __restrict__ table_grad) {
    __shared__ float part[16][64];
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int sub0 = 0; sub0 < dim; sub0 += 64) {
        const int sub = sub0 + lane;
        const bool cok = sub < dim;
        float acc = 0.f;
        for (int r0 = w * 64; r0 < R; r0 += 1024) {
            const int r = r0 + lane;
            int64_t key = -1;
            if (r < R) key = ids ? keysrc[ids[r]] : keysrc[r];
            unsigned long long m = __ballot(key == row);
            while (m) {                                   // wave-uniform; 4 independent row loads in flight
                int b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    b[u] = m ? __ffsll((long long)m) - 1 : -1;
                    m &= m - 1;
                }
                float x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = (b[u] >= 0 && cok) ? dxs[(size_t)(r0 + b[u]) * F + c0 + sub] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += x[u];
            }
        }
        part[w][lane] = acc;
        __syncthreads();
        if (w == 0 && cok) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += part[i][lane];
            table_grad[(size_t)row * dim + sub] = t * gamma[c0 + sub];
        }
        __syncthreads();
    }
}

// (b) the trainable item-embedding table (46 k ... 5 M rows, <= a few 10 k of them touched per step): the item rows are ranked by
//     (id, row) - rank = number of smaller keys, counted against LDS tiles of the key list: O(R^2 / chip) integer compares, a few
//     microseconds for R ~ 10^4 and independent of the gradients, so it runs in the forward pass - which makes equal ids
//     contiguous in `perm`; one workgroup per segment adds its rows (4 waves take every 4th member, ascending; wave sums in
//     wave order) and STORES the table row: distinct segments = distinct table rows.
#define RANK_KEY(id, r) (((unsigned long long)(id) << 20) | (unsigned long long)(r))
__global__ __launch_bounds__(256) void k_rank_keys(const int64_t* __restrict__ ids, int R, int span, int* __restrict__ rank) {
    __shared__ unsigned long long tile[1024];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long key = r < R ? RANK_KEY(ids[r], r) : ~0ull;
    const int j0 = blockIdx.y * span, j1 = min(R, j0 + span);
    int cnt = 0;
    for (int t0 = j0; t0 < j1; t0 += 1024) {
        __syncthreads();
        for (int q = threadIdx.x; q < 1024; q += 256) {
            const int j = t0 + q;
            tile[q] = j < j1 ? RANK_KEY(ids[j], j) : ~0ull;
        }
        __syncthreads();
        const int n = min(1024, j1 - t0);
        for (int q = 0; q < n; ++q) cnt += tile[q] < key;
    }
    if (r < R && cnt) atomicAdd(rank + r, cnt);           // integer: order independent
}
__global__ __launch_bounds__(256) void k_perm_from_rank(const int* __restrict__ rank, int R, int* __restrict__ perm) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) perm[rank[r]] = r;
}
// Segment table of the sorted row list (depends on the ids only: built in the forward pass next to `perm`), int32 words:
//   seg[0] = number of segments, seg[1] = number of LONG segments (> 32 rows), seg[2] = number of work items of the long segments
//   (chunks of EMB_CHUNK rows); seg[4 ..] = first sorted position of every segment (+ R as the end marker); then the indices of the
//   long segments; then the first work item of every long segment (+ end marker); then the partial sums of the work items (as floats).
// One workgroup, block scans, ascending order everywhere.
// (Round 2 launched one workgroup per SORTED POSITION, twice, and let the non-heads exit, and a popular article's rows - Zipf ids: the
// top item holds ~12 % of a micro-batch's 23 k rows at the config-5 size - were summed by ONE workgroup in two stripes of dependent
// index -> row loads: 1.08 ms for 71 MB.  Now: one wave per short segment; long segments cut into chunks of 128 rows, one workgroup
// per chunk, the chunk sums added in chunk order.  profiles/r03_notes.md)
#define EMB_CHUNK 128
__host__ __device__ inline int seg_nl(int R) { return R / 33 + 2; }                       // capacity of the long-segment list
__host__ __device__ inline int seg_nw(int R) { return R / 33 + R / EMB_CHUNK + 4; }       // capacity of the work list
__host__ __device__ inline size_t seg_off_long(int R) { return (size_t)R + 6; }
__host__ __device__ inline size_t seg_off_work(int R) { return seg_off_long(R) + seg_nl(R); }
__host__ __device__ inline size_t seg_off_partial(int R) { return (seg_off_work(R) + seg_nl(R) + 1 + 3) & ~(size_t)3; }
__device__ __forceinline__ int block_excl_scan_1024(int v, int* wsum /*[16]*/, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int q = 0; q < 16; ++q) { if (q < w) off += wsum[q]; tot += wsum[q]; }
    *total = tot;
    return off + incl - v;
}
__global__ __launch_bounds__(1024) void k_seg_table(const int64_t* __restrict__ ids, const int* __restrict__ perm, int R, int* __restrict__ seg) {
    __shared__ int wsum[16];
    int* seg_start = seg + 4;
    int* long_list = seg + seg_off_long(R);
    int* work_first = seg + seg_off_work(R);
    const int tid = threadIdx.x;
    const int CH = (R + 1023) / 1024, i0 = min(R, tid * CH), i1 = min(R, i0 + CH);
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += (i == 0 || ids[perm[i]] != ids[perm[i - 1]]) ? 1 : 0;
    int n_seg;
    int k = block_excl_scan_1024(cnt, wsum, &n_seg);
    for (int i = i0; i < i1; ++i)
        if (i == 0 || ids[perm[i]] != ids[perm[i - 1]]) seg_start[k++] = i;
    if (tid == 0) { seg_start[n_seg] = R; seg[0] = n_seg; }
    __threadfence_block();
    __syncthreads();
    const int CH2 = (n_seg + 1023) / 1024, k0 = min(n_seg, tid * CH2), k1 = min(n_seg, k0 + CH2);
    int cl = 0, cw = 0;
    for (int q = k0; q < k1; ++q) {
        const int len = seg_start[q + 1] - seg_start[q];
        if (len > 32) { ++cl; cw += (len + EMB_CHUNK - 1) / EMB_CHUNK; }
    }
    int n_long, n_work;
    int o = block_excl_scan_1024(cl, wsum, &n_long);
    int ow = block_excl_scan_1024(cw, wsum, &n_work);
    for (int q = k0; q < k1; ++q) {
        const int len = seg_start[q + 1] - seg_start[q];
        if (len > 32) { long_list[o] = q; work_first[o] = ow; ++o; ow += (len + EMB_CHUNK - 1) / EMB_CHUNK; }
    }
    if (tid == 0) { seg[1] = n_long; seg[2] = n_work; work_first[n_long] = n_work; }
}
// short segments (<= 32 rows: almost all): one WAVE per segment, lanes over the columns, rows in order, 4 loads in flight
__global__ __launch_bounds__(256) void k_emb_grad_short(const float* __restrict__ dxs, int R, int F, int c0, int dim,
                                                        const float* __restrict__ gamma, const int64_t* __restrict__ ids,
                                                        const int* __restrict__ perm, const int* __restrict__ seg, float* __restrict__ table_grad) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= seg[0]) return;
    const int start = seg[4 + k], len = seg[4 + k + 1] - start;
    if (len > 32) return;
    const int* rows = perm + start;
    const int64_t id = ids[rows[0]];
    for (int sub = lane; sub < dim; sub += 64) {
        float a = 0.f;
        int m = 0;
        for (; m + 4 <= len; m += 4) {
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = dxs[(size_t)rows[m + u] * F + c0 + sub];
#pragma unroll
            for (int u = 0; u < 4; ++u) a += x[u];
        }
        for (; m < len; ++m) a += dxs[(size_t)rows[m] * F + c0 + sub];
        table_grad[(size_t)id * dim + sub] = a * gamma[c0 + sub];
    }
}
// long segments, pass 1: work item = EMB_CHUNK consecutive sorted rows of one long segment; thread = column, rows in order, 8 in flight
__global__ __launch_bounds__(256) void k_emb_grad_long_part(const float* __restrict__ dxs, int R, int F, int c0, int dim,
                                                            const int* __restrict__ perm, int* __restrict__ seg) {
    __shared__ int rws[EMB_CHUNK];
    const int wid = blockIdx.x;
    if (wid >= seg[2]) return;
    const int* work_first = seg + seg_off_work(R);
    int lo = 0, hi = seg[1];                       // the long segment q with work_first[q] <= wid < work_first[q + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (work_first[mid] <= wid) lo = mid; else hi = mid; }
    const int k = seg[seg_off_long(R) + lo];
    const int start = seg[4 + k], len = seg[4 + k + 1] - start;
    const int r0 = (wid - work_first[lo]) * EMB_CHUNK, n = min(EMB_CHUNK, len - r0);
    for (int i = threadIdx.x; i < n; i += 256) rws[i] = perm[start + r0 + i];
    __syncthreads();
    float* partial = reinterpret_cast<float*>(seg + seg_off_partial(R)) + (size_t)wid * dim;
    for (int sub = threadIdx.x; sub < dim; sub += 256) {
        float a = 0.f;
        int m = 0;
        for (; m + 8 <= n; m += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = dxs[(size_t)rws[m + u] * F + c0 + sub];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += x[u];
        }
        for (; m < n; ++m) a += dxs[(size_t)rws[m] * F + c0 + sub];
        partial[sub] = a;
    }
}
// pass 2: the chunk sums of a long segment, in chunk order
__global__ __launch_bounds__(256) void k_emb_grad_long_final(int R, int dim, int c0, const float* __restrict__ gamma, const int64_t* __restrict__ ids,
                                                             const int* __restrict__ perm, const int* __restrict__ seg, float* __restrict__ table_grad) {
    const int q = blockIdx.x;
    if (q >= seg[1]) return;
    const int* work_first = seg + seg_off_work(R);
    const int k = seg[seg_off_long(R) + q];
    const int64_t id = ids[perm[seg[4 + k]]];
    const int w0 = work_first[q], w1 = work_first[q + 1];
    const float* partial = reinterpret_cast<const float*>(seg + seg_off_partial(R));
    for (int sub = threadIdx.x; sub < dim; sub += 256) {
        float t = 0.f;
        for (int w = w0; w < w1; ++w) t += partial[(size_t)w * dim + sub];
        table_grad[(size_t)id * dim + sub] = t * gamma[c0 + sub];
    }
}

// occurrence counts of pool slots over the sampled negatives (only used for the empty-buffer first batch,
// where the normalisation population is "this call's non-pad ids WITH repetition", nar_model.py:1078-1084)
__global__ __launch_bounds__(256) void k_slot_weights(const int* __restrict__ neg_slot, size_t n, int pmax,
                                                      const int64_t* __restrict__ pool, float* __restrict__ w /*[pmax+1], zeroed*/) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = neg_slot[i];
    if (s >= 0 && s < pmax && pool[s] != 0) atomicAdd(w + s, 1.0f);   // integer-valued floats: order independent
}
__global__ __launch_bounds__(256) void k_nonzero_weights(const int64_t* __restrict__ ids, int n, float* __restrict__ w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = ids[i] != 0 ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------------------------------
extern "C" int cham_ctx_assemble(const int64_t* cat, const float* num, int R, const int64_t* desc, int F,
                                 const float* params, const float* gamma, const float* beta,
                                 float* xraw, float* xs, void* stream) {
    if (!desc || !params || !gamma || !beta || !xraw || !xs || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    const size_t n = (size_t)R * F;
    hipLaunchKernelGGL(k_ctx_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       cat, num, R, desc, F, params, gamma, beta, xraw, xs);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_item_dynamic_raw(const int64_t* ids, const int64_t* ref_ts, int R, const int64_t* created,
                                     const float* pop_norm, float* rec_raw, float* nov_raw, void* stream) {
    if (!ids || !ref_ts || !created || !pop_norm || !rec_raw || !nov_raw || R <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_item_dynamic_raw, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       ids, ref_ts, R, created, pop_norm, rec_raw, nov_raw, g_cham_ln_elapsed_base, g_cham_ln_pop_base);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// stats from the recent-clicks population (buffer non-empty): same stats for the 3 call groups
extern "C" int cham_norm_stats_from_recent(const int64_t* last_ids, int n_last, int64_t max_ts, const int64_t* created,
                                           const float* pop_norm, float* scratch /*2*n_last*/, float* stats /*[3][8]*/,
                                           void* stream) {
    if (!last_ids || n_last <= 0 || !created || !pop_norm || !scratch || !stats) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_last_dynamic_raw, dim3((n_last + 255) / 256), dim3(256), 0, st, last_ids, n_last, max_ts, created,
                       pop_norm, scratch, scratch + n_last, g_cham_ln_elapsed_base, g_cham_ln_pop_base);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch, (const float*)nullptr, n_last, stats, 3);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch + n_last, (const float*)nullptr, n_last, stats + 4, 3);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// same statistics straight from the device-resident recent-clicks buffer (csrc/state.hip): population = the first
// min(n_prefix, #valid) entries (valid entries are a zero-padded prefix; nar_model.py:1041-1044).  scratch: 3 * n_prefix floats.
extern "C" int cham_norm_stats_from_buffer(const int64_t* buffer_ids, int n_prefix, int64_t max_ts, const int64_t* created,
                                           const float* pop_norm, float* scratch, float* stats /*[3][8]*/, void* stream) {
    if (!buffer_ids || n_prefix <= 0 || !created || !pop_norm || !scratch || !stats) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* w = scratch + 2 * (size_t)n_prefix;
    hipLaunchKernelGGL(k_nonzero_weights, dim3((n_prefix + 255) / 256), dim3(256), 0, st, buffer_ids, n_prefix, w);
    hipLaunchKernelGGL(k_last_dynamic_raw, dim3((n_prefix + 255) / 256), dim3(256), 0, st, buffer_ids, n_prefix, max_ts, created,
                       pop_norm, scratch, scratch + n_prefix, g_cham_ln_elapsed_base, g_cham_ln_pop_base);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch, (const float*)w, n_prefix, stats, 3);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch + n_prefix, (const float*)w, n_prefix, stats + 4, 3);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// stats from the call's own rows (empty buffer = very first batch): one group at a time
extern "C" int cham_norm_stats_from_rows(const float* rec_raw, const float* nov_raw, const float* weights, int n,
                                         float* stats_group /*[8]*/, void* stream) {
    if (!rec_raw || !nov_raw || !weights || n <= 0 || !stats_group) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, rec_raw, weights, n, stats_group, 1);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, nov_raw, weights, n, stats_group + 4, 1);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_row_weights(const int64_t* ids, int n_ids, const int32_t* neg_slot, size_t n_neg, int pmax,
                                const int64_t* pool, float* w_ids, float* w_slots, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (ids && n_ids > 0) hipLaunchKernelGGL(k_nonzero_weights, dim3((n_ids + 255) / 256), dim3(256), 0, st, ids, n_ids, w_ids);
    if (neg_slot && n_neg > 0) {
        if (hipMemsetAsync(w_slots, 0, (size_t)(pmax + 1) * sizeof(float), st) != hipSuccess) return -CHAM_ERR_LAUNCH;
        hipLaunchKernelGGL(k_slot_weights, dim3((unsigned)((n_neg + 255) / 256)), dim3(256), 0, st, neg_slot, n_neg, pmax, pool, w_slots);
    }
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_item_assemble(const int64_t* ids, int R, int g1_begin, int g2_begin, const int64_t* meta_cat, int n_items,
                                  const float* ace, int ld_ace, const float* rec_raw, const float* nov_raw,
                                  const float* stats, const int64_t* desc, int F, const float* params,
                                  const float* gamma, const float* beta, float* xraw, float* xs, void* stream) {
    if (!ids || !desc || !params || !gamma || !beta || !xraw || !xs || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    const size_t n = (size_t)R * F;
    hipLaunchKernelGGL(k_item_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ids, R, g1_begin, g2_begin, meta_cat, n_items, ace, ld_ace, rec_raw, nov_raw, stats, desc, F,
                       params, gamma, beta, xraw, xs);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// item rows through LDS tiles (k_item_assemble_lds): segs [n_segs][6] int64 {kind 0 ACE / 1 item embedding / 2 metadata embedding, dst
// column, length, source offset in `params`, row pitch, metadata feature}, singles [n_singles] = the columns no segment covers.
extern "C" int cham_item_assemble_lds(const int64_t* ids, int R, int g1_begin, int g2_begin, const int64_t* meta_cat, int n_items,
                                      const float* ace, int ld_ace, const float* rec_raw, const float* nov_raw, const float* stats,
                                      const int64_t* desc, int F, const int64_t* segs, int n_segs, const int32_t* singles,
                                      int n_singles, const float* params, const float* gamma, const float* beta, float* xraw,
                                      float* xs, void* stream) {
    if (!ids || !desc || !segs || !params || !gamma || !beta || !xraw || !xs || R <= 0 || F <= 0 || (F & 3) || n_segs < 0 || n_singles < 0)
        return -CHAM_ERR_ARG;
    if (n_singles > 0 && !singles) return -CHAM_ERR_ARG;
    const size_t smem = (size_t)ITEM_TILE * F * sizeof(float);
    if (smem > 64 * 1024) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_item_assemble_lds, dim3((R + ITEM_TILE - 1) / ITEM_TILE), dim3(256), smem, (hipStream_t)stream, ids, R, g1_begin,
                       g2_begin, meta_cat, n_items, ace, ld_ace, rec_raw, nov_raw, stats, desc, F, segs, n_segs, singles, n_singles, params,
                       gamma, beta, xraw, xs);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_feature_bwd(const float* dxs, const float* xraw, int R, int F, float* dgamma, float* dbeta, void* stream) {
    if (!dxs || !xraw || !dgamma || !dbeta || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_scale_bwd_cols, dim3(F), dim3(256), 0, (hipStream_t)stream, dxs, xraw, R, F, dgamma, dbeta);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_emb_grad_scan(const float* dxs, int R, int F, int c0, int dim, const float* gamma, const int64_t* keysrc,
                                  const int64_t* ids, int cardinality, float* table_grad, void* stream) {
    if (!dxs || !gamma || !keysrc || !table_grad || R <= 0 || F <= 0 || c0 < 0 || dim <= 0 || c0 + dim > F || cardinality <= 0)
        return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_emb_grad_scan, dim3(cardinality), dim3(1024), 0, (hipStream_t)stream, dxs, R, F, c0, dim, gamma, keysrc,
                       ids, table_grad);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" size_t cham_group_rows_workspace_bytes(int R) { return R > 0 ? (size_t)R * sizeof(int) : 0; }
extern "C" size_t cham_group_rows_segments_len(int R) {      // int32 words of `seg` (the tail holds the chunk sums of the long segments, <= 512 columns)
    return R > 0 ? seg_off_partial(R) + (size_t)seg_nw(R) * 512 : 0;
}
extern "C" int cham_group_rows(const int64_t* ids, int R, int32_t* perm, int32_t* seg, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ids || !perm || !seg || !workspace || R <= 0 || R >= (1 << 20) || workspace_bytes < cham_group_rows_workspace_bytes(R))
        return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int* rank = reinterpret_cast<int*>(workspace);
    if (hipMemsetAsync(rank, 0, (size_t)R * sizeof(int), st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    int span = ((R + 31) / 32 + 1023) / 1024 * 1024;       // <= 32 key ranges, whole LDS tiles
    if (span < 1024) span = 1024;
    const int ns = (R + span - 1) / span;
    hipLaunchKernelGGL(k_rank_keys, dim3((R + 255) / 256, ns), dim3(256), 0, st, ids, R, span, rank);
    hipLaunchKernelGGL(k_perm_from_rank, dim3((R + 255) / 256), dim3(256), 0, st, rank, R, perm);
    hipLaunchKernelGGL(k_seg_table, dim3(1), dim3(1024), 0, st, ids, perm, R, seg);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_emb_grad_grouped(const float* dxs, int R, int F, int c0, int dim, const float* gamma, const int64_t* ids,
                                     const int32_t* perm, const int32_t* seg, float* table_grad, void* stream) {
    if (!dxs || !gamma || !ids || !perm || !seg || !table_grad || R <= 0 || F <= 0 || c0 < 0 || dim <= 0 || dim > 512 || c0 + dim > F)
        return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_emb_grad_short, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, dxs, R, F, c0, dim, gamma, ids, perm, seg, table_grad);
    hipLaunchKernelGGL(k_emb_grad_long_part, dim3(seg_nw(R)), dim3(256), 0, (hipStream_t)stream, dxs, R, F, c0, dim, perm, const_cast<int32_t*>(seg));
    hipLaunchKernelGGL(k_emb_grad_long_final, dim3(seg_nl(R)), dim3(256), 0, (hipStream_t)stream, R, dim, c0, gamma, ids, perm, seg, table_grad);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// ---------------------------------------------------------------------------------------------------
// Dropout (keep_prob < 1: nar_model.py:338, 352, 368 on the three feature tensors, :418 on FC1, :1331 DropoutWrapper on every
// recurrent layer's output).  tf.nn.dropout of TF 1.12: y = x / keep_prob * mask.  TF's own random streams are not reproducible
// without TF; the mask is defined by the same counter-based generator as the negative sampler (common.h philox_rand32, contract:
// oracle/philox.py): element (session row b [GLOBAL], time step t, column c, site, negative n) is kept iff
//     Philox4x32-10(ctr = (c, t, b, site + 256 n), key = (seed, step))[0] < floor(keep_prob * 2^32)
// - a pure function of the element's coordinates, so it is independent of row shards, valid-position compaction and padding, and
// the backward pass recomputes it instead of storing it.
//   row r of the matrix -> (position index r / group, sub = r % group); sub 0 uses site_first (n = 0), sub > 0 site_rest (n = sub - 1);
//   position index -> (b, t) through pos[] (compacted layouts) or directly ([B, T] layouts); column c -> logical feature column
//   c < col_split ? c : c - col_shift (the item columns of the padded [ctx | item] layout).
__global__ __launch_bounds__(256) void k_dropout(const float* __restrict__ x, float* __restrict__ y, size_t rows, int cols, int ld,
                                                 float keep, uint32_t thr, uint32_t seed, uint32_t step, int site_first, int site_rest,
                                                 int group, const int* __restrict__ pos, int T, int row_begin, int col_split, int col_shift) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (size_t)cols) return;
    const size_t r = i / cols;
    const int c = (int)(i % cols);
    const size_t pi = r / group;
    const int sub = (int)(r % group);
    const int q = pos ? pos[pi] : (int)pi;
    const uint32_t b = (uint32_t)(row_begin + q / T), t = (uint32_t)(q % T);
    const uint32_t site = (uint32_t)(sub == 0 ? site_first : site_rest) + 256u * (uint32_t)(sub == 0 ? 0 : sub - 1);
    const uint32_t cl = (uint32_t)(c < col_split ? c : c - col_shift);
    const bool kept = philox_rand32(cl, t, b, site, seed, step) < thr;
    const float v = x[r * ld + c];
    y[r * ld + c] = kept ? v / keep : 0.f;
}
extern "C" int cham_dropout(const float* x, float* y, long rows, int cols, int ld, float keep_prob, uint32_t seed, uint32_t step,
                            int site_first, int site_rest, int group, const int32_t* pos, int T, int row_begin, int col_split,
                            int col_shift, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ld < cols || group <= 0 || T <= 0 || !(keep_prob > 0.f) || keep_prob >= 1.f) return -CHAM_ERR_ARG;
    const uint32_t thr = (uint32_t)((double)keep_prob * 4294967296.0);
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(k_dropout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)rows, cols, ld,
                       keep_prob, thr, seed, step, site_first, site_rest, group, pos, T, row_begin, col_split, col_shift);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// Dense (un-factorised) PreCAR input rows for the dropout path: X[row] = [Xc_s[position] | Xi_s[item row of the CAR row]], rows in the
// CAR row order (BT clicked inputs, then BT*(1+N) candidates).  With a per-element mask the rows are occurrence-specific, so the
// de-duplication / factorisation of the default path (scorer.hip) does not apply.
__global__ __launch_bounds__(256) void k_dense_rows(const float* __restrict__ Xc, int Fc, const float* __restrict__ Xi, int Fi, int BT, int N,
                                                    int pmax, const int* __restrict__ neg_slot, float* __restrict__ X) {
    const int row = blockIdx.x, Fw = Fc + Fi;
    int u, v;
    if (row < BT) { u = row; v = row; }
    else {
        const int i = row - BT, bt = i / (N + 1), c = i % (N + 1);
        u = bt;
        if (c == 0) v = BT + bt;
        else { int s = neg_slot[(size_t)bt * N + (c - 1)]; if (s < 0) s = pmax; v = 2 * BT + s; }
    }
    float* o = X + (size_t)row * Fw;
    for (int k = threadIdx.x; k < Fc / 4; k += 256) reinterpret_cast<float4*>(o)[k] = reinterpret_cast<const float4*>(Xc + (size_t)u * Fc)[k];
    for (int k = threadIdx.x; k < Fi / 4; k += 256) reinterpret_cast<float4*>(o + Fc)[k] = reinterpret_cast<const float4*>(Xi + (size_t)v * Fi)[k];
}
extern "C" int cham_dense_rows(const float* Xc_s, int Fc, const float* Xi_s, int Fi, int BT, int N, int pmax, const int32_t* neg_slot,
                               float* X, void* stream) {
    if (!Xc_s || !Xi_s || !neg_slot || !X || (Fc & 3) || (Fi & 3) || BT <= 0 || N <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_dense_rows, dim3((unsigned)(BT + BT * (N + 1))), dim3(256), 0, (hipStream_t)stream, Xc_s, Fc, Xi_s, Fi, BT, N, pmax,
                       neg_slot, X);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
