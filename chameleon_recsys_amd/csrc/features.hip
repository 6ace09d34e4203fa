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

thread_local float g_cham_ln_elapsed_base = logf(1.3f), g_cham_ln_pop_base = logf(2.0f), g_cham_inv_log2_pop_base = 1.0f;
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
                                                          float* __restrict__ rec_raw, float* __restrict__ nov_raw, float ln_e, float ln_p,
                                                          const ChamStepScalars* __restrict__ sc) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    if (sc) max_ts = sc->max_ts;
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

// (b) the trainable item-embedding table (46 k ... 5 M rows, <= a few 10 k of them touched per step): the item rows are SORTED by
//     (id, row) - a stable least-significant-digit radix sort of the 32-bit ids, 8 bits per pass, ceil(key_bits / 8) passes; integer
//     work that depends on the ids only, so it runs in the forward pass - which makes equal ids contiguous in `perm`, their rows in
//     ascending order; one wave / workgroup per segment then adds its rows in that order and STORES the table row (distinct segments =
//     distinct table rows): deterministic, no float atomics.  (Rounds 1-3 ranked every key against every other key - O(R^2) compares
//     and a 2^20-row limit; the sort is O(R) per pass and has no limit but memory.)
//     One pass = k_rs_hist (per-workgroup digit histogram of its tile of RS_TILE keys) -> k_rs_scan (exclusive scan over [digit][workgroup])
//     -> k_rs_scatter: a workgroup walks its tile in rounds of 256 consecutive keys; a key's position = scanned base of its digit +
//     the digit's count in earlier rounds + in earlier waves of this round + among the lower lanes of its wave (equal-digit lane mask
//     from eight ballots) - ascending input order within a digit, i.e. stable.
#define RS_TILE 2048
__global__ __launch_bounds__(256) void k_rs_hist(const unsigned* __restrict__ keys, const int64_t* __restrict__ ids, int R, int shift, int nb,
                                                 int* __restrict__ hist /*[256][nb]*/) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * RS_TILE;
    for (int q = 0; q < RS_TILE / 256; ++q) {
        const int i = t0 + q * 256 + threadIdx.x;
        if (i < R) atomicAdd(&h[((keys ? keys[i] : (unsigned)ids[i]) >> shift) & 255u], 1);       // integer: order independent
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_rs_scan(int* __restrict__ hist, int n) {       // exclusive scan in place, one workgroup
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? hist[i] : 0;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int off = carry_s, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < w) off += wsum[q]; tot += wsum[q]; }
        if (i < n) hist[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_rs_scatter(const unsigned* __restrict__ keys_in, const int64_t* __restrict__ ids, const int* __restrict__ vals_in,
                                                    int R, int shift, int nb, const int* __restrict__ hist, unsigned* __restrict__ keys_out,
                                                    int* __restrict__ vals_out) {
    __shared__ int base[256], run[256], wc[4][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    base[tid] = hist[(size_t)tid * nb + blockIdx.x];
    run[tid] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) wc[q][tid] = 0;
    __syncthreads();
    const int t0 = blockIdx.x * RS_TILE;
    for (int q = 0; q < RS_TILE / 256; ++q) {
        const int i = t0 + q * 256 + tid;
        const bool ok = i < R;
        const unsigned key = ok ? (keys_in ? keys_in[i] : (unsigned)ids[i]) : 0u;
        const int val = ok ? (vals_in ? vals_in[i] : i) : 0;
        const unsigned d = (key >> shift) & 255u;
        unsigned long long m = __ballot(ok);              // lanes of this wave with the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const int below = __popcll(m & ((1ull << lane) - 1ull));
        if (ok && below == 0) wc[w][d] = __popcll(m);     // the digit's first lane of the wave records the wave's count
        __syncthreads();
        if (ok) {
            int pos = base[d] + run[d] + below;
            for (int ww = 0; ww < w; ++ww) pos += wc[ww][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        run[tid] += wc[0][tid] + wc[1][tid] + wc[2][tid] + wc[3][tid];       // thread = digit
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) wc[ww][tid] = 0;
        __syncthreads();
    }
}
// Segment table of the sorted row list (depends on the ids only: built in the forward pass next to `perm`), int32 words:
//   seg[0] = number of segments, seg[1] = number of LONG segments (> 32 rows), seg[2] = number of work items of the long segments
//   (chunks of EMB_CHUNK rows); seg[4 ..] = first sorted position of every segment (+ R as the end marker); then the indices of the
//   long segments; then the first work item of every long segment (+ end marker); then the partial sums of the work items (as floats).
// One workgroup, block scans, ascending order everywhere.
// (Round 2 launched one workgroup per SORTED POSITION, twice, and let the non-heads exit, and a popular article's rows - Zipf ids: the
// top item holds ~12 % of a micro-batch's 23 k rows at the config-5 size - were summed by ONE workgroup in two stripes of dependent
// index -> row loads: 1.08 ms for 71 MB.  Now: one wave per short segment; long segments cut into chunks of EMB_CHUNK rows, one workgroup
// per chunk, the chunk sums added in chunk order.  profiles/r03_notes.md)
#define EMB_CHUNK 64
__host__ __device__ inline int seg_nl(int R) { return R / 33 + 2; }                       // capacity of the long-segment list
__host__ __device__ inline int seg_nw(int R) { return R / 33 + R / EMB_CHUNK + 4; }       // capacity of the work list
__host__ __device__ inline size_t seg_off_long(int R) { return (size_t)R + 6; }
__host__ __device__ inline size_t seg_off_work(int R) { return seg_off_long(R) + seg_nl(R); }
__host__ __device__ inline size_t seg_off_ticket(int R) { return seg_off_work(R) + seg_nl(R) + 1; }     // one arrival counter per long segment
__host__ __device__ inline size_t seg_off_partial(int R) { return (seg_off_ticket(R) + seg_nl(R) + 3) & ~(size_t)3; }
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
        if (len > 32) { long_list[o] = q; work_first[o] = ow; seg[seg_off_ticket(R) + o] = 0; ++o; ow += (len + EMB_CHUNK - 1) / EMB_CHUNK; }
    }
    if (tid == 0) { seg[1] = n_long; seg[2] = n_work; work_first[n_long] = n_work; }
}
// ONE launch (rounds 2-3: three - short segments, chunk sums, chunk-sum totals).  At a few 10 k rows this is not a bandwidth problem
// but a chain of dependent HBM latencies (the Zipf-hot id of a config-5 micro-batch has ~ 6 000 rows), so every stage keeps 32-64 loads
// per lane in flight and the long segments come FIRST in the grid:
//  - workgroups [0, n_long_wg): work item = EMB_CHUNK (64) consecutive sorted rows of one LONG segment (> 32 rows); wave w adds rows
//    16 w .. 16 w + 15 in order, lanes over the columns (NG groups of 64 columns), all of them in flight; the four wave sums are added in
//    wave order -> the chunk sum.  A one-chunk segment stores its table row directly; otherwise the chunk sum goes to `partial`, the
//    workgroup takes a ticket of its segment, and the one that arrives LAST adds the segment's chunk sums (each wave a contiguous
//    quarter of the chunks in chunk order, then the quarters in wave order - a fixed tree whoever is last: bit-reproducible) and stores
//    the table row.  Release / acquire at agent scope around the ticket (the chunk sums cross XCDs); the last workgroup re-arms it;
//  - the rest: short segments (<= 32 rows: almost all), one WAVE per segment, rows in order, 32 / NG rows x NG column groups in flight.
template <int NG>
__global__ __launch_bounds__(256) void k_emb_grad_all(const float* __restrict__ dxs, int R, int F, int c0, int dim,
                                                      const float* __restrict__ gamma, const int64_t* __restrict__ ids,
                                                      const int* __restrict__ perm, int* seg, float* __restrict__ table_grad, int n_long_wg) {
    __shared__ float comb[4][64 * NG];
    __shared__ int last_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)blockIdx.x >= n_long_wg) {
        const int k = ((int)blockIdx.x - n_long_wg) * 4 + w;
        if (k >= seg[0]) return;
        const int start = seg[4 + k], len = seg[4 + k + 1] - start;
        if (len > 32) return;
        const int my_row = perm[start + min(lane, len - 1)];       // the segment's rows, one per lane
        const int64_t id = ids[__shfl(my_row, 0, 64)];
        constexpr int RF = 32 / NG;
        float a[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) a[j] = 0.f;
        for (int m = 0; m < len; m += RF) {
            float x[RF][NG];
#pragma unroll
            for (int u = 0; u < RF; ++u) {
                const float* src = dxs + (size_t)__shfl(my_row, min(m + u, len - 1), 64) * F + c0;
#pragma unroll
                for (int j = 0; j < NG; ++j) x[u][j] = (m + u < len && lane + 64 * j < dim) ? src[lane + 64 * j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < RF; ++u)
#pragma unroll
                for (int j = 0; j < NG; ++j) a[j] += x[u][j];        // rows in ascending order per column (adding 0.f for the tail is exact)
        }
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if (lane + 64 * j < dim) table_grad[(size_t)id * dim + lane + 64 * j] = a[j] * gamma[c0 + lane + 64 * j];
        return;
    }
    const int wid = blockIdx.x;
    if (wid >= seg[2]) return;
    const int* work_first = seg + seg_off_work(R);
    int lo = 0, hi = seg[1];                       // the long segment q with work_first[q] <= wid < work_first[q + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (work_first[mid] <= wid) lo = mid; else hi = mid; }
    const int k = seg[seg_off_long(R) + lo];
    const int start = seg[4 + k], len = seg[4 + k + 1] - start;
    const int w0 = work_first[lo], w1 = work_first[lo + 1];
    const int r0 = (wid - w0) * EMB_CHUNK, n = min(EMB_CHUNK, len - r0);
    float a[NG];
    {
        const int nw = max(0, min(EMB_CHUNK / 4, n - w * (EMB_CHUNK / 4)));         // this wave's rows of the chunk
        const int my_row = lane < nw ? perm[start + r0 + w * (EMB_CHUNK / 4) + lane] : 0;
        constexpr int RB = 64 / NG;
#pragma unroll
        for (int j = 0; j < NG; ++j) a[j] = 0.f;
        for (int m = 0; m < nw; m += RB) {
            float x[RB][NG];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const float* src = dxs + (size_t)__shfl(my_row, min(m + u, nw - 1), 64) * F + c0;
#pragma unroll
                for (int j = 0; j < NG; ++j) x[u][j] = (m + u < nw && lane + 64 * j < dim) ? src[lane + 64 * j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < RB; ++u)
#pragma unroll
                for (int j = 0; j < NG; ++j) a[j] += x[u][j];
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) comb[w][lane + 64 * j] = a[j];
    }
    __syncthreads();
    const bool single = (w1 - w0 == 1);
    const int64_t id = ids[perm[start]];
    float* partial = reinterpret_cast<float*>(seg + seg_off_partial(R));
    for (int sub = threadIdx.x; sub < dim; sub += 256) {
        const float t = ((comb[0][sub] + comb[1][sub]) + comb[2][sub]) + comb[3][sub];
        if (single) table_grad[(size_t)id * dim + sub] = t * gamma[c0 + sub];
        else partial[(size_t)wid * dim + sub] = t;
    }
    if (single) return;
    __threadfence();                               // release: this workgroup's chunk sum before its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        int* ticket = seg + seg_off_ticket(R) + lo;
        const int t = atomicAdd(ticket, 1);
        last_s = (t == w1 - w0 - 1);
        if (last_s) atomicExch(ticket, 0);
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();                               // acquire: the other workgroups' chunk sums
    {
        const int per = (w1 - w0 + 3) / 4, c_lo = min(w1, w0 + w * per), c_hi = min(w1, c_lo + per);
        constexpr int CB = 32 / NG;
#pragma unroll
        for (int j = 0; j < NG; ++j) a[j] = 0.f;
        for (int c = c_lo; c < c_hi; c += CB) {
            float x[CB][NG];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int j = 0; j < NG; ++j)
                    x[u][j] = (c + u < c_hi && lane + 64 * j < dim) ? __builtin_nontemporal_load(partial + (size_t)(c + u) * dim + lane + 64 * j) : 0.f;
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int j = 0; j < NG; ++j) a[j] += x[u][j];
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) comb[w][lane + 64 * j] = a[j];
    }
    __syncthreads();
    for (int sub = threadIdx.x; sub < dim; sub += 256)
        table_grad[(size_t)id * dim + sub] = (((comb[0][sub] + comb[1][sub]) + comb[2][sub]) + comb[3][sub]) * gamma[c0 + sub];
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
                       pop_norm, scratch, scratch + n_last, g_cham_ln_elapsed_base, g_cham_ln_pop_base, (const ChamStepScalars*)nullptr);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch, (const float*)nullptr, n_last, stats, 3);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch + n_last, (const float*)nullptr, n_last, stats + 4, 3);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// same statistics straight from the device-resident recent-clicks buffer (csrc/state.hip): population = the first
// min(n_prefix, #valid) entries (valid entries are a zero-padded prefix; nar_model.py:1041-1044).  scratch: 3 * n_prefix floats.
static int norm_stats_from_buffer_impl(const int64_t* buffer_ids, int n_prefix, int64_t max_ts, const int64_t* created,
                                       const float* pop_norm, float* scratch, float* stats /*[3][8]*/, const void* scalars, void* stream) {
    if (!buffer_ids || n_prefix <= 0 || !created || !pop_norm || !scratch || !stats) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* w = scratch + 2 * (size_t)n_prefix;
    hipLaunchKernelGGL(k_nonzero_weights, dim3((n_prefix + 255) / 256), dim3(256), 0, st, buffer_ids, n_prefix, w);
    hipLaunchKernelGGL(k_last_dynamic_raw, dim3((n_prefix + 255) / 256), dim3(256), 0, st, buffer_ids, n_prefix, max_ts, created,
                       pop_norm, scratch, scratch + n_prefix, g_cham_ln_elapsed_base, g_cham_ln_pop_base, reinterpret_cast<const ChamStepScalars*>(scalars));
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch, (const float*)w, n_prefix, stats, 3);
    hipLaunchKernelGGL(k_norm_stats, dim3(1), dim3(1024), 0, st, scratch + n_prefix, (const float*)w, n_prefix, stats + 4, 3);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_norm_stats_from_buffer(const int64_t* buffer_ids, int n_prefix, int64_t max_ts, const int64_t* created,
                                           const float* pop_norm, float* scratch, float* stats /*[3][8]*/, void* stream) {
    return norm_stats_from_buffer_impl(buffer_ids, n_prefix, max_ts, created, pop_norm, scratch, stats, nullptr, stream);
}
// max_ts from the ChamStepScalars record (device; common.h)
extern "C" int cham_norm_stats_from_buffer_dev(const int64_t* buffer_ids, int n_prefix, const void* scalars, const int64_t* created,
                                               const float* pop_norm, float* scratch, float* stats /*[3][8]*/, void* stream) {
    if (!scalars) return -CHAM_ERR_ARG;
    return norm_stats_from_buffer_impl(buffer_ids, n_prefix, 0, created, pop_norm, scratch, stats, scalars, stream);
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

// The same column sums with COALESCED reads (round 6).  k_scale_bwd_cols gives every column a workgroup whose lanes walk the ROWS: each
// load touches 4 bytes of a different 128-byte line - 32 x read amplification (164 MB of HBM / 1.1 GB of L2 traffic for the 17 MB item
// matrix) and 0.06-0.13 ms per call on the serial tail of the step, behind the W2 weight gradient, where nothing else runs.  Here a
// workgroup owns 64 COLUMNS x a chunk of rows: wave w adds rows r0 + w, r0 + w + 4, ... (lanes = 64 consecutive columns: 256-byte row
// segments, eight loads in flight), writes its partial sums; a second launch adds the partials of a column in ascending (chunk, wave)
// order: deterministic (another order than k_scale_bwd_cols': the sums differ in the last bits).  workspace: SBW_CHUNKS x 4 x 2 x F floats.
#define SBW_CHUNKS 64
__global__ __launch_bounds__(256) void k_scale_bwd_part(const float* __restrict__ dxs, const float* __restrict__ xraw, int R, int F, int rows_per_chunk,
                                                        float* __restrict__ part) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float sg = 0.f, sb = 0.f;
    if (c < F) {
        for (int r = r0 + w; r < r1; r += 16) {
            float g[4], x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = min(r + 4 * u, r1 - 1);
                g[u] = dxs[(size_t)rr * F + c]; x[u] = xraw[(size_t)rr * F + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 4 * u < r1) { sg += g[u] * x[u]; sb += g[u]; }
        }
        float* o = part + ((size_t)(chunk * 4 + w) * 2) * F;
        o[c] = sg; o[F + c] = sb;
    }
}
__global__ __launch_bounds__(256) void k_scale_bwd_fin(const float* __restrict__ part, int F, int nparts, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    float sg = 0.f, sb = 0.f;
    for (int q = 0; q < nparts; q += 8) {
        float g[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* o = part + ((size_t)min(q + u, nparts - 1) * 2) * F;
            g[u] = o[c]; b[u] = o[F + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q + u < nparts) { sg += g[u]; sb += b[u]; }
    }
    dgamma[c] = sg; dbeta[c] = sb;
}

extern "C" int cham_feature_bwd(const float* dxs, const float* xraw, int R, int F, float* dgamma, float* dbeta, void* stream) {
    if (!dxs || !xraw || !dgamma || !dbeta || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_scale_bwd_cols, dim3(F), dim3(256), 0, (hipStream_t)stream, dxs, xraw, R, F, dgamma, dbeta);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" size_t cham_feature_bwd_workspace_bytes(int F) { return (size_t)SBW_CHUNKS * 4 * 2 * (size_t)F * sizeof(float); }
// cham_feature_bwd through the coalesced two-launch form above (workspace >= cham_feature_bwd_workspace_bytes(F); private to the call's stream)
extern "C" int cham_feature_bwd_ws(const float* dxs, const float* xraw, int R, int F, float* dgamma, float* dbeta, float* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!dxs || !xraw || !dgamma || !dbeta || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    if (!workspace || workspace_bytes < cham_feature_bwd_workspace_bytes(F)) return -CHAM_ERR_ARG;
    int rows_per_chunk = (R + SBW_CHUNKS - 1) / SBW_CHUNKS;
    if (rows_per_chunk < 32) rows_per_chunk = 32;
    const int nch = (R + rows_per_chunk - 1) / rows_per_chunk;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scale_bwd_part, dim3((F + 63) / 64, nch), dim3(256), 0, st, dxs, xraw, R, F, rows_per_chunk, workspace);
    hipLaunchKernelGGL(k_scale_bwd_fin, dim3((F + 255) / 256), dim3(256), 0, st, workspace, F, nch * 4, dgamma, dbeta);
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

// workspace of cham_group_rows: two key arrays + one value array of R words and the [256][workgroups] digit histogram
static inline size_t rs_words(int R) { return ((size_t)R + 63) & ~(size_t)63; }
extern "C" size_t cham_group_rows_workspace_bytes(int R) {
    if (R <= 0) return 0;
    const size_t nb = ((size_t)R + RS_TILE - 1) / RS_TILE;
    return (3 * rs_words(R) + 256 * nb) * sizeof(int);
}
extern "C" size_t cham_group_rows_segments_len(int R) {      // int32 words of `seg` (the tail holds the chunk sums of the long segments, <= 512 columns)
    return R > 0 ? seg_off_partial(R) + (size_t)seg_nw(R) * 512 : 0;
}
// perm = the rows 0 .. R-1 sorted by (ids[row], row) and the segment table of equal ids (k_seg_table).  key_bits: the ids are < 2^key_bits
// (0 or > 32: 32) - ceil(key_bits / 8) radix passes.  ids must be non-negative and < 2^32.
extern "C" int cham_group_rows(const int64_t* ids, int R, int key_bits, int32_t* perm, int32_t* seg, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (!ids || !perm || !seg || !workspace || R <= 0 || workspace_bytes < cham_group_rows_workspace_bytes(R) || ((uintptr_t)workspace & 15))
        return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (key_bits <= 0 || key_bits > 32) key_bits = 32;
    const int passes = (key_bits + 7) / 8;
    const int nb = (R + RS_TILE - 1) / RS_TILE;
    unsigned* kbuf[2] = {reinterpret_cast<unsigned*>(workspace), reinterpret_cast<unsigned*>(workspace) + rs_words(R)};
    int* vws = reinterpret_cast<int*>(workspace) + 2 * rs_words(R);
    int* hist = reinterpret_cast<int*>(workspace) + 3 * rs_words(R);
    for (int p = 0; p < passes; ++p) {
        // values ping-pong between the workspace and `perm` so that the LAST pass writes `perm`
        int* vout = ((passes - 1 - p) & 1) ? vws : perm;
        const int* vin = p == 0 ? nullptr : (((passes - p) & 1) ? vws : perm);
        const unsigned* kin = p == 0 ? nullptr : kbuf[(p - 1) & 1];
        hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(256), 0, st, kin, ids, R, 8 * p, nb, hist);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, hist, 256 * nb);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(256), 0, st, kin, ids, vin, R, 8 * p, nb, hist, kbuf[p & 1], vout);
    }
    hipLaunchKernelGGL(k_seg_table, dim3(1), dim3(1024), 0, st, ids, perm, R, seg);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_emb_grad_grouped(const float* dxs, int R, int F, int c0, int dim, const float* gamma, const int64_t* ids,
                                     const int32_t* perm, const int32_t* seg, float* table_grad, void* stream) {
    if (!dxs || !gamma || !ids || !perm || !seg || !table_grad || R <= 0 || F <= 0 || c0 < 0 || dim <= 0 || dim > 512 || c0 + dim > F)
        return -CHAM_ERR_ARG;
    const dim3 grid(seg_nw(R) + (R + 3) / 4);
    if (dim <= 256)
        hipLaunchKernelGGL(k_emb_grad_all<4>, grid, dim3(256), 0, (hipStream_t)stream, dxs, R, F, c0, dim, gamma, ids, perm,
                           const_cast<int32_t*>(seg), table_grad, seg_nw(R));
    else
        hipLaunchKernelGGL(k_emb_grad_all<8>, grid, dim3(256), 0, (hipStream_t)stream, dxs, R, F, c0, dim, gamma, ids, perm,
                           const_cast<int32_t*>(seg), table_grad, seg_nw(R));
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

// ---- head of a step: the integer row sets in ONE launch (round 5; eight copy / fill launches of NARModuleModel.forward before):
//   ids_all = [clicked ids (BT) | positive ids (BT) | candidate pool (pmax) | pad item 0], ref_ts = [click time stamps (BT) | max_ts ...],
//   seq_len (B, int32) and mask (BT, uint8) into the plan's buffers.  nar_model.py:217-248 (inputs / mask block), :343, 356 (reference time
//   stamps of positive / negative rows = the batch's max time stamp).
__global__ __launch_bounds__(256) void k_step_ints(const int64_t* __restrict__ ic, const int64_t* __restrict__ ln, const int64_t* __restrict__ pool,
                                                   const int64_t* __restrict__ ets, int64_t max_ts, int BT, int pmax,
                                                   const int32_t* __restrict__ seq_len_in, int B, const unsigned char* __restrict__ mask_in,
                                                   int64_t* __restrict__ ids_all, int64_t* __restrict__ ref_ts, int32_t* __restrict__ seq_len,
                                                   unsigned char* __restrict__ mask, const ChamStepScalars* __restrict__ sc) {
    const int i = blockIdx.x * 256 + threadIdx.x, RV = 2 * BT + pmax + 1;
    if (sc) max_ts = sc->max_ts;
    if (i < RV) {
        int64_t id;
        if (i < BT) id = ic[i];
        else if (i < 2 * BT) id = ln[i - BT];
        else if (i < 2 * BT + pmax) id = pool[i - 2 * BT];
        else id = 0;
        ids_all[i] = id;
        ref_ts[i] = i < BT ? ets[i] : max_ts;
    }
    if (i < B) seq_len[i] = seq_len_in[i];
    if (i < BT) mask[i] = mask_in[i];
}

static int step_ints_impl(const int64_t* ic_rows, const int64_t* ln_rows, const int64_t* pool, const int64_t* ets_rows, int64_t max_ts, int BT,
                          int pmax, const int32_t* seq_len_in, int B, const uint8_t* mask_in, int64_t* ids_all, int64_t* ref_ts,
                          int32_t* seq_len, uint8_t* mask, const void* scalars, void* stream) {
    if (!ic_rows || !ln_rows || !pool || !ets_rows || !seq_len_in || !mask_in || !ids_all || !ref_ts || !seq_len || !mask || BT < 0 || pmax < 0 || B < 0)
        return -CHAM_ERR_ARG;
    const int n = 2 * BT + pmax + 1 > B ? 2 * BT + pmax + 1 : B;
    hipLaunchKernelGGL(k_step_ints, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ic_rows, ln_rows, pool, ets_rows, max_ts, BT, pmax,
                       seq_len_in, B, mask_in, ids_all, ref_ts, seq_len, mask, reinterpret_cast<const ChamStepScalars*>(scalars));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_step_ints(const int64_t* ic_rows, const int64_t* ln_rows, const int64_t* pool, const int64_t* ets_rows, int64_t max_ts, int BT,
                              int pmax, const int32_t* seq_len_in, int B, const uint8_t* mask_in, int64_t* ids_all, int64_t* ref_ts,
                              int32_t* seq_len, uint8_t* mask, void* stream) {
    return step_ints_impl(ic_rows, ln_rows, pool, ets_rows, max_ts, BT, pmax, seq_len_in, B, mask_in, ids_all, ref_ts, seq_len, mask, nullptr, stream);
}
// max_ts from the ChamStepScalars record (device; common.h)
extern "C" int cham_step_ints_dev(const int64_t* ic_rows, const int64_t* ln_rows, const int64_t* pool, const int64_t* ets_rows, const void* scalars, int BT,
                                  int pmax, const int32_t* seq_len_in, int B, const uint8_t* mask_in, int64_t* ids_all, int64_t* ref_ts,
                                  int32_t* seq_len, uint8_t* mask, void* stream) {
    if (!scalars) return -CHAM_ERR_ARG;
    return step_ints_impl(ic_rows, ln_rows, pool, ets_rows, 0, BT, pmax, seq_len_in, B, mask_in, ids_all, ref_ts, seq_len, mask, scalars, stream);
}
