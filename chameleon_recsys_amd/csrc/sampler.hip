// Negative sampler on device (integer; bit-exact contract with oracle/sampler.py).
//
// Replaces nar_module/nar/nar_model.py:1220-1304 (get_sample_from_recently_clicked_items_buffer,
// get_batch_negative_samples -> get_negative_samples -> get_neg_items_session -> get_neg_items_click),
// i.e. two nested tf.map_fn = B*(T+1) sequential TF while-loop iterations, with 6 small kernels.
// Every tf.random_shuffle is the keyed sort of Philox4x32-10 keys ((rand32 << 32) | q), see common.h.
//
//   1. keys over the recent-clicks buffer      -> rank-select the first n_from_buffer  (buffer sample)
//   2. keys over [batch ids ; buffer sample]   -> rank-select the first 20*N           (candidate pool)
//   3. canon[q] = first pool position holding the same id (de-duplication handle: "slot")
//   4. one workgroup per click (b,j): ordered setdiff against the session's ids, per-value MIN key via
//      LDS ds_min_u64, bitonic sort in LDS, first N distinct ids (+ their slots), zero padding.
//
// Rank-select = "out[r] = value whose key is the r-th smallest": O(n^2) key comparisons out of LDS,
// n <= buffer size (20 000): ~40 us, deterministic, no global sort needed.
#include "common.h"

// (step_dev != NULL: the key of the step is read from device memory - a ChamStepScalars field, common.h - instead of the argument)
__global__ __launch_bounds__(256) void k_keys_buffer(const int64_t* __restrict__ buf, int n, uint64_t* __restrict__ keys,
                                                     uint32_t seed, uint32_t step, const uint32_t* __restrict__ step_dev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (step_dev) step = *step_dev;
    keys[i] = buf[i] != 0 ? philox_sort_key((uint32_t)i, 0u, 0u, 0u, seed, step) : CHAM_INF_KEY;
}

// concatenation [all_clicked_items.ravel() ; buffer-sample slots]
__global__ __launch_bounds__(256) void k_keys_pool(const int64_t* __restrict__ aci, int n_aci,
                                                   const int64_t* __restrict__ buf_sample, int n_slots,
                                                   int64_t* __restrict__ cat_vals, uint64_t* __restrict__ keys,
                                                   uint32_t seed, uint32_t step, const uint32_t* __restrict__ step_dev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_aci + n_slots) return;
    if (step_dev) step = *step_dev;
    const int64_t v = i < n_aci ? aci[i] : buf_sample[i - n_aci];
    cat_vals[i] = v;
    keys[i] = v != 0 ? philox_sort_key((uint32_t)i, 0u, 0u, 1u, seed, step) : CHAM_INF_KEY;
}

// ---- rank-select with a key-threshold prefilter -----------------------------------------------------------------
// Only the `limit` smallest keys are ever used.  The upper 32 key bits are uniform (Philox), so all of them lie below
// thr = ((1.5 * limit + 64) / n_valid) * 2^32 except with probability < 1e-40; ranks are then computed for those ~1.5 * limit
// candidates only (n_cand x n comparisons instead of n x n: the pool of an 8-GPU global batch has n = 44 k keys).
// sel[0] = #valid keys, sel[1] = #keys below thr, sel[2..3] = thr (uint64).  If fewer than min(limit, #valid) keys fall
// below thr every kernel falls back to the full computation, so the result is ALWAYS the exact rank-select.
__global__ __launch_bounds__(256) void k_sel_count_valid(const uint64_t* __restrict__ keys, int n, unsigned* __restrict__ sel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool v = i < n && keys[i] != CHAM_INF_KEY;
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(sel, (unsigned)__popcll(b));
}
__global__ void k_sel_threshold(unsigned* __restrict__ sel, int limit) {
    if (threadIdx.x != 0) return;
    const double want = 1.5 * (double)limit + 64.0, nv = (double)sel[0];
    uint64_t thr = CHAM_INF_KEY;
    if (nv > want) thr = (uint64_t)((want / nv) * 4294967296.0) << 32;
    *reinterpret_cast<uint64_t*>(sel + 2) = thr;
}
__global__ __launch_bounds__(256) void k_sel_count_below(const uint64_t* __restrict__ keys, int n, unsigned* __restrict__ sel) {
    const uint64_t thr = *reinterpret_cast<const uint64_t*>(sel + 2);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool v = i < n && keys[i] < thr;
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(sel + 1, (unsigned)__popcll(b));
}
__device__ __forceinline__ uint64_t sel_threshold(const unsigned* __restrict__ sel, int limit) {
    const unsigned need = min((unsigned)limit, sel[0]);
    return sel[1] >= need ? *reinterpret_cast<const uint64_t*>(sel + 2) : CHAM_INF_KEY;     // not enough candidates: everything
}

// rank[i] = #{j : key[j] < key[i]} for the candidates: 2-D grid (256 elements) x (slice of 2048 keys staged in LDS)
__global__ __launch_bounds__(256) void k_rank_count(const uint64_t* __restrict__ keys, int n, int* __restrict__ rank,
                                                    const unsigned* __restrict__ sel, int limit) {
    __shared__ uint64_t tile[2048];
    const uint64_t thr = sel_threshold(sel, limit);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t mine = i < n ? keys[i] : CHAM_INF_KEY;
    const bool cand = mine != CHAM_INF_KEY && mine < thr;
    if (!__syncthreads_or(cand)) return;                 // no candidate in this block of 256 elements
    const int t0 = blockIdx.y * 2048;
    for (int t = threadIdx.x; t < 2048; t += 256) tile[t] = (t0 + t) < n ? keys[t0 + t] : CHAM_INF_KEY;
    __syncthreads();
    if (!cand) return;
    const int m = min(2048, n - t0);
    int r = 0;
#pragma unroll 8
    for (int t = 0; t < m; ++t) r += (tile[t] < mine) ? 1 : 0;
    if (r) atomicAdd(rank + i, r);
}
// out[rank(i)] = vals[i] for rank < limit; *count = min(limit, #valid)   (out / count / rank pre-zeroed)
__global__ __launch_bounds__(256) void k_rank_place(const uint64_t* __restrict__ keys, const int64_t* __restrict__ vals,
                                                    const int* __restrict__ rank, int n, int limit, int64_t* __restrict__ out,
                                                    int* __restrict__ count, const unsigned* __restrict__ sel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || keys[i] == CHAM_INF_KEY || keys[i] >= sel_threshold(sel, limit)) return;
    const int r = rank[i];
    if (r < limit) { out[r] = vals[i]; atomicAdd(count, 1); }
}

__global__ __launch_bounds__(256) void k_canon(const int64_t* __restrict__ pool, const int* __restrict__ pcount, int pmax,
                                               int* __restrict__ canon) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int64_t* sp = reinterpret_cast<int64_t*>(smem_raw);             // the pool staged in LDS: every thread scans a prefix of it
    const int P = *pcount;
    for (int t = threadIdx.x; t < P; t += 256) sp[t] = pool[t];
    __syncthreads();
    for (int q = threadIdx.x + blockIdx.x * 256; q < pmax; q += gridDim.x * 256) {
        int c = q;
        if (q < P) {
            const int64_t v = sp[q];
            for (int t = 0; t < q; ++t)
                if (sp[t] == v) { c = t; break; }
        }
        canon[q] = c;
    }
}
__global__ __launch_bounds__(256) void k_click_select(const int64_t* __restrict__ aci, int T1, int row_begin,
                                                      const int64_t* __restrict__ pool, const int* __restrict__ canon,
                                                      const int* __restrict__ pcount, int pmax, int pp /*pow2 >= pmax*/,
                                                      int N, uint32_t seed, uint32_t step,
                                                      int64_t* __restrict__ neg_ids, int* __restrict__ neg_slot, const uint32_t* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (step_dev) step = *step_dev;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);            // [pp]
    int64_t* sess = reinterpret_cast<int64_t*>(keys + pp);             // [T1]
    const int j = blockIdx.x, bl = blockIdx.y, b = row_begin + bl;
    const int T = T1 - 1;
    int64_t* oid = neg_ids + ((size_t)bl * T + j) * N;
    int* osl = neg_slot + ((size_t)bl * T + j) * N;
    const int64_t click = aci[(size_t)b * T1 + j];
    if (click == 0) {                                  // nar_model.py:1262-1263: padded click -> zeros
        for (int n = threadIdx.x; n < N; n += 256) { oid[n] = 0; osl[n] = -1; }
        return;
    }
    const int P = *pcount;
    for (int t = threadIdx.x; t < T1; t += 256) sess[t] = aci[(size_t)b * T1 + t];
    for (int q = threadIdx.x; q < pp; q += 256) keys[q] = CHAM_INF_KEY;
    __syncthreads();
    for (int q = threadIdx.x; q < P; q += 256) {
        const int64_t v = pool[q];
        bool valid = true;                              // ordered setdiff vs the session's ids (:1259)
        for (int t = 0; t < T1; ++t) valid &= (sess[t] != v);
        if (valid) {
            const uint64_t k = philox_sort_key((uint32_t)q, (uint32_t)j, (uint32_t)b, 2u, seed, step);
            atomicMin(reinterpret_cast<unsigned long long*>(&keys[canon[q]]), (unsigned long long)k);
        }
    }
    __syncthreads();
    // bitonic sort ascending
    for (int k = 2; k <= pp; k <<= 1) {
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int i = threadIdx.x; i < pp; i += 256) {
                const int x = i ^ s;
                if (x > i) {
                    const uint64_t a = keys[i], c = keys[x];
                    const bool asc = (i & k) == 0;
                    if ((a > c) == asc) { keys[i] = c; keys[x] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int n = threadIdx.x; n < N; n += 256) {
        const uint64_t k = n < pp ? keys[n] : CHAM_INF_KEY;
        if (k == CHAM_INF_KEY) { oid[n] = 0; osl[n] = pmax; }           // zero padding (:1252) -> pad slot
        else {
            const int q = (int)(k & 0xFFFFFFFFull);
            oid[n] = pool[q]; osl[n] = canon[q];
        }
    }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t cham_neg_sample_workspace_bytes(int n_aci, int buf_size, int n_from_buffer) {
    const size_t ncat = (size_t)n_aci + n_from_buffer;
    const size_t nmax = ncat > (size_t)buf_size ? ncat : (size_t)buf_size;
    return align256((size_t)buf_size * 8) + align256((size_t)n_from_buffer * 8) + align256(ncat * 8) * 2 + align256(nmax * 4) + 256;
}

static int neg_sample_impl(const int64_t* aci, int Bg, int T1, const int64_t* buffer, int buf_size,
                           uint32_t seed, uint32_t step, const uint32_t* step_dev, int row_begin, int row_count, int N, int n_from_buffer,
                           int64_t* neg_ids, int32_t* neg_slot, int64_t* pool, int32_t* canon, int32_t* meta,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!aci || !buffer || !neg_ids || !neg_slot || !pool || !canon || !meta || !workspace) return -CHAM_ERR_ARG;
    if (Bg <= 0 || T1 < 2 || N <= 0 || n_from_buffer < 0 || row_begin < 0 || row_begin + row_count > Bg) return -CHAM_ERR_ARG;
    const int n_aci = Bg * T1;
    if (workspace_bytes < cham_neg_sample_workspace_bytes(n_aci, buf_size, n_from_buffer)) return -CHAM_ERR_ARG;
    const int pmax = 20 * N;
    int pp = 1; while (pp < pmax) pp <<= 1;
    if ((size_t)pp * 8 + (size_t)T1 * 8 > 150 * 1024) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* w = static_cast<unsigned char*>(workspace);
    uint64_t* keys0 = reinterpret_cast<uint64_t*>(w); w += align256((size_t)buf_size * 8);
    int64_t* buf_sample = reinterpret_cast<int64_t*>(w); w += align256((size_t)n_from_buffer * 8);
    const int ncat = n_aci + n_from_buffer;
    int64_t* cat_vals = reinterpret_cast<int64_t*>(w); w += align256((size_t)ncat * 8);
    uint64_t* keys1 = reinterpret_cast<uint64_t*>(w); w += align256((size_t)ncat * 8);
    int* rank = reinterpret_cast<int*>(w); w += align256((size_t)(ncat > buf_size ? ncat : buf_size) * 4);
    unsigned* sel = reinterpret_cast<unsigned*>(w);          // two selections x 4 words

    if (hipMemsetAsync(meta, 0, 4 * sizeof(int), st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    if (hipMemsetAsync(sel, 0, 8 * sizeof(unsigned), st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    if (n_from_buffer > 0 && hipMemsetAsync(buf_sample, 0, (size_t)n_from_buffer * 8, st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    if (hipMemsetAsync(pool, 0, (size_t)pmax * 8, st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    if (buf_size > 0 && n_from_buffer > 0) {
        hipLaunchKernelGGL(k_keys_buffer, dim3((buf_size + 255) / 256), dim3(256), 0, st, buffer, buf_size, keys0, seed, step, step_dev);
        if (hipMemsetAsync(rank, 0, (size_t)buf_size * 4, st) != hipSuccess) return -CHAM_ERR_LAUNCH;
        hipLaunchKernelGGL(k_sel_count_valid, dim3((buf_size + 255) / 256), dim3(256), 0, st, keys0, buf_size, sel);
        hipLaunchKernelGGL(k_sel_threshold, dim3(1), dim3(64), 0, st, sel, n_from_buffer);
        hipLaunchKernelGGL(k_sel_count_below, dim3((buf_size + 255) / 256), dim3(256), 0, st, keys0, buf_size, sel);
        hipLaunchKernelGGL(k_rank_count, dim3((buf_size + 255) / 256, (buf_size + 2047) / 2048), dim3(256), 0, st, keys0, buf_size, rank,
                           sel, n_from_buffer);
        hipLaunchKernelGGL(k_rank_place, dim3((buf_size + 255) / 256), dim3(256), 0, st, keys0, buffer, rank, buf_size,
                           n_from_buffer, buf_sample, meta + 1, sel);
    }
    hipLaunchKernelGGL(k_keys_pool, dim3((ncat + 255) / 256), dim3(256), 0, st, aci, n_aci, buf_sample, n_from_buffer,
                       cat_vals, keys1, seed, step, step_dev);
    if (hipMemsetAsync(rank, 0, (size_t)ncat * 4, st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_sel_count_valid, dim3((ncat + 255) / 256), dim3(256), 0, st, keys1, ncat, sel + 4);
    hipLaunchKernelGGL(k_sel_threshold, dim3(1), dim3(64), 0, st, sel + 4, pmax);
    hipLaunchKernelGGL(k_sel_count_below, dim3((ncat + 255) / 256), dim3(256), 0, st, keys1, ncat, sel + 4);
    hipLaunchKernelGGL(k_rank_count, dim3((ncat + 255) / 256, (ncat + 2047) / 2048), dim3(256), 0, st, keys1, ncat, rank, sel + 4, pmax);
    hipLaunchKernelGGL(k_rank_place, dim3((ncat + 255) / 256), dim3(256), 0, st, keys1, cat_vals, rank, ncat, pmax, pool, meta + 3,
                       sel + 4);
    {
        CHAM_SET_DYNAMIC_LDS(k_canon, 150 * 1024);          // per DEVICE, not per process
        hipLaunchKernelGGL(k_canon, dim3((pmax + 255) / 256), dim3(256), (size_t)pmax * 8, st, pool, meta + 3, pmax, canon);
    }
    if (row_count > 0) {
        const size_t smem = (size_t)pp * 8 + (size_t)T1 * 8;
        CHAM_SET_DYNAMIC_LDS(k_click_select, 150 * 1024);
        hipLaunchKernelGGL(k_click_select, dim3(T1 - 1, row_count), dim3(256), smem, st, aci, T1, row_begin, pool, canon,
                           meta + 3, pmax, pp, N, seed, step, neg_ids, neg_slot, step_dev);
    }
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_neg_sample(const int64_t* aci, int Bg, int T1, const int64_t* buffer, int buf_size,
                               uint32_t seed, uint32_t step, int row_begin, int row_count, int N, int n_from_buffer,
                               int64_t* neg_ids, int32_t* neg_slot, int64_t* pool, int32_t* canon, int32_t* meta,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return neg_sample_impl(aci, Bg, T1, buffer, buf_size, seed, step, nullptr, row_begin, row_count, N, n_from_buffer, neg_ids, neg_slot, pool, canon, meta,
                           workspace, workspace_bytes, stream);
}
// the step's key from the ChamStepScalars record `scalars` (device; common.h): which = 0 -> .step, 1 -> .step_next (the NEXT step's negatives,
// drawn behind the current step: NARModuleModel.presample)
extern "C" int cham_neg_sample_dev(const int64_t* aci, int Bg, int T1, const int64_t* buffer, int buf_size,
                                   uint32_t seed, const void* scalars, int which, int row_begin, int row_count, int N, int n_from_buffer,
                                   int64_t* neg_ids, int32_t* neg_slot, int64_t* pool, int32_t* canon, int32_t* meta,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!scalars || (which != 0 && which != 1)) return -CHAM_ERR_ARG;
    const ChamStepScalars* sc = reinterpret_cast<const ChamStepScalars*>(scalars);
    return neg_sample_impl(aci, Bg, T1, buffer, buf_size, seed, 0u, which ? &sc->step_next : &sc->step, row_begin, row_count, N, n_from_buffer, neg_ids,
                           neg_slot, pool, canon, meta, workspace, workspace_bytes, stream);
}
