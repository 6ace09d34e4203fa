// Device-resident recent-clicks state: the ClickedItemsState of the reference kept in HBM.
//
// Replaces, for the training path, nar_module/nar/clicked_items_state.py:187-250 (update_items_state:
// _update_recently_clicked_items_buffer :206-223, truncate_last_hours_recent_clicks_buffer :225-228,
// _update_recent_pop_items :231-240, _update_recent_pop_norm :242-246, _update_pop_items :248-250) and the hook's
// flattening of the batch into (ids, timestamps), nar_module/nar/nar_model.py:1635-1646.
// Integer work + one float64 division per article: results are BIT-IDENTICAL to the reference class
// (tests/test_state_gpu.py replays tests/golden/state_trace.npz).  The buffer is small (<= a few 10^4 entries), so the
// order-preserving compactions run in ONE workgroup (contiguous segment per thread + LDS scan of the segment counts);
// the histogram / normalisation passes are plain grid kernels with integer atomics (order-independent, deterministic).
#include "common.h"

#define ST_THREADS 1024

// exclusive scan of one int per thread across the block; returns this thread's prefix, *total = block total
__device__ __forceinline__ int block_exclusive_scan(int v, int* sh /*[ST_THREADS/64 + 1]*/, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < ST_THREADS / 64; ++i) { if (i < w) off += sh[i]; tot += sh[i]; }
    *total = tot;
    return off + incl - v;
}

__device__ __forceinline__ int64_t click_ts(const int64_t* __restrict__ event_ts, const int64_t* __restrict__ rowmax, int T, int b, int j) {
    return j < T ? event_ts[(size_t)b * T + j] : rowmax[b];   // the last label re-uses the max timestamp of its session (:1642)
}

// aci [B, T+1] = concat(item_clicked, label_last_item); event_ts [B, T].  One workgroup.
__global__ __launch_bounds__(ST_THREADS) void k_state_buffer_update(
    const int64_t* __restrict__ aci, const int64_t* __restrict__ event_ts, int B, int T, long long hours_ms,
    int64_t* __restrict__ buf_ids, int64_t* __restrict__ buf_ts, int buffer_size,
    int64_t* __restrict__ rowmax /*[B]*/, int64_t* __restrict__ new_ids, int64_t* __restrict__ new_ts /*[buffer_size] each*/,
    int32_t* __restrict__ n_valid) {
    __shared__ int sh[ST_THREADS / 64 + 1];
    __shared__ long long smin[ST_THREADS / 64];
    const int T1 = T + 1, n = B * T1, tid = threadIdx.x;
    for (int b = tid; b < B; b += ST_THREADS) {
        long long m = event_ts[(size_t)b * T];
        for (int t = 1; t < T; ++t) { const long long v = event_ts[(size_t)b * T + t]; m = v > m ? v : m; }
        rowmax[b] = m;
    }
    __syncthreads();
    // ---- batch: min timestamp over the non-padding clicks, count per contiguous segment
    const int seg = (n + ST_THREADS - 1) / ST_THREADS;
    const int i0 = tid * seg, i1 = min(n, i0 + seg);
    long long mn = 0x7FFFFFFFFFFFFFFFLL;
    int cnt = 0;
    for (int i = i0; i < i1; ++i)
        if (aci[i] != 0) { ++cnt; const long long t = click_ts(event_ts, rowmax, T, i / T1, i % T1); mn = t < mn ? t : mn; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(mn, o, 64); mn = t < mn ? t : mn; }
    if ((tid & 63) == 0) smin[tid >> 6] = mn;
    int nb;
    const int pre = block_exclusive_scan(cnt, sh, &nb);     // (barriers inside also publish smin)
    if (nb == 0) return;                                    // no click in the batch: state unchanged (uniform exit)
    for (int i = 0; i < ST_THREADS / 64; ++i) mn = smin[i] < mn ? smin[i] : mn;
    const long long thr = mn - hours_ms;                    // :225-228
    // reversed batch clicks first (:212): the j-th non-padding click (row-major order) lands at nb-1-j
    int j = pre;
    for (int i = i0; i < i1; ++i) {
        const int64_t id = aci[i];
        if (id != 0) {
            const int dst = nb - 1 - j;
            if (dst < buffer_size) { new_ids[dst] = id; new_ts[dst] = click_ts(event_ts, rowmax, T, i / T1, i % T1); }
            ++j;
        }
    }
    // ---- old buffer rows within the time window, order kept, appended behind the batch (:218)
    const int bseg = (buffer_size + ST_THREADS - 1) / ST_THREADS;
    const int k0 = tid * bseg, k1 = min(buffer_size, k0 + bseg);
    int kc = 0;
    for (int k = k0; k < k1; ++k) kc += buf_ts[k] >= thr ? 1 : 0;
    int nk;
    int r = block_exclusive_scan(kc, sh, &nk);
    for (int k = k0; k < k1; ++k)
        if (buf_ts[k] >= thr) {
            const int dst = nb + r;
            if (dst < buffer_size) { new_ids[dst] = buf_ids[k]; new_ts[dst] = buf_ts[k]; }
            ++r;
        }
    const int total = min(nb + nk, buffer_size);
    __syncthreads();
    // ---- publish (zero padding behind the valid prefix, :220-223)
    for (int k = tid; k < buffer_size; k += ST_THREADS) {
        buf_ids[k] = k < total ? new_ids[k] : 0;
        buf_ts[k] = k < total ? new_ts[k] : 0;
    }
    if (tid == 0) { n_valid[0] = total; n_valid[1] = 0; }       // [1]: clicks counted into recent_pop (k_state_hist)
}

__global__ __launch_bounds__(256) void k_state_hist(const int64_t* __restrict__ buf_ids, int32_t* __restrict__ n_valid,
                                                    int32_t* __restrict__ recent_pop, int n_items) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool counted = false;
    if (i < n_valid[0]) {
        const int64_t id = buf_ids[i];
        counted = id > 0 && id < n_items;
        if (counted) atomicAdd(&recent_pop[id], 1);
    }
    // the denominator of pop_norm is sum(recent_pop) (clicked_items_state.py:242-246): the clicks that were COUNTED - a retained
    // buffer row with a zero id (zero-padding that survives the time window when timestamps are tiny) is not one of them
    const unsigned long long m = __ballot(counted);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&n_valid[1], __popcll(m));
}
// pop_norm = max(recent_pop / (sum(recent_pop) + 1), 1 / recent_clicks_for_normalization)  in float64, fed as float32 (:242-246)
__global__ __launch_bounds__(256) void k_state_pop_norm(const int32_t* __restrict__ recent_pop, const int32_t* __restrict__ n_valid,
                                                        int n_items, double min_norm, float* __restrict__ pop_norm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_items) {
        const double v = (double)recent_pop[i] / (double)(n_valid[1] + 1);
        pop_norm[i] = (float)(v > min_norm ? v : min_norm);
    }
}
__global__ __launch_bounds__(256) void k_state_global_pop(const int64_t* __restrict__ aci, int n, unsigned long long* __restrict__ articles_pop,
                                                          int n_items) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int64_t id = aci[i];
        if (id > 0 && id < n_items) atomicAdd(&articles_pop[id], 1ull);     // :248-250
    }
}

extern "C" size_t cham_state_workspace_bytes(int B, int buffer_size) {
    return ((size_t)B + 2 * (size_t)buffer_size) * sizeof(int64_t);
}

// One state update from a batch that is already in HBM.  buf_ids/buf_ts [buffer_size] (newest first, zero padded),
// recent_pop [n_items] int32, pop_norm [n_items] float32, articles_pop [n_items] int64, n_valid: device int32[2] = {rows retained in
// the buffer, clicks counted into recent_pop}.
extern "C" int cham_state_update(const int64_t* aci, const int64_t* event_ts, int B, int T, double buffer_hours,
                                 int64_t* buf_ids, int64_t* buf_ts, int buffer_size, int32_t* recent_pop, float* pop_norm,
                                 int64_t* articles_pop, int n_items, int for_norm, int32_t* n_valid, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!aci || !event_ts || !buf_ids || !buf_ts || !recent_pop || !pop_norm || !articles_pop || !n_valid || !workspace)
        return -CHAM_ERR_ARG;
    if (B <= 0 || T <= 0 || buffer_size <= 0 || n_items <= 0 || for_norm <= 0) return -CHAM_ERR_ARG;
    if (workspace_bytes < cham_state_workspace_bytes(B, buffer_size)) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t* rowmax = (int64_t*)workspace;
    int64_t* new_ids = rowmax + B;
    int64_t* new_ts = new_ids + buffer_size;
    const long long hours_ms = (long long)(buffer_hours * 3600000.0);      // int(hours * MILISECS_BY_HOUR), :226
    hipLaunchKernelGGL(k_state_buffer_update, dim3(1), dim3(ST_THREADS), 0, st, aci, event_ts, B, T, hours_ms, buf_ids, buf_ts,
                       buffer_size, rowmax, new_ids, new_ts, n_valid);
    if (hipMemsetAsync(recent_pop, 0, (size_t)n_items * sizeof(int32_t), st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    hipLaunchKernelGGL(k_state_hist, dim3((buffer_size + 255) / 256), dim3(256), 0, st, buf_ids, n_valid, recent_pop, n_items);
    hipLaunchKernelGGL(k_state_pop_norm, dim3((n_items + 255) / 256), dim3(256), 0, st, recent_pop, n_valid, n_items,
                       1.0 / (double)for_norm, pop_norm);
    const int n = B * (T + 1);
    hipLaunchKernelGGL(k_state_global_pop, dim3((n + 255) / 256), dim3(256), 0, st, aci, n, (unsigned long long*)articles_pop, n_items);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
