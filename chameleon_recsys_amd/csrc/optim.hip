// L2 regularisation, TF-flavoured Adam, bias-gradient column sums, loss finalisation (HBM-bound).
//
// Replaces nar_module/nar/nar_model.py:655 (tf.losses.get_regularization_loss: sum of scale * l2_loss over
// Dense kernels, embeddings, gamma/beta - NOT biases, NOT the RNN cell), :660-667 (loss) and :708-722
// (tf.train.AdamOptimizer(lr, 0.9, 0.999, 1e-8).apply_gradients).  TF Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
// p -= lr_t * m / (sqrt(v) + eps); embedding tables are updated DENSELY (IndexedSlices + dense L2 term are
// densified by TF; m, v of every row decay every step).  All parameters live in ONE flat fp32 buffer with the
// regularised tensors first, so the whole optimizer is a single streaming pass: 28 B/param/step.
#include "common.h"

__global__ __launch_bounds__(256) void k_adam_tf(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, size_t n4, size_t n_reg4, float lambda, float lr_t,
                                                 float b1, float b2, float eps, const ChamStepScalars* __restrict__ sc) {
    if (sc) lr_t = sc->lr_t;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        const float l = i < n_reg4 ? lambda : 0.f;           // d/dw (lambda * |w|^2 / 2) = lambda * w
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = G[k] + l * P[k];
            M[k] = b1 * M[k] + (1.f - b1) * gr;
            V[k] = b2 * V[k] + (1.f - b2) * gr * gr;
            P[k] = P[k] - lr_t * M[k] / (sqrtf(V[k]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ p, size_t n4, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = reinterpret_cast<const float4*>(p)[i];
        s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// loss[0] = xe + reg, loss[1] = xe = sum(nll)/sum_mask, loss[2] = reg = lambda/2 * sum(w^2)
__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ nll, int BT, float inv_sum_mask,
                                                       const float* __restrict__ sumsq_partial, int n_partial, float lambda,
                                                       float* __restrict__ loss, const ChamStepScalars* __restrict__ sc) {
    __shared__ float red[4];
    if (sc) inv_sum_mask = 1.0f / sc->sum_mask;       // (the by-value entry point computes the same fp32 quotient on the host)
    float s = 0.f;
    for (int i = threadIdx.x; i < BT; i += 256) s += nll[i];
    s = block_sum(s, red);
    float q = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) q += sumsq_partial[i];
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        const float xe = s * inv_sum_mask, reg = 0.5f * lambda * q;
        loss[0] = xe + reg; loss[1] = xe; loss[2] = reg;
    }
}

// out[c] = sum_r w[r] * X[r, c]   (w optional).  Stage 1: fixed row chunks -> partial[chunk][F]; stage 2 sums
// the chunks in order.  Deterministic; X is streamed once with float4 loads when F/4 divides 256.
template <typename T>
__global__ __launch_bounds__(256) void k_colsum_vec(const T* __restrict__ X, int ld, int R, int F, const float* __restrict__ w,
                                                    int rows_per_chunk, float* __restrict__ partial) {
    __shared__ float4 sm[256];
    const int nf4 = F / 4, rpi = 256 / nf4;
    const int c4 = threadIdx.x % nf4, rl = threadIdx.x / nf4;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = r0 + rl;
    for (; r + 3 * rpi < r1; r += 4 * rpi) {            // 4 independent row loads in flight (a lane's loads are otherwise a latency chain)
        float4 x[4]; float ww[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = ld4(X + (size_t)(r + u * rpi) * ld + c4 * 4); ww[u] = w ? w[r + u * rpi] : 1.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += ww[u] * x[u].x; acc.y += ww[u] * x[u].y; acc.z += ww[u] * x[u].z; acc.w += ww[u] * x[u].w; }
    }
    for (; r < r1; r += rpi) {
        const float4 x = ld4(X + (size_t)r * ld + c4 * 4);
        const float ww = w ? w[r] : 1.f;
        acc.x += ww * x.x; acc.y += ww * x.y; acc.z += ww * x.z; acc.w += ww * x.w;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rpi; ++k) {
            const float4 y = sm[k * nf4 + c4];
            acc.x += y.x; acc.y += y.y; acc.z += y.z; acc.w += y.w;
        }
        *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * F + c4 * 4) = acc;
    }
}
__global__ __launch_bounds__(256) void k_colsum_gen(const float* __restrict__ X, int ld, int R, int F, const float* __restrict__ w,
                                                    int rows_per_chunk, float* __restrict__ partial) {
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    for (int c = threadIdx.x; c < F; c += 256) {
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) acc += (w ? w[r] : 1.f) * X[(size_t)r * ld + c];
        partial[(size_t)blockIdx.x * F + c] = acc;
    }
}
// 16 columns x 16 chunk-lanes per workgroup; lane l sums chunks l, l+16, ... (4 independent loads in flight: the partials
// come from L2 and a lane's loads are otherwise a dependent latency chain - 188 us for 1024 chunks with 8 lanes), then the 16
// lanes are added in order.  Fixed order -> deterministic.
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int nchunks, int F, float* __restrict__ out,
                                                      int accumulate) {
    __shared__ float sm[16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < F) {
        int k = kl;
        for (; k + 48 < nchunks; k += 64) {
            s0 += partial[(size_t)k * F + c];
            s1 += partial[(size_t)(k + 16) * F + c];
            s2 += partial[(size_t)(k + 32) * F + c];
            s3 += partial[(size_t)(k + 48) * F + c];
        }
        for (; k < nchunks; k += 16) s0 += partial[(size_t)k * F + c];
    }
    sm[kl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (kl == 0 && c < F) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// scalars != NULL: lr_t is read from the ChamStepScalars record (device) instead of the argument
static int adam_tf_impl(float* params, const float* grads, float* m, float* v, size_t n, size_t n_reg, float lambda,
                        float lr_t, float beta1, float beta2, float eps, const void* scalars, void* stream) {
    if (!params || !grads || !m || !v || (n & 3) || (n_reg & 3) || n_reg > n) return -CHAM_ERR_ARG;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) return CHAM_OK;
    hipLaunchKernelGGL(k_adam_tf, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n4, n_reg / 4,
                       lambda, lr_t, beta1, beta2, eps, reinterpret_cast<const ChamStepScalars*>(scalars));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_adam_tf(float* params, const float* grads, float* m, float* v, size_t n, size_t n_reg, float lambda,
                            float lr_t, float beta1, float beta2, float eps, void* stream) {
    return adam_tf_impl(params, grads, m, v, n, n_reg, lambda, lr_t, beta1, beta2, eps, nullptr, stream);
}
extern "C" int cham_adam_tf_dev(float* params, const float* grads, float* m, float* v, size_t n, size_t n_reg, float lambda,
                                const void* scalars, float beta1, float beta2, float eps, void* stream) {
    if (!scalars) return -CHAM_ERR_ARG;
    return adam_tf_impl(params, grads, m, v, n, n_reg, lambda, 0.f, beta1, beta2, eps, scalars, stream);
}

// ---- the step-scalar record (common.h ChamStepScalars): written by a one-thread kernel that takes the values BY VALUE (a kernel's
// parameters are copied at launch: no host staging buffer to keep alive), stream-ordered in front of the step that reads them
__global__ void k_step_scalars_set(ChamStepScalars* __restrict__ rec, uint32_t step, uint32_t step_next, int64_t max_ts, float sum_mask, float lr_t,
                                   int fields) {
    if (fields & 1) { rec->step = step; rec->step_next = step_next; }
    if (fields & 2) { rec->max_ts = max_ts; rec->sum_mask = sum_mask; }
    if (fields & 4) rec->lr_t = lr_t;
}
// fields: bit 0 = the sampler keys, bit 1 = the batch scalars (max_ts, sum_mask), bit 2 = lr_t (the others are left as they are)
extern "C" int cham_step_scalars_set(void* rec, uint32_t step, uint32_t step_next, int64_t max_ts, float sum_mask, float lr_t, int fields, void* stream) {
    if (!rec || ((uintptr_t)rec & 7) || !(fields & 7)) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_step_scalars_set, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<ChamStepScalars*>(rec), step, step_next, max_ts,
                       sum_mask, lr_t, fields);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_step_scalars_bytes(void) { return (int)sizeof(ChamStepScalars); }

extern "C" int cham_sumsq_partial(const float* params, size_t n_reg, float* partial /*[1024]*/, void* stream) {
    if (!params || !partial || (n_reg & 3)) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_sumsq_partial, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, params, n_reg / 4, partial);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_loss_finalize(const float* nll, int BT, float sum_mask, const float* sumsq_partial, float lambda,
                                  float* loss /*[3]*/, void* stream) {
    if (!nll || !sumsq_partial || !loss || BT <= 0 || sum_mask <= 0.f) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, BT, 1.0f / sum_mask, sumsq_partial,
                       SUMSQ_BLOCKS, lambda, loss, (const ChamStepScalars*)nullptr);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
// sum(mask) from the ChamStepScalars record (device)
extern "C" int cham_loss_finalize_dev(const float* nll, int BT, const void* scalars, const float* sumsq_partial, float lambda,
                                      float* loss /*[3]*/, void* stream) {
    if (!nll || !sumsq_partial || !loss || BT <= 0 || !scalars) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, BT, 0.f, sumsq_partial,
                       SUMSQ_BLOCKS, lambda, loss, reinterpret_cast<const ChamStepScalars*>(scalars));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" size_t cham_colsum_workspace_bytes(int R, int F) {
    int rpc = (R + 1023) / 1024; if (rpc < 64) rpc = 64;
    const int nchunks = (R + rpc - 1) / rpc;
    return (size_t)nchunks * F * sizeof(float);
}

extern "C" int cham_colsum(const float* X, int ld, int R, int F, const float* w, float* out, int accumulate,
                           float* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !out || !workspace || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    if (workspace_bytes < cham_colsum_workspace_bytes(R, F)) return -CHAM_ERR_ARG;
    int rpc = (R + 1023) / 1024; if (rpc < 64) rpc = 64;
    const int nchunks = (R + rpc - 1) / rpc;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (F % 4 == 0) && (F / 4 <= 256) && (256 % (F / 4) == 0) && (ld % 4 == 0);
    if (vec) hipLaunchKernelGGL(k_colsum_vec<float>, dim3(nchunks), dim3(256), 0, st, X, ld, R, F, w, rpc, workspace);
    else hipLaunchKernelGGL(k_colsum_gen, dim3(nchunks), dim3(256), 0, st, X, ld, R, F, w, rpc, workspace);
    hipLaunchKernelGGL(k_colsum_final, dim3((F + 15) / 16), dim3(256), 0, st, workspace, nchunks, F, out, accumulate);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
// bf16 configuration: column sums of a bf16 matrix (bias gradients over the candidate rows); F % 4 == 0, F <= 1024, 256 % (F/4) == 0
extern "C" int cham_colsum_b16(const void* X, int ld, int R, int F, const float* w, float* out, int accumulate,
                               float* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !out || !workspace || R <= 0 || F <= 0) return -CHAM_ERR_ARG;
    if (workspace_bytes < cham_colsum_workspace_bytes(R, F)) return -CHAM_ERR_ARG;
    if ((F % 4) || (F / 4 > 256) || (256 % (F / 4)) || (ld % 4)) return -CHAM_ERR_ARG;
    int rpc = (R + 1023) / 1024; if (rpc < 64) rpc = 64;
    const int nchunks = (R + rpc - 1) / rpc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_colsum_vec<__bf16>, dim3(nchunks), dim3(256), 0, st, reinterpret_cast<const __bf16*>(X), ld, R, F, w, rpc, workspace);
    hipLaunchKernelGGL(k_colsum_final, dim3((F + 15) / 16), dim3(256), 0, st, workspace, nchunks, F, out, accumulate);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// bf16 shadow of a weight matrix W [R, Cc] (fp32 master copy): dst = bf16(W) [R, Cc] and / or dstT = bf16(W)^T [Cc, R] - the
// k-contiguous operand forms of the forward (W^T) and dgrad (W) GEMMs of the bf16 configuration.  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void k_cast_b16(const float* __restrict__ W, int R, int Cc, __bf16* __restrict__ dst, __bf16* __restrict__ dstT) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < Cc) ? W[(size_t)r * Cc + c] : 0.f;
        t[i][tx] = v;
        if (dst && r < R && c < Cc) dst[(size_t)r * Cc + c] = (__bf16)v;
    }
    __syncthreads();
    if (dstT)
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, r = r0 + tx;
            if (r < R && c < Cc) dstT[(size_t)c * R + r] = (__bf16)t[tx][i];
        }
}
extern "C" int cham_cast_b16(const float* W, int R, int Cc, void* dst, void* dstT, void* stream) {
    if (!W || (!dst && !dstT) || R <= 0 || Cc <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_cast_b16, dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, W, R, Cc,
                       reinterpret_cast<__bf16*>(dst), reinterpret_cast<__bf16*>(dstT));
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// bf16 -> fp32 copy of a contiguous array (n % 4 == 0): the bf16 configuration's dropout path hands the bf16-resident gradient of the
// PreCAR output to the dense (fp32-storage) PreCAR backward.  Exact.
__global__ __launch_bounds__(256) void k_upcast_b16(const __bf16* __restrict__ src, size_t n4, float* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4(dst + 4 * i, ld4(src + 4 * i));
}
extern "C" int cham_upcast_b16(const void* src, size_t n, float* dst, void* stream) {
    if (!src || !dst || (n & 3)) return -CHAM_ERR_ARG;
    if (n == 0) return CHAM_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_upcast_b16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(src), n / 4, dst);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// ---- gradient accumulation over session micro-batches (large-catalog / large-batch configurations): the step's row
// shards share the pool, the denominators and the weights, so their gradient buffers simply add up.
__global__ __launch_bounds__(256) void k_accumulate(float* __restrict__ acc, const float* __restrict__ x, size_t n4, size_t n, int first) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 b = reinterpret_cast<const float4*>(x)[i];
        float4 a = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(acc)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(acc)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        acc[i] = (first ? 0.f : acc[i]) + x[i];
    }
}
// loss_acc = [xe_acc + reg, xe_acc (+)= xe, reg]
__global__ void k_loss_accumulate(float* __restrict__ acc, const float* __restrict__ loss, int first) {
    if (threadIdx.x == 0) {
        const float xe = (first ? 0.f : acc[1]) + loss[1];
        acc[1] = xe; acc[2] = loss[2]; acc[0] = xe + loss[2];
    }
}
extern "C" int cham_accumulate(float* acc, const float* x, size_t n, int first, void* stream) {
    if (!acc || !x || n == 0) return -CHAM_ERR_ARG;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, acc, x, n4, n, first);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_loss_accumulate(float* acc, const float* loss, int first, void* stream) {
    if (!acc || !loss) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_loss_accumulate, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, loss, first);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
