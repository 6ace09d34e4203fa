// UGRNN time steps with the recurrent weights RESIDENT IN LDS, eight cooperating workgroups per 32 sessions (round 4; gfx950, wave64).
//
// Replaces, for short steps, k_ugrnn_fwd / k_ugrnn_bwd of rnn.hip (nar_module/nar/nar_model.py:1308-1361 of the reference:
// tf.contrib.rnn.UGRNNCell under tf.nn.dynamic_rnn(sequence_length); same outputs, same saved activations).
//
// Why: rnn.hip gives one workgroup 32 sessions for all T steps and streams W_h (512 KB fp32 at H = 256) from L2 through registers
// EVERY time step: 32-34 us per step against a 13.7 us fp32-MFMA floor on one CU, 0.64 + 0.58 ms per training step whatever the batch -
// hidden behind the candidate-row GEMMs of a full-length 256-session batch, but the critical path of a ragged (G1-like) batch and of
// the 32-session shard a rank gets under strong scaling (profiles/r03_emulated_rank_of_n.txt: 2.8 ms, 1.2 of them this chain).
// Here the 2 H gate / candidate columns of a group of 32 sessions are cut into EIGHT slices of 32 hidden units; slice q's workgroup keeps
// its 64 columns of W_h (forward) / its 32 rows of W_h (= columns of W_h^T, backward) in LDS for the whole sequence (64 KB), multiplies
// on v_mfma_f32_32x32x2_f32 with the K range split over its four waves (64 MFMAs per wave and step: ~1.8 us), and the eight
// workgroups of a group exchange their 32 x 32 slice of h_t (forward) / of [dz_g | dz_c] (backward) through L2 every step:
//     write-through (sc1) 16-byte stores -> every wave s_waitcnt vmcnt(0) -> __syncthreads -> ONE lane stores flag[group][q] = t + 1
//     (relaxed, agent scope) | ONE wave polls the group's eight flags (relaxed) -> ONE agent-scope acquire -> __syncthreads -> plain loads
// (cdna_hip_programming.md Guideline 16, recipe R1; the payload buffers are double-buffered by step parity - a workgroup can be at
// most one step ahead of the slowest of its group).  Flags and the time-out word are zeroed by a hipMemsetAsync before every launch;
// every spin is bounded (a workgroup that is never scheduled beside its group - it cannot happen with <= 64 workgroups on 256 CUs -
// sets the time-out word instead of hanging the device).
// LDS per workgroup: forward 114 KB (W slice 66 KB, h_{t-1} 33 KB, partial sums 16 KB), backward 153 KB: it does not share a CU with a
// plane-GEMM workgroup (128 KB) - the caller uses these kernels where the recurrent chain is the critical path (nar_model.py: ragged /
// short steps) and the single-workgroup kernels beside the big GEMMs of a full batch.
#include "common.h"

#define RC_SLICES 8
#define RC_PF 132                    // forward fragment row pitch (floats): 128 k-pairs + 4 -> conflict-free ds_read_b128
#define RC_PB 260                    // backward: 256 k-pairs + 4
#define RC_ROW(e, kl) (((e) & 3) + 8 * ((e) >> 2) + 4 * (kl))
#define RC_SPIN_MAX (1u << 21)

typedef __attribute__((address_space(1))) unsigned rc_gu32;
typedef unsigned int rc_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rc_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rc_rsrc(void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)bytes, 0x00020000);
}

// publish: every storing wave has drained its sc1 stores, ONE lane stores the epoch; then ONE wave waits for the eight flags of the
// group, ONE acquire; returns after a __syncthreads (every thread may issue plain loads of the group's payload)
__device__ __forceinline__ void rc_exchange(unsigned* flags, int q, unsigned epoch, unsigned* tmo) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((rc_gu32*)(flags + q), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned spins = 0;
        for (;;) {
            unsigned v = epoch;
            if (lane < RC_SLICES) v = __hip_atomic_load((rc_gu32*)(flags + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v >= epoch))) break;
            if (++spins > RC_SPIN_MAX) {          // bounded: report and go on (the results are garbage, the device is not hung)
                if (lane == 0) __hip_atomic_store((rc_gu32*)tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// grid (8 slices, groups of 32 sessions), 256 threads.  hx: [2][groups][32][256] fp32; flags: [groups][8]
__global__ __launch_bounds__(256) void k_ugrnn_fwd_coop(const float* __restrict__ xproj, const float* __restrict__ Wh,
                                                        const int* __restrict__ seq_len, int B, int T, float* __restrict__ out,
                                                        float* __restrict__ hprev, float* __restrict__ G, float* __restrict__ Cc,
                                                        float* hx, unsigned* flags, unsigned* tmo) {
    constexpr int Hp = 256, H2 = 512, P = RC_PF;
    extern __shared__ __attribute__((aligned(16))) float rc_sm[];
    float* Wl = rc_sm;                          // [tile 0 gate | 1 candidate][k parity][32 columns][P]: element (k, col) at [k & 1][col][k >> 1]
    float* hA = Wl + 2 * 2 * 32 * P;            // [k parity][32 rows][P]: h_{t-1}
    float* zs = hA + 2 * 32 * P;                // [4 waves][16][64]: partial accumulators
    float* hT = zs + 4 * 16 * 64;               // [32 rows][36]: this slice's new h
    const int q = blockIdx.x, grp = blockIdx.y, ng = gridDim.y;
    const int b0 = grp * 32, hs0 = 32 * q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fl = lane & 31, kl = lane >> 5;
    for (int i = tid; i < 2 * 256 * 32; i += 256) {
        const int tt = i / (256 * 32), k = (i / 32) % 256, c = i % 32;
        Wl[((tt * 2 + (k & 1)) * 32 + c) * P + (k >> 1)] = Wh[(size_t)k * H2 + tt * Hp + hs0 + c];
    }
    for (int i = tid; i < 2 * 32 * P; i += 256) hA[i] = 0.f;
    int slv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int b = b0 + RC_ROW(4 * wave + i, kl); slv[i] = b < B ? seq_len[b] : 0; }
    __syncthreads();
    unsigned* gflags = flags + (size_t)grp * RC_SLICES;
    const int tt = wave & 1, kh = wave >> 1;
    const float* ap = hA + (kl * 32 + fl) * P + 64 * kh;
    const float* bp = Wl + ((tt * 2 + kl) * 32 + fl) * P + 64 * kh;
    const int u = hs0 + fl;
    for (int t = 0; t < T; ++t) {
        float xg[4], xc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + RC_ROW(4 * wave + i, kl);
            const size_t o = ((size_t)(b < B ? b : 0) * T + t) * H2 + u;
            xg[i] = xproj[o]; xc[i] = xproj[o + Hp];
        }
        floatx16 acc, acc1;                       // two accumulation chains (even / odd k-pairs): no MFMA waits for its predecessor's result
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int p4 = 0; p4 < 16; ++p4) {
            const float4 a = *reinterpret_cast<const float4*>(ap + 4 * p4);
            const float4 b = *reinterpret_cast<const float4*>(bp + 4 * p4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) zs[(wave * 16 + e) * 64 + lane] = acc[e] + acc1[e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * wave + i, row = RC_ROW(e, kl), b = b0 + row;
            // fixed order: K half 0 + K half 1 + input projection
            const float zg = (zs[(0 * 16 + e) * 64 + lane] + zs[(2 * 16 + e) * 64 + lane]) + xg[i];
            const float zc = (zs[(1 * 16 + e) * 64 + lane] + zs[(3 * 16 + e) * 64 + lane]) + xc[i];
            const float g = rc_sigmoid(zg + 1.0f);           // forget_bias = 1.0
            const float c = cham_tanhf(zc);
            const float ho = hA[((u & 1) * 32 + row) * P + (u >> 1)];
            const float hn = g * ho + (1.f - g) * c;
            const bool valid = t < slv[i];
            if (b < B) {
                const size_t o = ((size_t)b * T + t) * Hp + u;
                out[o] = valid ? hn : 0.f; hprev[o] = ho; G[o] = g; Cc[o] = c;
            }
            hT[row * 36 + fl] = valid ? hn : ho;
        }
        if (t + 1 == T) break;                    // nobody reads the last state
        __syncthreads();
        float* hxb = hx + ((size_t)(t & 1) * ng + grp) * 32 * Hp;
        {   // publish this slice: 32 rows x 32 floats = 256 x 16 bytes, write-through
            const int row = tid >> 3, c4 = tid & 7;
            const float4 v = *reinterpret_cast<const float4*>(hT + row * 36 + 4 * c4);
            rc_u32x4 w; w.x = __float_as_uint(v.x); w.y = __float_as_uint(v.y); w.z = __float_as_uint(v.z); w.w = __float_as_uint(v.w);
            __builtin_amdgcn_raw_buffer_store_b128(w, rc_rsrc(hxb, (size_t)32 * Hp * 4), (unsigned)((row * Hp + hs0 + 4 * c4) * 4), 0, 16);
        }
        rc_exchange(gflags, q, (unsigned)(t + 1), tmo);
#pragma unroll
        for (int j = 0; j < 8; ++j) {             // the group's h_t -> fragment layout
            const int idx = tid + 256 * j, row = idx >> 6, c4 = idx & 63;
            const float4 v = *reinterpret_cast<const float4*>(hxb + row * Hp + 4 * c4);
            *reinterpret_cast<float2*>(hA + (0 * 32 + row) * P + 2 * c4) = make_float2(v.x, v.z);
            *reinterpret_cast<float2*>(hA + (1 * 32 + row) * P + 2 * c4) = make_float2(v.y, v.w);
        }
        __syncthreads();
    }
}

// backward through time.  dout [B,T,Hp]; Wh [Hp, 2 Hp] AS STORED (slice q needs its 32 rows = columns of W_h^T);
// writes dxproj [B,T,2Hp].  dzx: [2][groups][32][512] fp32; flags: [groups][8]
__global__ __launch_bounds__(256) void k_ugrnn_bwd_coop(const float* __restrict__ dout, const float* __restrict__ Wh,
                                                        const int* __restrict__ seq_len, int B, int T, const float* __restrict__ hprev,
                                                        const float* __restrict__ G, const float* __restrict__ Cc,
                                                        float* __restrict__ dxproj, float* dzx, unsigned* flags, unsigned* tmo) {
    constexpr int Hp = 256, H2 = 512, P = RC_PB;
    extern __shared__ __attribute__((aligned(16))) float rc_sm[];
    float* Wb = rc_sm;                          // [k parity][32 columns = this slice's hidden units][P]: W_h^T[k][u] = W_h[u][k]
    float* dzA = Wb + 2 * 32 * P;               // [k parity][32 rows][P]: [dz_g | dz_c] of ALL hidden units
    float* zs = dzA + 2 * 32 * P;               // [4 waves][16][64]
    float* dT = zs + 4 * 16 * 64;               // [32 rows][68]: this slice's dz_g (columns 0..31) | dz_c (32..63)
    const int q = blockIdx.x, grp = blockIdx.y, ng = gridDim.y;
    const int b0 = grp * 32, hs0 = 32 * q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fl = lane & 31, kl = lane >> 5;
    for (int i = tid; i < 32 * H2; i += 256) {
        const int c = i / H2, k = i % H2;
        Wb[((k & 1) * 32 + c) * P + (k >> 1)] = Wh[(size_t)(hs0 + c) * H2 + k];
    }
    int slv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int b = b0 + RC_ROW(4 * wave + i, kl); slv[i] = b < B ? seq_len[b] : 0; }
    float carry[4] = {0.f, 0.f, 0.f, 0.f};       // dL/dh_t flowing to step t, for (row(4 wave + i), unit hs0 + fl)
    __syncthreads();
    unsigned* gflags = flags + (size_t)grp * RC_SLICES;
    const float* ap = dzA + (kl * 32 + fl) * P + 64 * wave;          // wave w: k-pairs [64 w, 64 w + 64)
    const float* bp = Wb + (kl * 32 + fl) * P + 64 * wave;
    const int u = hs0 + fl;
    for (int t = T - 1; t >= 0; --t) {
        float direct[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * wave + i, row = RC_ROW(e, kl), b = b0 + row;
            float dzg = 0.f, dzc = 0.f, dd = 0.f;
            if (t < slv[i]) {
                const size_t o = ((size_t)b * T + t) * Hp + u;
                const float dh = dout[o] + carry[i];
                const float g = G[o], c = Cc[o], hp = hprev[o];
                dzg = dh * (hp - c) * g * (1.f - g);
                dzc = dh * (1.f - g) * (1.f - c * c);
                dd = dh * g;
            }
            direct[i] = dd;
            dT[row * 68 + fl] = dzg;
            dT[row * 68 + 32 + fl] = dzc;
            if (b < B) {
                const size_t o2 = ((size_t)b * T + t) * H2;
                dxproj[o2 + u] = dzg;
                dxproj[o2 + Hp + u] = dzc;
            }
        }
        if (t == 0) break;                        // nothing flows out of the first step
        __syncthreads();
        float* dzb = dzx + ((size_t)(t & 1) * ng + grp) * 32 * H2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {             // publish: 32 rows x 64 floats = 512 x 16 bytes, write-through
            const int idx = tid + 256 * j, row = idx >> 4, c4 = idx & 15;
            const float4 v = *reinterpret_cast<const float4*>(dT + row * 68 + 4 * c4);
            rc_u32x4 w; w.x = __float_as_uint(v.x); w.y = __float_as_uint(v.y); w.z = __float_as_uint(v.z); w.w = __float_as_uint(v.w);
            const int col = (c4 < 8 ? 0 : Hp - 32) + hs0 + 4 * c4;              // dz_g -> column u, dz_c -> column Hp + u
            __builtin_amdgcn_raw_buffer_store_b128(w, rc_rsrc(dzb, (size_t)32 * H2 * 4), (unsigned)((row * H2 + col) * 4), 0, 16);
        }
        rc_exchange(gflags, q, (unsigned)(T - t), tmo);
#pragma unroll
        for (int j = 0; j < 16; ++j) {            // the group's dz -> fragment layout
            const int idx = tid + 256 * j, row = idx >> 7, c4 = idx & 127;
            const float4 v = *reinterpret_cast<const float4*>(dzb + row * H2 + 4 * c4);
            *reinterpret_cast<float2*>(dzA + (0 * 32 + row) * P + 2 * c4) = make_float2(v.x, v.z);
            *reinterpret_cast<float2*>(dzA + (1 * 32 + row) * P + 2 * c4) = make_float2(v.y, v.w);
        }
        __syncthreads();
        floatx16 acc, acc1;                       // two accumulation chains (even / odd k-pairs): no MFMA waits for its predecessor's result
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int p4 = 0; p4 < 16; ++p4) {
            const float4 a = *reinterpret_cast<const float4*>(ap + 4 * p4);
            const float4 b = *reinterpret_cast<const float4*>(bp + 4 * p4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) zs[(wave * 16 + e) * 64 + lane] = acc[e] + acc1[e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * wave + i;
            const float s = ((zs[(0 * 16 + e) * 64 + lane] + zs[(1 * 16 + e) * 64 + lane]) + zs[(2 * 16 + e) * 64 + lane]) + zs[(3 * 16 + e) * 64 + lane];
            carry[i] = (t < slv[i]) ? (direct[i] + s) : carry[i];
        }
        __syncthreads();                          // zs / dT / dzA are rewritten by the next step
    }
}

// exchange buffers [2][groups][32][2 Hp] fp32 (the forward uses half) + flags [groups][8] + the time-out word, 256-byte aligned pieces
extern "C" size_t cham_rnn_coop_workspace_bytes(int B, int Hp) {
    if (B <= 0 || Hp != 256) return 0;
    const size_t ng = (size_t)(B + 31) / 32;
    return 2 * ng * 32 * 2 * Hp * sizeof(float) + ((ng * RC_SLICES * sizeof(unsigned) + 255) & ~(size_t)255) + 256;
}

static int rc_prepare(int B, int Hp, void* workspace, size_t workspace_bytes, hipStream_t st, float** payload, unsigned** flags, unsigned** tmo, int* ng) {
    if (Hp != 256 || B <= 0 || !workspace || ((uintptr_t)workspace & 255) || workspace_bytes < cham_rnn_coop_workspace_bytes(B, Hp)) return -CHAM_ERR_ARG;
    *ng = (B + 31) / 32;
    // every workgroup of a launch must be resident at once (one per CU: 114 / 153 KB of LDS each): the CU count of the CURRENT device,
    // queried once per device - not an assumed 256; -EINVAL keeps the caller on cham_rnn_fwd / _bwd
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -CHAM_ERR_LAUNCH;
    int n_cu = cus[dev & 63].load(std::memory_order_relaxed);
    if (n_cu == 0) {
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return -CHAM_ERR_LAUNCH;
        cus[dev & 63].store(n_cu, std::memory_order_relaxed);
    }
    if (*ng * RC_SLICES > n_cu) return -CHAM_ERR_ARG;
    const size_t pay = 2 * (size_t)*ng * 32 * 2 * Hp * sizeof(float);
    *payload = reinterpret_cast<float*>(workspace);
    *flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + pay);
    const size_t fl = ((size_t)*ng * RC_SLICES * sizeof(unsigned) + 255) & ~(size_t)255;
    *tmo = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + pay + fl);
    // flags zeroed before EVERY launch (epochs count within the call); the time-out word is sticky: cham_rnn_coop_timeouts reads it
    if (hipMemsetAsync(*flags, 0, fl, st) != hipSuccess) return -CHAM_ERR_LAUNCH;
    return CHAM_OK;
}

// UGRNN forward / backward for Hp == 256 on eight cooperating workgroups per 32 sessions - same contract as cham_rnn_fwd / cham_rnn_bwd
// with cell_kind 0 (the backward takes W_h AS STORED, not its transpose).  Results equal the single-workgroup kernels' up to the fp32
// summation order of the recurrent product (K in two / four fixed pieces).  Returns -EINVAL for Hp != 256, B > 1024 or a short workspace
// (the caller keeps cham_rnn_fwd / _bwd).  `workspace`: cham_rnn_coop_workspace_bytes(B, Hp) bytes, 256-byte aligned, zero-initialised
// once; one workspace per stream (a forward and a backward on the same stream may share it).
extern "C" int cham_ugrnn_fwd_coop(const float* xproj, const float* Wh, const int32_t* seq_len, int B, int T, int Hp, float* out,
                                   float* hprev, float* G, float* Cc, void* workspace, size_t workspace_bytes, void* stream) {
    if (!xproj || !Wh || !seq_len || !out || !hprev || !G || !Cc || T <= 0) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* pay; unsigned *flags, *tmo; int ng;
    const int rc = rc_prepare(B, Hp, workspace, workspace_bytes, st, &pay, &flags, &tmo, &ng);
    if (rc != CHAM_OK) return rc;
    constexpr int smem = (2 * 2 * 32 * RC_PF + 2 * 32 * RC_PF + 4 * 16 * 64 + 32 * 36) * 4;
    CHAM_SET_DYNAMIC_LDS(k_ugrnn_fwd_coop, smem);
    hipLaunchKernelGGL(k_ugrnn_fwd_coop, dim3(RC_SLICES, ng), dim3(256), smem, st, xproj, Wh, seq_len, B, T, out, hprev, G, Cc, pay, flags, tmo);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

extern "C" int cham_ugrnn_bwd_coop(const float* dout, const float* Wh, const int32_t* seq_len, int B, int T, int Hp, const float* hprev,
                                   const float* G, const float* Cc, float* dxproj, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !Wh || !seq_len || !hprev || !G || !Cc || !dxproj || T <= 0) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* pay; unsigned *flags, *tmo; int ng;
    const int rc = rc_prepare(B, Hp, workspace, workspace_bytes, st, &pay, &flags, &tmo, &ng);
    if (rc != CHAM_OK) return rc;
    constexpr int smem = (2 * 32 * RC_PB + 2 * 32 * RC_PB + 4 * 16 * 64 + 32 * 68) * 4;
    CHAM_SET_DYNAMIC_LDS(k_ugrnn_bwd_coop, smem);
    hipLaunchKernelGGL(k_ugrnn_bwd_coop, dim3(RC_SLICES, ng), dim3(256), smem, st, dout, Wh, seq_len, B, T, hprev, G, Cc, dxproj, pay, flags, tmo);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// 1 if any workgroup of any launch on this workspace gave up a bounded spin (device-to-host copy: synchronises the stream)
extern "C" int cham_rnn_coop_timeouts(const void* workspace, int B, int Hp, void* stream) {
    if (!workspace || Hp != 256 || B <= 0) return -CHAM_ERR_ARG;
    const size_t ng = (size_t)(B + 31) / 32;
    const size_t off = 2 * ng * 32 * 2 * Hp * sizeof(float) + ((ng * RC_SLICES * sizeof(unsigned) + 255) & ~(size_t)255);
    unsigned v = 0;
    if (hipMemcpyAsync(&v, reinterpret_cast<const char*>(workspace) + off, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return -CHAM_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -CHAM_ERR_LAUNCH;
    return (int)v;
}
