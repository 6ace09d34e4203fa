// Shared device helpers for the CHAMELEON NAR kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CHAM_OK 0
#define CHAM_ERR_ARG 22       /* EINVAL */
#define CHAM_ERR_LAUNCH 5     /* EIO    */

#define CHAM_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return -CHAM_ERR_LAUNCH;      \
    } while (0)

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER DEVICE (the attribute
// lives with the device's code object): `mask` is a per-kernel-instance bit set of the device ids that have it, atomically updated
// (two host threads may launch the same kernel; a second device of the same process gets its own call - ADVICE r04).
#include <atomic>
static inline int cham_set_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -CHAM_ERR_LAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return CHAM_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -CHAM_ERR_LAUNCH;
    mask.fetch_or(bit, std::memory_order_release);
    return CHAM_OK;
}
#define CHAM_SET_DYNAMIC_LDS(kernel, bytes)                                                                  \
    do {                                                                                                     \
        static std::atomic<unsigned long long> lds_mask__{0};                                                \
        if (cham_set_dynamic_lds(reinterpret_cast<const void*>(kernel), (bytes), lds_mask__) != CHAM_OK)     \
            return -CHAM_ERR_LAUNCH;                                                                         \
    } while (0)

// ---- STEP SCALARS in device memory (round 6).  The launch parameters of a step that change from one optimizer step to the next - the
// sampler key, the batch's max time stamp, sum(mask), Adam's bias-corrected learning rate - live in ONE 32-byte device record written by
// cham_step_scalars_set (a one-thread kernel taking them by value).  The `*_dev` entry points read them from the record instead of taking
// them by value, so that a step's launches carry no per-step host value: the step can be captured in a hipGraph once and replayed
// (reference: one session.run per step, nar_model.py:1434-1470).  Same arithmetic on the same values: results are bit-identical to the
// by-value entry points.
struct ChamStepScalars {
    uint32_t step;            // sampler key of this step (global_step, or the evaluation key)
    uint32_t step_next;       // ... of the NEXT step (presampling of the next batch's negatives)
    int64_t max_ts;           // max event time stamp of the (global) batch, nar_model.py:235
    int64_t max_ts_next;      // reserved
    float sum_mask;           // sum(mask) of the (global) batch: the loss denominator, nar_model.py:664
    float lr_t;               // lr * sqrt(1 - b2^t) / (1 - b1^t), tf.train.AdamOptimizer
};

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_TANH = 2 };

// ln of the two smoothing-log bases of the item features (nar_model.py:122-123 elapsed_days_smooth_log_base, popularity_smooth_log_base;
// log_base :28-34), launch scalars of the kernels that use them; set by cham_set_log_bases (csrc/features.hip), defaults 1.3 / 2.0.
extern thread_local float g_cham_ln_elapsed_base, g_cham_ln_pop_base, g_cham_inv_log2_pop_base;      // per host thread; (the last: 1 / log2(base), exactly 1 for base 2)

// tanh for the GEMM epilogues: branch-free, ~17 VALU (the libdevice tanhf is ~45 with divergent range branches; VALU issue
// slots are MFMA issue slots, the tanh epilogue cost 8 % of the CAR layer-2 GEMM).  |x| < 0.625: odd minimax polynomial
// x * P5(x^2) (max rel. error 1.0e-7); otherwise 1 - 2 / (exp(2|x|) + 1) on v_exp_f32 / v_rcp_f32 (1.5e-7): fp32-roundoff class.
__device__ __forceinline__ float cham_tanhf(float x) {
    const float ax = fabsf(x), u = x * x;
    float p = -0.005664775148034096f;
    p = fmaf(p, u, 0.020595679059624672f);
    p = fmaf(p, u, -0.05372267961502075f);
    p = fmaf(p, u, 0.13331149518489838f);
    p = fmaf(p, u, -0.3333326280117035f);
    p = fmaf(p, u, 1.0f);
    p *= x;
    const float e = __expf(2.f * ax);
    const float t = fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
    return ax < 0.625f ? p : copysignf(t, x);
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == ACT_TANH) return cham_tanhf(v);
    return v;
}
// derivative expressed through the saved POST-activation value y
__device__ __forceinline__ float act_bwd_from_out(float y, int act) {
    if (act == ACT_LEAKY) return y > 0.f ? 1.f : 0.2f;
    if (act == ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// ---- Philox4x32-10, word 0 (contract: oracle/philox.py) -------------------------------------
__device__ __forceinline__ uint32_t philox_rand32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
#define CHAM_INF_KEY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ uint64_t philox_sort_key(uint32_t q, uint32_t j, uint32_t b, uint32_t stage,
                                                    uint32_t seed, uint32_t step) {
    return ((uint64_t)philox_rand32(q, j, b, stage, seed, step) << 32) | (uint64_t)q;
}

// ---- 4 consecutive elements of a row as float4: fp32 rows (16-byte access) or bf16 rows (8-byte access, the bf16 configuration's
// candidate-row matrices).  bf16 -> fp32 is exact; fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32).
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void st4(__bf16* p, const float4& v) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t b; b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = b;
}

// a = h + m + l with h = bf16(a), m = bf16(a - h), l = bf16(a - h - m) (round to nearest even; the two subtractions are exact in fp32):
// three 8-bit significands cover fp32's 24, so the sum is exact up to the last bit (gemm_x3.hip, gemm_p3.hip; tests/test_split3_cpu.py)
// (The compiler SLP-packs pairs of the subtractions into v_pk_add_f32, a costly filler beside MFMAs per MI355X_MICROARCH.md; forcing
// scalar v_sub_f32 through inline asm also un-pairs the v_cvt_pk_bf16_f32 conversions and measured 1-3 % slower: profiles/r02_notes.md.)
__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    float r = a - (float)h;
    m = (__bf16)r;
    r -= (float)m;
    l = (__bf16)r;
}

// 4 consecutive elements of a row -> the three planes of a plane-resident matrix (planes `ps` elements apart): 8-byte stores
__device__ __forceinline__ void st4_planes(__bf16* p, long long ps, const float4& v) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t h, m, l;
    __bf16 a, b, c;
    split3(v.x, a, b, c); h[0] = a; m[0] = b; l[0] = c;
    split3(v.y, a, b, c); h[1] = a; m[1] = b; l[1] = c;
    split3(v.z, a, b, c); h[2] = a; m[2] = b; l[2] = c;
    split3(v.w, a, b, c); h[3] = a; m[3] = b; l[3] = c;
    *reinterpret_cast<bf16x4_t*>(p) = h;
    *reinterpret_cast<bf16x4_t*>(p + ps) = m;
    *reinterpret_cast<bf16x4_t*>(p + 2 * ps) = l;
}

// ---- two fp16 planes + a power-of-two scale ("h2"; csrc/gemm_h2.hip, round 4) ----------------------------------------------------
// x = (h + l) / s with h = fp16(x s), l = fp16(x s - h) (round to nearest even; x s and the subtraction are exact in fp32): 11 + 11
// significand bits while l is a normal fp16 number (|x s| >= 2^-3), absolute error <= 2^-25 / s below that (fp16 subnormals are kept:
// float_denorm_mode_16_64 = 3).  s is an exact power of two chosen by the matrix's producer from a rigorous bound of max |x| so that
// |x s| <= 2^15 (fp16 max 65 504): see k_h2_scale_* in gemm_h2.hip.  A plane-product GEMM over such operands needs THREE MFMA products
// (h h + h l + l h; the dropped l l term is <= 2^-22 |a b|) where the bf16 x3 split needs six (tests/test_split2h_cpu.py).
// The h plane keeps the SIGN of a value that underflows: x > 0 never stores +0 (smallest subnormal instead), so a consumer may take
// leaky-ReLU's derivative from the h plane's bit pattern alone ((short) bits > 0 <=> x > 0).
struct H2Scale { float scale, inv, bound, pad; unsigned max_bits, ticket, pad1, pad2; };     // 32-byte device record (zero-initialised)
__device__ __forceinline__ void split2h(float xs /* already scaled */, _Float16& h, _Float16& l) {
    h = (_Float16)xs;
    l = (_Float16)(xs - (float)h);
}
__device__ __forceinline__ unsigned short h2_keep_sign(_Float16 h, float x) {
    const unsigned short b = __builtin_bit_cast(unsigned short, h);
    return (x > 0.f && b == 0) ? (unsigned short)1 : b;
}
// ---- TILE-BLOCKED plane layout (round 6; csrc/gemm_h2.hip "blocked operands"): a [rows, ld] matrix of 16-bit elements stored as
//   [row tiles of 256][column blocks of 32][256 rows][32 columns]
// i.e. a (256-row x 32-column) block is 16 KB contiguous, a row's 32 columns 64 bytes, 16 consecutive rows of a block 1 KB.  What the NT
// plane GEMMs stage per request (16 rows x 32 k) is then 1 KB of whole 128-byte lines instead of sixteen half lines; the TN form's
// request (2 k-rows x 256 m) is eight whole lines as before; the 16-byte pieces both kernels move are the same element sets, so their LDS
// images, products and results are bit-identical to the row-major operands'.  ld % 32 == 0; the rows of the last tile beyond the
// matrix are allocated and ZERO (the TN form contracts over rows).
#define H2B_ROWS 256
#define H2B_COLS 32
#ifndef H2B_PAD
#define H2B_PAD 0                                /* elements of padding behind each block (% 8 == 0): see profiles/r06_notes.md */
#endif
#define H2B_BLOCK (H2B_ROWS * H2B_COLS + H2B_PAD) /* elements from one block to the next */
__host__ __device__ __forceinline__ size_t h2b_index(size_t row, unsigned col, unsigned col_blocks) {
    return ((row >> 8) * col_blocks + (col >> 5)) * (size_t)H2B_BLOCK + (row & 255u) * H2B_COLS + (col & 31u);
}

// 4 consecutive elements of a row -> the two planes (planes `ps` elements apart): 8-byte stores
__device__ __forceinline__ void st4_planes_h2(_Float16* p, long long ps, const float4& v, float s) {
    typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    u16x4_t h; f16x4_t l;
    _Float16 a, b;
    split2h(v.x * s, a, b); h[0] = h2_keep_sign(a, v.x); l[0] = b;
    split2h(v.y * s, a, b); h[1] = h2_keep_sign(a, v.y); l[1] = b;
    split2h(v.z * s, a, b); h[2] = h2_keep_sign(a, v.z); l[2] = b;
    split2h(v.w * s, a, b); h[3] = h2_keep_sign(a, v.w); l[3] = b;
    *reinterpret_cast<u16x4_t*>(p) = h;
    *reinterpret_cast<f16x4_t*>(p + ps) = l;
}

// ---- wave64 / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// deterministic block sum (fixed tree); `red` = >= blockDim/64 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
