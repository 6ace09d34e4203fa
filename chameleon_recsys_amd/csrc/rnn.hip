// Recurrent session encoder time-step kernels (UGRNN = the reference's cell; GRU selectable: cell_kind 1).
//
// Replaces nar_module/nar/nar_model.py:1308-1361: MultiRNNCell([DropoutWrapper(UGRNNCell(H))]) unrolled by
// tf.nn.dynamic_rnn(sequence_length).  UGRNN step (tf.contrib.rnn.UGRNNCell, TF r1.12):
//     [g_act, c_act] = [x_t, h] W + b ;  c = tanh(c_act) ; g = sigmoid(g_act + 1) ; h' = g*h + (1-g)*c
// dynamic_rnn: for t >= len[b] the output row is zero and the state is carried unchanged.
//
// MI355X design: the input half x_t*W_x (+b) for ALL t is one big MFMA GEMM (gemm.hip).  The recurrence is
// independent across sessions, so one workgroup owns 32 session rows for ALL time steps - no inter-workgroup
// synchronisation, h lives in LDS, h*W_h runs on v_mfma_f32_32x32x2_f32 (W_h streamed from L2: 512 KB at
// H=256), gates / tanh / sigmoid / length masking fused into the MFMA epilogue.  Hidden size is padded to a
// multiple of 128 (H=255 -> 256); pad lanes stay exactly 0 (zero weights, tanh(0)=0).  Each workgroup runs Hp/32 waves
// (one 32-wide hidden tile per wave, two waves per SIMD at Hp=256): when the recurrence shares its CUs with the big CAR
// GEMM on the other stream lane, its share of the MFMA issue slots scales with its wave count (4 waves: 4.8 ms, see
// profiles/r01_notes.md).
#include "common.h"

// branch-free gate non-linearities on v_exp_f32 / v_rcp_f32 (~1e-6 relative; libdevice expf / tanhf /
// IEEE division cost ~100 VALU instructions per element with divergent range branches - the recurrence is latency-bound)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// acc[j] += A[32 x K] * W[K x (32-column tile j)], j < NA, on v_mfma_f32_32x32x2_f32.
//   a : this lane's A stream in LDS, already offset to (row = lane & 31, k parity = lane >> 5): element of k-pair p is a[2 p]
//   w : this lane's W stream in global memory (L2-resident), already offset to (k parity, first column + lane & 31):
//       element of k-pair p, tile j is w[2 p ldw + off[j]]
// W comes from L2 with ~1 us latency and the compiler does not software-pipeline a rolled loop (it waits for every load of
// an unrolled body before the first MFMA: the latency was exposed 16x per time step, 42 us/step measured against a 14 us
// MFMA floor).  Here the loads of k-block i+1 are issued before the MFMAs of block i (two register blocks, ping-pong).
template <int K, int NA, int KB>
__device__ __forceinline__ void mm_stream(floatx16 (&acc)[NA], const float* __restrict__ a, const float* __restrict__ w,
                                          const size_t ldw, const int (&off)[NA]) {
    static_assert(K % (4 * KB) == 0, "K must be an even number of k-blocks");
    constexpr int NB = K / (2 * KB);
    float w0[KB][NA], w1[KB][NA];           // ping-pong register blocks: one is consumed while the other is in flight
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int j = 0; j < NA; ++j) w0[i][j] = w[(size_t)(2 * i) * ldw + off[j]];
#pragma unroll 1
    for (int blk = 0; blk < NB; blk += 2) {
        const float* wn = w + (size_t)(2 * (blk + 1) * KB) * ldw;
#pragma unroll
        for (int i = 0; i < KB; ++i)
#pragma unroll
            for (int j = 0; j < NA; ++j) w1[i][j] = wn[(size_t)(2 * i) * ldw + off[j]];
        __builtin_amdgcn_sched_barrier(0);      // keep the block's loads ahead of the MFMAs (the scheduler sinks them to their use)
        const float* ab = a + 2 * blk * KB;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const float av = ab[2 * i];
#pragma unroll
            for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w0[i][j], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // block blk + 2 (the last iteration re-reads the final block: no branch around a memory op, result unused)
        const float* wnn = w + (size_t)(2 * (blk + 2 < NB ? blk + 2 : NB - 1) * KB) * ldw;
#pragma unroll
        for (int i = 0; i < KB; ++i)
#pragma unroll
            for (int j = 0; j < NA; ++j) w0[i][j] = wnn[(size_t)(2 * i) * ldw + off[j]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const float av = ab[2 * (KB + i)];
#pragma unroll
            for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w1[i][j], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// row of accumulator element e for this lane (32x32 MFMA C/D layout)
#define ACC_ROW(e, kl) (((e) & 3) + 8 * ((e) >> 2) + 4 * (kl))

// NT = 32-wide hidden tiles per wave (Hp = 128*NT)
template <int NT, int NTW>
__global__ __launch_bounds__(256 * NT / NTW) void k_ugrnn_fwd(const float* __restrict__ xproj, const float* __restrict__ Wh,
                                                   const int* __restrict__ seq_len, int B, int T,
                                                   float* __restrict__ out, float* __restrict__ hprev,
                                                   float* __restrict__ G, float* __restrict__ Cc) {
    constexpr int Hp = 128 * NT, LDH = Hp + 1, H2 = 2 * Hp;
    extern __shared__ __attribute__((aligned(16))) float hL[];       // [32][LDH]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 32;
    const int fl = lane & 31, kl = lane >> 5;
    for (int i = threadIdx.x; i < 32 * LDH; i += blockDim.x) hL[i] = 0.f;
    __syncthreads();
    const int hid0 = wave * (32 * NTW);      // 4 * NT / NTW waves, NTW 32-wide hidden tiles each
    constexpr int KB = NT <= 2 ? 8 : 4;      // W prefetch depth (registers: 2 KB NA)
    constexpr bool PF = NT <= 2;             // 12 / 16-wave workgroups have 168 / 128 VGPRs per lane: no room for the x prefetch
    int sl[16];                              // session lengths of the 16 rows this lane's accumulators cover
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int b = b0 + ACC_ROW(e, kl); sl[e] = b < B ? seq_len[b] : 0; }
    int off[2 * NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { off[j] = j * 32; off[NTW + j] = Hp + j * 32; }
    for (int t = 0; t < T; ++t) {
        // x_t W_x + b of this step: issued before the recurrent product so that their latency hides behind it
        float xg[NTW][16], xc[NTW][16];
        if (PF) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int b = b0 + ACC_ROW(e, kl);
                    const size_t o = ((size_t)(b < B ? b : 0) * T + t) * H2 + hid0 + j * 32 + fl;
                    xg[j][e] = xproj[o]; xc[j][e] = xproj[o + Hp];
                }
        }
        floatx16 acc[2 * NTW];               // gate tiles, then candidate tiles
#pragma unroll
        for (int j = 0; j < 2 * NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        mm_stream<Hp, 2 * NTW, KB>(acc, hL + fl * LDH + kl, Wh + (size_t)kl * H2 + hid0 + fl, H2, off);
        __syncthreads();                                   // every wave finished reading h_{t-1}
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int hid = hid0 + j * 32 + fl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ACC_ROW(e, kl);
                const int b = b0 + row;
                if (b >= B) continue;
                const size_t bt = (size_t)b * T + t;
                const float zg = acc[j][e] + (PF ? xg[j][e] : xproj[bt * H2 + hid]);
                const float zc = acc[NTW + j][e] + (PF ? xc[j][e] : xproj[bt * H2 + Hp + hid]);
                const float g = sigmoidf_(zg + 1.0f);      // forget_bias = 1.0
                const float c = cham_tanhf(zc);
                const float ho = hL[row * LDH + hid];
                const float hn = g * ho + (1.f - g) * c;
                const bool valid = t < sl[e];
                out[bt * Hp + hid] = valid ? hn : 0.f;
                hprev[bt * Hp + hid] = ho;
                G[bt * Hp + hid] = g;
                Cc[bt * Hp + hid] = c;
                if (valid) hL[row * LDH + hid] = hn;
            }
        }
        __syncthreads();
    }
}

// backward through time.  dout = dL/d(out) [B,T,Hp];  writes dxproj [B,T,2Hp] (= dL/d[g_act, c_act]).
// WhT = transpose(Wh) [2Hp, Hp].
template <int NT, int NTW>
__global__ __launch_bounds__(256 * NT / NTW) void k_ugrnn_bwd(const float* __restrict__ dout, const float* __restrict__ WhT,
                                                   const int* __restrict__ seq_len, int B, int T,
                                                   const float* __restrict__ hprev, const float* __restrict__ G,
                                                   const float* __restrict__ Cc, float* __restrict__ dxproj) {
    constexpr int Hp = 128 * NT, H2 = 2 * Hp, LDZ = H2 + 1;
    extern __shared__ __attribute__((aligned(16))) float dzL[];      // [32][LDZ]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 32;
    const int fl = lane & 31, kl = lane >> 5;
    const int hid0 = wave * (32 * NTW);      // 4 * NT / NTW waves, NTW 32-wide hidden tiles each
    constexpr int KB = NT <= 2 ? 8 : 4;
    constexpr bool PF = NT <= 2;                             // prefetch step t-1's saved activations behind step t's product
    int sl[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { const int b = b0 + ACC_ROW(e, kl); sl[e] = b < B ? seq_len[b] : 0; }
    int off[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) off[j] = j * 32;
    floatx16 carry[NTW];                                     // dL/dh_t flowing to step t (accumulator layout)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) carry[j][e] = 0.f;
    float pd[NTW][16], pg[NTW][16], pc[NTW][16], ph[NTW][16];    // dout, g, c, h_prev of the step being processed
    auto fetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int b = b0 + ACC_ROW(e, kl);
                const bool on = t >= 0 && t < sl[e];
                const size_t o = ((size_t)(on ? b : 0) * T + (on ? t : 0)) * Hp + hid0 + j * 32 + fl;
                pd[j][e] = dout[o]; pg[j][e] = G[o]; pc[j][e] = Cc[o]; ph[j][e] = hprev[o];
            }
    };
    if (PF) fetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        floatx16 direct[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int hid = hid0 + j * 32 + fl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ACC_ROW(e, kl);
                const int b = b0 + row;
                float dzg = 0.f, dzc = 0.f, dd = 0.f;
                if (t < (PF ? sl[e] : (b < B ? seq_len[b] : 0))) {
                    const size_t o = ((size_t)b * T + t) * Hp + hid;
                    const float dh = (PF ? pd[j][e] : dout[o]) + carry[j][e];
                    const float g = PF ? pg[j][e] : G[o], c = PF ? pc[j][e] : Cc[o], hp = PF ? ph[j][e] : hprev[o];
                    dzg = dh * (hp - c) * g * (1.f - g);
                    dzc = dh * (1.f - g) * (1.f - c * c);
                    dd = dh * g;
                }
                direct[j][e] = dd;
                dzL[row * LDZ + hid] = dzg;
                dzL[row * LDZ + Hp + hid] = dzc;
                if (b < B) {
                    const size_t o2 = ((size_t)b * T + t) * H2;
                    dxproj[o2 + hid] = dzg;
                    dxproj[o2 + Hp + hid] = dzc;
                }
            }
        }
        __syncthreads();
        if (PF) fetch(t - 1);
        floatx16 acc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        mm_stream<H2, NTW, KB>(acc, dzL + fl * LDZ + kl, WhT + (size_t)kl * Hp + hid0 + fl, Hp, off);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int b = b0 + ACC_ROW(e, kl);
                carry[j][e] = (t < (PF ? sl[e] : (b < B ? seq_len[b] : 0))) ? (direct[j][e] + acc[j][e]) : carry[j][e];
            }
        __syncthreads();                                   // dzL is rewritten by the next step
    }
}


// ---------------------------------------------------------------------------------------------------------------
// GRU (tf.nn.rnn_cell.GRUCell, TF r1.12; BASELINE config 4 / north-star "stacked GRU"):
//     [r, u] = sigmoid([x, h] W_g + b_g) ;  c = tanh([x, r*h] W_c + b_c) ;  h' = u*h + (1-u)*c
// xproj [B,T,3Hp] = x W_x + b with column blocks r | u | c;  Wh = W_gh [Hp,2Hp] followed by W_ch [Hp,Hp].
// Same ownership as the UGRNN kernel (one workgroup = 32 sessions, all time steps, h in LDS); the candidate needs
// r*h of ALL hidden units, so a step is two MFMA phases with an LDS exchange of r*h in between.
template <int NT, int NTW>
__global__ __launch_bounds__(256 * NT / NTW) void k_gru_fwd(const float* __restrict__ xproj, const float* __restrict__ Wh,
                                                 const int* __restrict__ seq_len, int B, int T,
                                                 float* __restrict__ out, float* __restrict__ hprev, float* __restrict__ U,
                                                 float* __restrict__ Cc, float* __restrict__ R, float* __restrict__ RH) {
    constexpr int Hp = 128 * NT, LDH = Hp + 1, H2 = 2 * Hp, H3 = 3 * Hp;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hL = sm;                    // [32][LDH]  h_{t-1}
    float* rhL = sm + 32 * LDH;        // [32][LDH]  r * h_{t-1}
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 32;
    const int fl = lane & 31, kl = lane >> 5;
    const float* Wgh = Wh;
    const float* Wch = Wh + (size_t)Hp * H2;
    for (int i = threadIdx.x; i < 64 * LDH; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int hid0 = wave * (32 * NTW);      // 4 * NT / NTW waves, NTW 32-wide hidden tiles each
    constexpr int KB = NT <= 2 ? 8 : 4;
    constexpr bool PF = NT <= 2;
    int off1[NTW], off2[2 * NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { off1[j] = j * 32; off2[j] = j * 32; off2[NTW + j] = Hp + j * 32; }
    for (int t = 0; t < T; ++t) {
        // x_t W_x + b of this step: issued before the recurrent products so that their latency hides behind them
        float xr[NTW][16], xu[NTW][16], xc[NTW][16];
        if (PF) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int b = b0 + ACC_ROW(e, kl);
                    const size_t o = ((size_t)(b < B ? b : 0) * T + t) * H3 + hid0 + j * 32 + fl;
                    xr[j][e] = xproj[o]; xu[j][e] = xproj[o + Hp]; xc[j][e] = xproj[o + H2];
                }
        }
        floatx16 aru[2 * NTW];               // reset-gate tiles, then update-gate tiles
#pragma unroll
        for (int j = 0; j < 2 * NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) aru[j][e] = 0.f;
        mm_stream<Hp, 2 * NTW, KB>(aru, hL + fl * LDH + kl, Wgh + (size_t)kl * H2 + hid0 + fl, H2, off2);
        floatx16 (&ar)[NTW] = *reinterpret_cast<floatx16 (*)[NTW]>(&aru[0]);
        floatx16 (&au)[NTW] = *reinterpret_cast<floatx16 (*)[NTW]>(&aru[NTW]);
        // gates; r*h -> LDS (each wave writes its own hidden columns of rhL; hL is only read here)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int hid = hid0 + j * 32 + fl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int b = b0 + row;
                float r = 0.f, u = 0.f;
                if (b < B) {
                    const size_t bt = (size_t)b * T + t;
                    r = sigmoidf_(ar[j][e] + (PF ? xr[j][e] : xproj[bt * H3 + hid]));
                    u = sigmoidf_(au[j][e] + (PF ? xu[j][e] : xproj[bt * H3 + Hp + hid]));
                }
                ar[j][e] = r; au[j][e] = u;
                rhL[row * LDH + hid] = r * hL[row * LDH + hid];
            }
        }
        __syncthreads();
        floatx16 ac[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) ac[j][e] = 0.f;
        mm_stream<Hp, NTW, KB>(ac, rhL + fl * LDH + kl, Wch + (size_t)kl * Hp + hid0 + fl, Hp, off1);
        __syncthreads();                                   // all reads of hL / rhL of this step are done
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int hid = hid0 + j * 32 + fl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int b = b0 + row;
                if (b >= B) continue;
                const size_t bt = (size_t)b * T + t;
                const float c = cham_tanhf(ac[j][e] + (PF ? xc[j][e] : xproj[bt * H3 + H2 + hid]));
                const float ho = hL[row * LDH + hid];
                const float r = ar[j][e], u = au[j][e];
                const float hn = u * ho + (1.f - u) * c;
                const bool valid = t < seq_len[b];
                const size_t o = bt * Hp + hid;
                out[o] = valid ? hn : 0.f;
                hprev[o] = ho; U[o] = u; Cc[o] = c; R[o] = r; RH[o] = r * ho;
                if (valid) hL[row * LDH + hid] = hn;
            }
        }
        __syncthreads();
    }
}

// WhT = [ transpose(W_gh) [2Hp,Hp] ; transpose(W_ch) [Hp,Hp] ].  dxproj [B,T,3Hp] = dL/d[r_act, u_act, c_act].
template <int NT, int NTW>
__global__ __launch_bounds__(256 * NT / NTW) void k_gru_bwd(const float* __restrict__ dout, const float* __restrict__ WhT,
                                                 const int* __restrict__ seq_len, int B, int T,
                                                 const float* __restrict__ hprev, const float* __restrict__ U,
                                                 const float* __restrict__ Cc, const float* __restrict__ R,
                                                 float* __restrict__ dxproj) {
    constexpr int Hp = 128 * NT, H2 = 2 * Hp, H3 = 3 * Hp, LDZ = H2 + 1, LDC = Hp + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dzL = sm;                   // [32][LDZ]  dz_r | dz_u
    float* dcL = sm + 32 * LDZ;        // [32][LDC]  dz_c
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * 32;
    const int fl = lane & 31, kl = lane >> 5;
    const int hid0 = wave * (32 * NTW);      // 4 * NT / NTW waves, NTW 32-wide hidden tiles each
    const float* WghT = WhT;
    const float* WchT = WhT + (size_t)H2 * Hp;
    constexpr int KB = NT <= 2 ? 8 : 4;
    int off1[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) off1[j] = j * 32;
    floatx16 carry[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) carry[j][e] = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        floatx16 direct[NTW], duz[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int hid = hid0 + j * 32 + fl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int b = b0 + row;
                float dzc = 0.f, dzu = 0.f, dd = 0.f;
                if (b < B && t < seq_len[b]) {
                    const size_t o = ((size_t)b * T + t) * Hp + hid;
                    const float dh = dout[o] + carry[j][e];
                    const float u = U[o], c = Cc[o], hp = hprev[o];
                    dzu = dh * (hp - c) * u * (1.f - u);
                    dzc = dh * (1.f - u) * (1.f - c * c);
                    dd = dh * u;
                }
                direct[j][e] = dd; duz[j][e] = dzu;
                dcL[row * LDC + hid] = dzc;
                if (b < B) dxproj[((size_t)b * T + t) * H3 + H2 + hid] = dzc;
            }
        }
        __syncthreads();
        // d(r*h) = dz_c * W_ch^T
        floatx16 acc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        mm_stream<Hp, NTW, KB>(acc, dcL + fl * LDC + kl, WchT + (size_t)kl * Hp + hid0 + fl, Hp, off1);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int hid = hid0 + j * 32 + fl;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int b = b0 + row;
                float dzr = 0.f;
                if (b < B && t < seq_len[b]) {
                    const size_t o = ((size_t)b * T + t) * Hp + hid;
                    const float r = R[o], hp = hprev[o], drh = acc[j][e];
                    dzr = drh * hp * r * (1.f - r);
                    direct[j][e] += drh * r;
                }
                dzL[row * LDZ + hid] = dzr;
                dzL[row * LDZ + Hp + hid] = duz[j][e];
                if (b < B) {
                    const size_t o2 = ((size_t)b * T + t) * H3;
                    dxproj[o2 + hid] = dzr;
                    dxproj[o2 + Hp + hid] = duz[j][e];
                }
            }
        }
        __syncthreads();
        // dh_{t-1} += [dz_r, dz_u] * W_gh^T
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        mm_stream<H2, NTW, KB>(acc, dzL + fl * LDZ + kl, WghT + (size_t)kl * Hp + hid0 + fl, Hp, off1);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kl;
                const int b = b0 + row;
                const bool valid = (b < B) && (t < seq_len[b]);
                carry[j][e] = valid ? (direct[j][e] + acc[j][e]) : carry[j][e];
            }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ outp) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8)
        if (by + r < rows && bx + tx < cols) tile[r][tx] = in[(size_t)(by + r) * cols + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < cols && by + tx < rows) outp[(size_t)(bx + r) * rows + by + tx] = tile[tx][r];
}

extern "C" int cham_transpose_f32(const float* in, int rows, int cols, float* out, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, rows, cols, out);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// ---- valid-position compaction (nar_model.py:231 sequence_mask): the step runs its row-wise stages on the P non-padded
// (session, time) positions only; the recurrent stack keeps the [B, T] layout.  These two move rows between the layouts
// (rows are `words` 32-bit words wide: fp32 activations, int32 slots, int64 ids as 2 words).
__global__ __launch_bounds__(256) void k_rows_gather(const uint32_t* __restrict__ src, const int32_t* __restrict__ pos, long n_rows,
                                                     int words, uint32_t* __restrict__ dst) {
    const long total = n_rows * words;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / words; const int c = (int)(i - r * words);
        dst[i] = src[(size_t)pos[r] * words + c];
    }
}
__global__ __launch_bounds__(256) void k_rows_scatter(const uint32_t* __restrict__ src, const int32_t* __restrict__ pos, long n_rows,
                                                      int words, uint32_t* __restrict__ dst) {
    const long total = n_rows * words;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / words; const int c = (int)(i - r * words);
        dst[(size_t)pos[r] * words + c] = src[i];
    }
}
static int rows_move(bool gather, const void* src, const int32_t* pos, long n_rows, int words, void* dst, void* stream) {
    if (!src || !pos || !dst || n_rows < 0 || words <= 0) return -CHAM_ERR_ARG;
    if (n_rows == 0) return CHAM_OK;
    long blocks = (n_rows * words + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gather ? k_rows_gather : k_rows_scatter, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)src, pos, n_rows, words, (uint32_t*)dst);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
// dst[i, :] = src[pos[i], :]   (i < n_rows)
extern "C" int cham_rows_gather(const void* src, const int32_t* pos, long n_rows, int words, void* dst, void* stream) {
    return rows_move(true, src, pos, n_rows, words, dst, stream);
}
// dst[pos[i], :] = src[i, :]   (i < n_rows; pos holds distinct rows; rows of dst not named by pos are left untouched)
extern "C" int cham_rows_scatter(const void* src, const int32_t* pos, long n_rows, int words, void* dst, void* stream) {
    return rows_move(false, src, pos, n_rows, words, dst, stream);
}

// When the recurrent kernels run on a side stream next to the big CAR GEMM they must not share a CU with GEMM
// workgroups (the MFMA pipes would be time-sliced and the latency-bound recurrence would stretch 5x): requesting
// most of the 160 KB LDS makes every CU that hosts a recurrent workgroup exclusive to it.

template <int NT>
static int launch_ugrnn_fwd(const float* xproj, const float* Wh, const int* seq_len, int B, int T, float* out, float* hprev,
                            float* G, float* Cc, hipStream_t st) {
    size_t smem = (size_t)32 * (128 * NT + 1) * sizeof(float);
    auto kern = k_ugrnn_fwd<NT, 1>;
    CHAM_SET_DYNAMIC_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((B + 31) / 32), dim3(256 * NT), smem, st, xproj, Wh, seq_len, B, T, out, hprev, G, Cc);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
template <int NT>
static int launch_ugrnn_bwd(const float* dout, const float* WhT, const int* seq_len, int B, int T, const float* hprev,
                            const float* G, const float* Cc, float* dxproj, hipStream_t st) {
    size_t smem = (size_t)32 * (256 * NT + 1) * sizeof(float);
    auto kern = k_ugrnn_bwd<NT, 1>;
    CHAM_SET_DYNAMIC_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((B + 31) / 32), dim3(256 * NT), smem, st, dout, WhT, seq_len, B, T, hprev, G, Cc, dxproj);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

template <int NT>
static int launch_gru_fwd(const float* xproj, const float* Wh, const int* seq_len, int B, int T, float* out, float* hprev,
                          float* U, float* Cc, float* R, float* RH, hipStream_t st) {
    size_t smem = (size_t)64 * (128 * NT + 1) * sizeof(float);
    auto kern = k_gru_fwd<NT, 1>;
    CHAM_SET_DYNAMIC_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((B + 31) / 32), dim3(256 * NT), smem, st, xproj, Wh, seq_len, B, T, out, hprev, U, Cc, R, RH);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
template <int NT>
static int launch_gru_bwd(const float* dout, const float* WhT, const int* seq_len, int B, int T, const float* hprev,
                          const float* U, const float* Cc, const float* R, float* dxproj, hipStream_t st) {
    size_t smem = (size_t)32 * (384 * NT + 2) * sizeof(float);
    auto kern = k_gru_bwd<NT, 1>;
    CHAM_SET_DYNAMIC_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((B + 31) / 32), dim3(256 * NT), smem, st, dout, WhT, seq_len, B, T, hprev, U, Cc, R, dxproj);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}

// cell_kind: 0 = UGRNN
extern "C" int cham_rnn_fwd(int cell_kind, const float* xproj, const float* Wh, const int32_t* seq_len, int B, int T, int Hp,
                            float* out, float* hprev, float* G, float* Cc, float* R, float* RH, void* stream) {
    if (!xproj || !Wh || !seq_len || !out || !hprev || !G || !Cc || B <= 0 || T <= 0) return -CHAM_ERR_ARG;
    if (cell_kind != 0 && cell_kind != 1) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (cell_kind == 1) {
        if (!R || !RH) return -CHAM_ERR_ARG;
        switch (Hp) {
            case 128: return launch_gru_fwd<1>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, R, RH, st);
            case 256: return launch_gru_fwd<2>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, R, RH, st);
            case 384: return launch_gru_fwd<3>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, R, RH, st);
            default: return -CHAM_ERR_ARG;
        }
    }
    switch (Hp) {
        case 128: return launch_ugrnn_fwd<1>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, st);
        case 256: return launch_ugrnn_fwd<2>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, st);
        case 384: return launch_ugrnn_fwd<3>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, st);
        case 512: return launch_ugrnn_fwd<4>(xproj, Wh, seq_len, B, T, out, hprev, G, Cc, st);
        default: return -CHAM_ERR_ARG;
    }
}

extern "C" int cham_rnn_bwd(int cell_kind, const float* dout, const float* WhT, const int32_t* seq_len, int B, int T, int Hp,
                            const float* hprev, const float* G, const float* Cc, const float* R, float* dxproj, void* stream) {
    if (!dout || !WhT || !seq_len || !hprev || !G || !Cc || !dxproj || B <= 0 || T <= 0) return -CHAM_ERR_ARG;
    if (cell_kind != 0 && cell_kind != 1) return -CHAM_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (cell_kind == 1) {
        if (!R) return -CHAM_ERR_ARG;
        switch (Hp) {
            case 128: return launch_gru_bwd<1>(dout, WhT, seq_len, B, T, hprev, G, Cc, R, dxproj, st);
            case 256: return launch_gru_bwd<2>(dout, WhT, seq_len, B, T, hprev, G, Cc, R, dxproj, st);
            case 384: return launch_gru_bwd<3>(dout, WhT, seq_len, B, T, hprev, G, Cc, R, dxproj, st);
            default: return -CHAM_ERR_ARG;
        }
    }
    switch (Hp) {
        case 128: return launch_ugrnn_bwd<1>(dout, WhT, seq_len, B, T, hprev, G, Cc, dxproj, st);
        case 256: return launch_ugrnn_bwd<2>(dout, WhT, seq_len, B, T, hprev, G, Cc, dxproj, st);
        case 384: return launch_ugrnn_bwd<3>(dout, WhT, seq_len, B, T, hprev, G, Cc, dxproj, st);
        case 512: return launch_ugrnn_bwd<4>(dout, WhT, seq_len, B, T, hprev, G, Cc, dxproj, st);
        default: return -CHAM_ERR_ARG;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Step-wise fallback for hidden sizes beyond what the fused time-loop kernels hold in LDS (UGRNN Hp > 512, GRU Hp > 384;
// the reference's hypertuning searches rnn_units up to 1024, nar_mlengine_hypertuning.yaml:28-33): the host runs the
// recurrent products h W_h as GEMM launches per time step (csrc/gemm.hip) and these kernels do the gate arithmetic,
// length masking and state carry.  zh = h_{t-1} W_h for the step (UGRNN: [B, 2Hp]; GRU gates: [B, 2Hp], candidate: [B, Hp]).
__global__ __launch_bounds__(256) void k_ugrnn_point_fwd(const float* __restrict__ xproj, const float* __restrict__ zh,
                                                         const int* __restrict__ seq_len, int B, int T, int t, int Hp,
                                                         float* __restrict__ h /*[B,Hp] state in/out*/, float* __restrict__ out,
                                                         float* __restrict__ hprev, float* __restrict__ G, float* __restrict__ Cc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Hp) return;
    const int b = i / Hp, hid = i % Hp;
    const size_t bt = (size_t)b * T + t, o = bt * Hp + hid;
    const float g = sigmoidf_(zh[(size_t)b * 2 * Hp + hid] + xproj[bt * 2 * Hp + hid] + 1.0f);
    const float c = cham_tanhf(zh[(size_t)b * 2 * Hp + Hp + hid] + xproj[bt * 2 * Hp + Hp + hid]);
    const float ho = h[i], hn = g * ho + (1.f - g) * c;
    const bool valid = t < seq_len[b];
    out[o] = valid ? hn : 0.f; hprev[o] = ho; G[o] = g; Cc[o] = c;
    if (valid) h[i] = hn;
}
// dh = dout[:,t] + carry;  dz -> dxproj[:,t] and dzs [B,2Hp] (for carry_next = direct + dzs W_h^T, a GEMM);  direct [B,Hp]
__global__ __launch_bounds__(256) void k_ugrnn_point_bwd(const float* __restrict__ dout, const float* __restrict__ carry,
                                                         const int* __restrict__ seq_len, int B, int T, int t, int Hp,
                                                         const float* __restrict__ hprev, const float* __restrict__ G,
                                                         const float* __restrict__ Cc, float* __restrict__ dxproj,
                                                         float* __restrict__ dzs, float* __restrict__ direct) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Hp) return;
    const int b = i / Hp, hid = i % Hp;
    const size_t bt = (size_t)b * T + t, o = bt * Hp + hid;
    float dzg = 0.f, dzc = 0.f, dd = carry[i];                 // beyond the session's length the carry passes through unchanged
    if (t < seq_len[b]) {
        const float dh = dout[o] + carry[i];
        const float g = G[o], c = Cc[o], hp = hprev[o];
        dzg = dh * (hp - c) * g * (1.f - g);
        dzc = dh * (1.f - g) * (1.f - c * c);
        dd = dh * g;
    }
    dxproj[bt * 2 * Hp + hid] = dzg; dxproj[bt * 2 * Hp + Hp + hid] = dzc;
    dzs[(size_t)b * 2 * Hp + hid] = dzg; dzs[(size_t)b * 2 * Hp + Hp + hid] = dzc;
    direct[i] = dd;
}
extern "C" int cham_ugrnn_point_fwd(const float* xproj, const float* zh, const int32_t* seq_len, int B, int T, int t, int Hp,
                                    float* h, float* out, float* hprev, float* G, float* Cc, void* stream) {
    if (!xproj || !zh || !seq_len || !h || !out || !hprev || !G || !Cc || B <= 0 || T <= 0 || t < 0 || t >= T) return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_ugrnn_point_fwd, dim3((B * Hp + 255) / 256), dim3(256), 0, (hipStream_t)stream, xproj, zh, seq_len, B, T, t,
                       Hp, h, out, hprev, G, Cc);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
extern "C" int cham_ugrnn_point_bwd(const float* dout, const float* carry, const int32_t* seq_len, int B, int T, int t, int Hp,
                                    const float* hprev, const float* G, const float* Cc, float* dxproj, float* dzs, float* direct,
                                    void* stream) {
    if (!dout || !carry || !seq_len || !hprev || !G || !Cc || !dxproj || !dzs || !direct || B <= 0 || T <= 0 || t < 0 || t >= T)
        return -CHAM_ERR_ARG;
    hipLaunchKernelGGL(k_ugrnn_point_bwd, dim3((B * Hp + 255) / 256), dim3(256), 0, (hipStream_t)stream, dout, carry, seq_len, B, T, t,
                       Hp, hprev, G, Cc, dxproj, dzs, direct);
    CHAM_CHECK_LAUNCH();
    return CHAM_OK;
}
