/*
 * chameleon_nar.h - C ABI of libchameleon_nar.so: the MI355X (gfx950) NAR training-step kernels.
 *
 * The reference (gabrielspmoreira/chameleon_recsys) has NO FFI: its hot path is a TensorFlow-1.12 graph built
 * by nar_module/nar/nar_model.py.  Each entry point below replaces one cluster of TF ops of that graph; the
 * cluster is cited as file:line (relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions: every pointer is a DEVICE pointer unless marked host; row-major; `void* stream` is a
 * hipStream_t; kernels are stream-ordered and re-entrant per stream; the caller owns every buffer; return value
 * 0 = ok, -22 (-EINVAL) = bad argument, -5 (-EIO) = launch failure.  No exceptions cross the boundary.
 * Feature / hidden dimensions are padded by the host to multiples of 4 floats (16-byte rows).
 */
#ifndef CHAMELEON_NAR_H
#define CHAMELEON_NAR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CHAM_ACT_NONE 0
#define CHAM_ACT_LEAKY 1 /* tf.nn.leaky_relu, alpha = 0.2 */
#define CHAM_ACT_TANH 2

/* --- K0 negative sampling: nar_model.py:1220-1304 (+ :265-276) -------------------------------------------
 * aci = all_clicked_items of the GLOBAL batch [Bg, T1] (T1 = T+1, nar_model.py:241); output rows
 * [row_begin, row_begin+row_count).  neg_ids [row_count, T, N]; neg_slot = pool position of each negative
 * (20*N = zero-padding item, -1 = padded click); pool [20*N]; canon [20*N]; meta[4] = {_, n_buffer_sample, _, P}.
 * RNG: Philox4x32-10 keyed (seed, step); see oracle/philox.py for the contract. */
size_t cham_neg_sample_workspace_bytes(int n_aci, int buf_size, int n_from_buffer);
int cham_neg_sample(const int64_t* aci, int Bg, int T1, const int64_t* buffer, int buf_size, uint32_t seed, uint32_t step,
                    int row_begin, int row_count, int N, int n_from_buffer, int64_t* neg_ids, int32_t* neg_slot,
                    int64_t* pool, int32_t* canon, int32_t* meta, void* workspace, size_t workspace_bytes, void* stream);

/* --- K1 feature gather / assemble ------------------------------------------------------------------------
 * Column descriptors: 5 x int64 per output column {kind, feat, sub, dim, param_offset};
 * kind: 0 zero-pad, 1 one-hot, 2 embedding, 3 numeric, 4 ACE column, 5 item embedding, 6 recency, 7 novelty. */
/* user context features, nar_model.py:730-773 + :887-907.  cat [n_cat][R] int64, num [n_num][R] float */
int cham_ctx_assemble(const int64_t* cat, const float* num, int R, const int64_t* desc, int F, const float* params,
                      const float* gamma, const float* beta, float* xraw, float* xs, void* stream);
/* The two smoothing-log bases of the dynamic item features - NARModuleModel(elapsed_days_smooth_log_base = 1.3,
 * popularity_smooth_log_base = 2.0), nar_model.py:122-123, used by log_base / log_1p (:28-34) in the recency feature (:1071-1075), the
 * novelty feature (:1148) and the novelty regulariser (:544, 673-683).  Launch scalars of cham_item_dynamic_raw, cham_norm_stats_from_*
 * and cham_score_softmax_fwd / _bwd; process-wide, set before the launches of a step (NARModuleModel.forward does).  Bases must be
 * > 0 and != 1. */
int cham_set_log_bases(float elapsed_days_smooth_log_base, float popularity_smooth_log_base);

/* head of a step, the integer row sets in one launch (nar_model.py:217-248, 343, 356): ids_all [2 BT + pmax + 1] = [clicked ids | positive ids |
 * candidate pool | pad item 0], ref_ts = [click time stamps | max_ts ...], seq_len [B] and mask [BT] copied into the step's buffers */
int cham_step_ints(const int64_t* ic_rows, const int64_t* ln_rows, const int64_t* pool, const int64_t* ets_rows, int64_t max_ts, int BT, int pmax,
                   const int32_t* seq_len_in, int B, const uint8_t* mask_in, int64_t* ids_all, int64_t* ref_ts, int32_t* seq_len, uint8_t* mask,
                   void* stream);

/* raw recency log_1.3(1+relu((f32(ts)-f32(created))/86.4e6)) and novelty -log2(pop_norm): nar_model.py:1055-1060,
 * 1074, 1147-1148 */
int cham_item_dynamic_raw(const int64_t* ids, const int64_t* ref_ts, int R, const int64_t* created, const float* pop_norm,
                          float* rec_raw, float* nov_raw, void* stream);
/* normalisation population = last `recent_clicks_for_normalization` buffer clicks: nar_model.py:1062-1089, 1150-1186;
 * stats [3][8] = per call group {mean, sd, zmin, zmax} x {recency, novelty} */
int cham_norm_stats_from_recent(const int64_t* last_ids, int n_last, int64_t max_ts, const int64_t* created,
                                const float* pop_norm, float* scratch, float* stats, void* stream);
/* the same statistics straight from the device-resident buffer (cham_state_update): population = the valid entries among
 * the first n_prefix (= recent_clicks_for_normalization) buffer slots; scratch holds 3 * n_prefix floats */
int cham_norm_stats_from_buffer(const int64_t* buffer_ids, int n_prefix, int64_t max_ts, const int64_t* created,
                                const float* pop_norm, float* scratch, float* stats, void* stream);
/* empty-buffer fallback (first batch): population = the call's own non-pad ids, nar_model.py:1078-1084, 1168-1181 */
int cham_norm_stats_from_rows(const float* rec_raw, const float* nov_raw, const float* weights, int n, float* stats_group,
                              void* stream);
int cham_row_weights(const int64_t* ids, int n_ids, const int32_t* neg_slot, size_t n_neg, int pmax, const int64_t* pool,
                     float* w_ids, float* w_slots, void* stream);
/* item rows, nar_model.py:921-994 (+ normalize_values :996-1039, scale/center :887-907).
 * rows [0,g1_begin) = clicked inputs, [g1_begin,g2_begin) = positives, rest = candidate-pool slots */
int cham_item_assemble(const int64_t* ids, int R, int g1_begin, int g2_begin, const int64_t* meta_cat, int n_items,
                       const float* ace, int ld_ace, const float* rec_raw, const float* nov_raw, const float* stats,
                       const int64_t* desc, int F, const float* params, const float* gamma, const float* beta, float* xraw,
                       float* xs, void* stream);
/* the same item rows assembled through LDS tiles (csrc/features.hip k_item_assemble_lds): every row is a few contiguous source
 * segments (ACE row, item-embedding row, metadata-embedding rows) copied with lane-contiguous loads into an LDS image and streamed
 * out with 16-byte stores.  segs [n_segs][6] int64 = {kind 0 ACE / 1 item embedding / 2 metadata embedding, destination column,
 * length, offset of the table in `params`, row pitch, metadata feature}; singles [n_singles] = columns no segment covers (one-hot
 * bits, numerics, recency, novelty: decoded through `desc`).  F % 4 == 0. */
int cham_item_assemble_lds(const int64_t* ids, int R, int g1_begin, int g2_begin, const int64_t* meta_cat, int n_items,
                           const float* ace, int ld_ace, const float* rec_raw, const float* nov_raw, const float* stats,
                           const int64_t* desc, int F, const int64_t* segs, int n_segs, const int32_t* singles, int n_singles,
                           const float* params, const float* gamma, const float* beta, float* xraw, float* xs, void* stream);
/* backward of the two above, scale/center part (nar_model.py:887-907): dgamma[c] = sum_r dxs[r,c] * xraw[r,c], dbeta[c] = sum_r dxs[r,c] */
int cham_feature_bwd(const float* dxs, const float* xraw, int R, int F, float* dgamma, float* dbeta, void* stream);
/* the same column sums with coalesced reads, two launches through a workspace private to the call's stream (round 6: the one-workgroup-
 * per-column form reads 4 bytes of every 128-byte line); deterministic, another summation order than cham_feature_bwd */
size_t cham_feature_bwd_workspace_bytes(int F);
int cham_feature_bwd_ws(const float* dxs, const float* xraw, int R, int F, float* dgamma, float* dbeta, float* workspace,
                        size_t workspace_bytes, void* stream);

/* --- K6 embedding-table gradients (the IndexedSlices TF builds for tf.nn.embedding_lookup, nar_model.py:741, 918), DETERMINISTIC:
 * rows of dxs that looked up the same table row are summed in a fixed order, no float atomics.
 * table_grad[key, sub] = gamma[c0 + sub] * sum_{r : key(r) == key} dxs[r, c0 + sub]   (table rows nobody looked up are not written:
 * the caller zeroes the embedding-gradient region first).
 *  - cham_emb_grad_scan: small tables (context / metadata embeddings), one workgroup per table row scanning the R source keys;
 *    key(r) = keysrc[r] (ids == NULL) or keysrc[ids[r]] (article metadata of item rows);
 *  - cham_group_rows + cham_emb_grad_grouped: the item-embedding table; cham_group_rows sorts the rows by (id, row) - a stable 8-bit
 *    LSD radix sort, ceil(key_bits / 8) passes, ids < 2^key_bits (0: 32 bits), workspace of cham_group_rows_workspace_bytes(R) - (depends on the
 *    ids only - run it in the forward pass), perm[i] = row with the i-th smallest key, and builds the segment table `seg`
 *    (cham_group_rows_segments_len(R) int32 words: number of segments, of segments longer than 32 rows and of their 64-row chunks,
 *    first sorted position of every segment, the indices of the long ones, their first chunk, scratch for the chunk sums - the buffer
 *    is written by cham_emb_grad_grouped too); cham_emb_grad_grouped then runs one wave per short segment, one workgroup per 128-row
 *    chunk of a long one and adds the chunk sums in chunk order; rows in ascending order throughout; 0 <= ids < 2^32, dim <= 512. */
int cham_emb_grad_scan(const float* dxs, int R, int F, int c0, int dim, const float* gamma, const int64_t* keysrc,
                       const int64_t* ids, int cardinality, float* table_grad, void* stream);
size_t cham_group_rows_workspace_bytes(int R);
size_t cham_group_rows_segments_len(int R);
int cham_group_rows(const int64_t* ids, int R, int key_bits, int32_t* perm, int32_t* seg, void* workspace, size_t workspace_bytes,
                    void* stream);
int cham_emb_grad_grouped(const float* dxs, int R, int F, int c0, int dim, const float* gamma, const int64_t* ids,
                          const int32_t* perm, const int32_t* seg, float* table_grad, void* stream);

/* --- dropout (dropout_keep_prob < 1): tf.layers.dropout at nar_model.py:338, 352, 368, 418 and DropoutWrapper(output_keep_prob) at
 * :1331.  y = x / keep_prob * mask (TF 1.12 tf.nn.dropout); the mask is a pure function of the element's coordinates:
 * kept iff Philox4x32-10(ctr = (column, t, GLOBAL session row, site + 256 n), key = (seed, step))[0] < floor(keep_prob * 2^32)
 * (oracle/philox.py) - independent of row shards / compaction, recomputed in the backward pass.  Row r -> position r / group,
 * sub = r % group (sub 0: site_first; sub > 0: site_rest, n = sub - 1); position -> (b, t) through pos[] or directly;
 * column c -> c < col_split ? c : c - col_shift.  In place (y == x) allowed.
 * cham_dense_rows: the un-factorised PreCAR input rows [clicked | candidates] x [ctx | item] the dropout path needs. */
int cham_dropout(const float* x, float* y, long rows, int cols, int ld, float keep_prob, uint32_t seed, uint32_t step,
                 int site_first, int site_rest, int group, const int32_t* pos, int T, int row_begin, int col_split, int col_shift,
                 void* stream);
int cham_dense_rows(const float* Xc_s, int Fc, const float* Xi_s, int Fi, int BT, int N, int pmax, const int32_t* neg_slot, float* X,
                    void* stream);

/* --- K2/K4 fp32 MFMA GEMM with fused prologue/epilogue: tf.layers.Dense at nar_model.py:374-405, 410-426, 447-473,
 * the RNN input projection (:1308-1361) and their gradients.
 * C[M,N] (+)= epi(op(A)[M,K] * op(B)[K,N]); transA: A stored [K,M]; transB: B stored [N,K];
 * epi: +bias, act; or *act'(dref) (dgrad); rowscale: A_stored[r,c] *= rowscale[(r / rs_div), c] */
int cham_gemm_f32(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc, int M, int N,
                  int K, const float* bias, int act, const float* dref, int ldr, int dact, const float* rowscale, int ldrs,
                  int rs_div, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, void* stream);

/* bf16-compute variant (BASELINE config 3): identical contract and fp32 storage; op(A), op(B) are rounded to bf16
 * (round-to-nearest-even) while staged into LDS, multiplied on v_mfma_f32_32x32x16_bf16, accumulated / finished in fp32 */
int cham_gemm_bf16(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc, int M, int N,
                   int K, const float* bias, int act, const float* dref, int ldr, int dact, const float* rowscale, int ldrs,
                   int rs_div, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, void* stream);

/* fp32 GEMM on the bf16 matrix cores (csrc/gemm_x3.hip): identical contract, fp32 storage and fp32-grade error; every operand is split
 * into three bf16 planes (a = a_h + a_m + a_l, exact) while staged into LDS and six plane products are accumulated in fp32 on
 * v_mfma_f32_32x32x16_bf16 (the dropped terms are <= 2^-23 |a b| in the worst case: tests/test_split3_cpu.py).  6/16 of the native fp32 matrix time on gfx950.  N <= 64 is
 * delegated to cham_gemm_f32.  An infinite operand yields NaN where cham_gemm_f32 yields inf. */
int cham_gemm_f32x3(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc, int M, int N,
                    int K, const float* bias, int act, const float* dref, int ldr, int dact, const float* rowscale, int ldrs,
                    int rs_div, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, void* stream);
/* test / tuning aids: tile variant (0 = 128x128 / 4 waves, 2 = 256x128 / 8 waves, -1 = automatic); launches since the last reset -
 * out8[0] / out8[1] = those tiles, out8[3] = delegated to cham_gemm_f32, out8[6] / out8[7] = epilogue variant / K-splits of the last launch */
void cham_gemm_f32x3_set_variant(int variant);
void cham_gemm_f32x3_launch_counts(long long* out8, int reset);

/* The same kernel over TWO fp16 planes per operand, split while staged (round 5): h = fp16(x * scale), l = fp16(x * scale - h), three plane
 * products on v_mfma_f32_32x32x16_f16 instead of six, the result scaled back by 1 / (scale_a scale_b) (powers of two: exact) - the arithmetic
 * of cham_gemm_h2 below for operands that exist in fp32 anyway.  sa_rec / sb_rec: the operands' H2Scale records (device memory, see
 * cham_gemm_h2; A's covers A x rowscale when a row-broadcast scale is given); |element x scale| < 65 504 is the caller's contract.  Forms:
 * NN (transA = transB = 0; bias + CHAM_ACT_NONE / LEAKY / TANH; rowscale on A) and TN (transA = 1; split-K as cham_gemm_f32; rowscale on
 * A), N > 64; -EINVAL otherwise.  Replaces the weight gradient of matching_dense_layer_1 over cand (.) pred (nar_model.py:447-451, 478-495,
 * autodiff under :718; |tanh x tanh| <= 1: a constant record) when the runtime's default arithmetic is on; the forward matmul of that layer
 * only under CHAM_S1_H2=a (it keeps its exact operands by default: DESIGN.md section 3).  Launches are counted in
 * cham_gemm_f32x3_launch_counts out8[4] (128 x 128 tile) / out8[5] (256 x 128). */
int cham_gemm_f32x2h(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc, int M, int N, int K,
                     const float* bias, int act, const float* rowscale, int ldrs, int rs_div, int accumulate, float* workspace,
                     size_t workspace_bytes, int splits_hint, const float* sa_rec, const float* sb_rec, void* stream);

/* fp32-grade GEMM over operands that already live in HBM as THREE bf16 PLANES (csrc/gemm_p3.hip, round 3): the same six plane products
 * as cham_gemm_f32x3, but the split is done once by the kernel that produces the matrix (cham_combine_fwd_p3, cham_mulpred_bwd_p3,
 * cham_split3 for weights) and the K loop is MFMAs + fragment reads + LDS-DMA only.  Replaces the CAR layer-2 matmul over the
 * B*T*(1+N) candidate rows (nar_model.py:374-405, tf.layers.Dense `CAR_representation`) and its two autodiff twins.
 *   A, B: plane 0 (bf16); planes `*_plane_stride` ELEMENTS apart (h, m, l).
 *   tn = 0 (NT): A [M, lda], B [N, ldb], k contiguous, K % 16 == 0; epilogue + bias and act (CHAM_ACT_NONE / CHAM_ACT_TANH), or
 *     x leaky'(saved activation) with dref_h = the h plane [M, ldr] of that activation and dact = CHAM_ACT_LEAKY.
 *   tn = 1 (TN): A stored [K, lda >= M], B stored [K, ldb >= N]; M % 256 == 0, N % 256 == 0, any K; split-K through `workspace`
 *     (splits_hint: 1 none, 0 automatic, n at most n; fixed-order reduction); accumulate adds to C.
 * Returns -EINVAL for a shape it does not take (the caller keeps cham_gemm_f32x3 for those).  Leading dimensions and plane strides
 * % 8 == 0, 16-byte aligned bases.  cham_gemm_p3_launch_counts: out8[0] / out8[1] = NT / TN launches since the last reset, out8[6] /
 * out8[7] = epilogue variant / K-splits of the last launch. */
int cham_gemm_p3(const void* A, long long a_plane_stride, int lda, const void* B, long long b_plane_stride, int ldb, int tn, float* C,
                 int ldc, int M, int N, int K, const float* bias, int act, const void* dref_h, int ldr, int dact, int accumulate,
                 float* workspace, size_t workspace_bytes, int splits_hint, void* stream);

/* The bf16 configuration's CAR layer-2 GEMMs (nar_model.py:374-405 and their autodiff twins under --gemm_dtype bf16) on the same
 * LDS-DMA core: ONE bf16 plane per operand, a stage = three consecutive 16-k chunks (csrc/gemm_p3.hip, gemm_b1_kernel).
 *   tn = 0 (NT): A [M, lda], B [N, ldb] bf16, k contiguous, K % 16 == 0; C bf16 [M, ldc] = bf16(tanh(A B^T + bias)) (bias fp32, act =
 *     CHAM_ACT_TANH), bf16((A B^T) x leaky'(dref)) (dref = saved bf16 activation [M, ldr], dact = CHAM_ACT_LEAKY) or bf16(A B^T).
 *   tn = 1 (TN): A stored [K, lda >= M], B stored [K, ldb >= N] bf16; M % 256 == 0, N % 256 == 0, any K; C fp32 [M, ldc] (accumulate
 *     adds to it); split-K through `workspace` as cham_gemm_p3.
 * Returns -EINVAL for a shape it does not take (the caller keeps cham_gemm_b16).  cham_gemm_p3_launch_counts: out8[2] / out8[3] = its
 * NT / TN launches. */
int cham_gemm_b16_dma(const void* A, int lda, const void* B, int ldb, int tn, void* C, int ldc, int M, int N, int K, const float* bias,
                      int act, const void* dref, int ldr, int dact, int accumulate, float* workspace, size_t workspace_bytes,
                      int splits_hint, void* stream);
void cham_gemm_p3_launch_counts(long long* out8, int reset);
/* cham_gemm_b16_dma, NT forms with K % 64 == 0 (round 6): gemm_b1w_kernel stages 64-byte source pieces (16 rows x 64 bytes per LDS-DMA request,
 * two 64-k buffers) where gemm_b1_kernel fetches a quarter of each 128-byte line per request; same products, another grouping of the K
 * loop (fp32 accumulation either way; not bit-identical).  out8[4] of cham_gemm_p3_launch_counts counts its launches;
 * cham_gemm_b16_dma_set_nt_wide(0) = always gemm_b1_kernel (A/B arm, tests); returns the previous setting. */
int cham_gemm_b16_dma_set_nt_wide(int on);
/* split3 of an fp32 matrix X [R, Cc] (row stride ld) into bf16 planes: dst[q][r][c] (planes plane_stride elements apart, row stride
 * ldd) and / or dstT[q][c][r] (the transposed matrix); either may be NULL.  a = h + m + l exactly (tests/test_split3_cpu.py). */
int cham_split3(const float* X, int R, int Cc, int ld, void* dst, long long plane_stride, int ldd, void* dstT, long long plane_strideT,
                int lddT, void* stream);

/* fp32-grade GEMM over operands that live in HBM as TWO fp16 PLANES + a power-of-two scale (csrc/gemm_h2.hip, round 4): three plane
 * products (a_h b_h + a_h b_l + a_l b_h) per fp32 product instead of cham_gemm_p3's six.  Replaces the same three matmuls - the CAR
 * layer-2 matmul over the candidate rows (nar_model.py:384-388), its dgrad and its weight gradient (autodiff of :384-388 under
 * tf.train.AdamOptimizer.compute_gradients, :718) - when the runtime's default arithmetic is on.
 *   An H2Scale RECORD is 32 bytes of device memory, zero-initialised once by the caller: {float scale, float inv_scale, float bound,
 *   pad, 3 words of kernel scratch, pad}.  x is stored as h = fp16(x * scale), l = fp16(x * scale - h); scale = 2^k with
 *   bound * scale in [2^14, 2^15).  cham_h2_scale_absmax: bound = max|x0| + max|x1| (x1 may be NULL; n0, n1 element counts, % 4 == 0).
 *   cham_h2_scale_rownorm: bound = (max over rows of ||X[r, 0:K]||_2) * (*factor if factor != NULL) - the Cauchy-Schwarz bound of a
 *   product of two matrices' rows (factor = &other_record.bound).  Both are stream-ordered, order-independent (bit-reproducible) and need no
 *   host synchronisation.  cham_split2h: planes of an fp32 matrix X [R, Cc] (dst[q][r][c] and / or the transposed dstT[q][c][r]) with the
 *   scale of `rec`, computed from max|X| first when recompute != 0 (then ld == Cc).
 *   cham_gemm_h2: A, B = plane 0 (h) of each operand, the l plane `*_plane_stride` ELEMENTS further; a_scale / b_scale = the operands'
 *   records; tn / epilogues / split-K / return codes exactly as cham_gemm_p3 (dref_h = the fp16 h plane of the saved activation: its sign
 *   is kept even where the value underflows).  cham_gemm_h2_launch_counts: out8[0] / out8[1] = NT / TN launches since the last reset,
 *   out8[2] = the NT launches among them that ran the 64-byte-source-piece kernel (round 5: K % 32 == 0; two 16-k chunks of a row per
 *   LDS-DMA request, half the L1 line fills per byte used), out8[6] / out8[7] = epilogue variant / K-splits of the last launch.
 *   cham_gemm_h2_set_nt_wide(on): 1 (default) = NT launches with K % 32 == 0 take that kernel, 0 = always the 32-byte-piece kernel of
 *   round 4 (A/B arm; results are bit-identical: same products, same summation order); returns the previous setting. */
int cham_h2_scale_absmax(const float* x0, size_t n0, const float* x1, size_t n1, void* rec, void* stream);
int cham_h2_scale_rownorm(const float* X, long R, int K, int ld, const float* factor, void* rec, void* stream);
/* ... and, from the same pass, a second record WITHOUT the factor (rec_plain, may be NULL): max row norm of X >= max |X| - the scale of X
 * itself as a two-plane operand of cham_gemm_f32x2h */
int cham_h2_scale_rownorm2(const float* X, long R, int K, int ld, const float* factor, void* rec, void* rec_plain, void* stream);
int cham_split2h(const float* X, int R, int Cc, int ld, void* dst, long long plane_stride, int ldd, void* dstT, long long plane_strideT,
                 int lddT, void* rec, int recompute, void* stream);
int cham_gemm_h2(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride, int ldb,
                 const float* b_scale, int tn, float* C, int ldc, int M, int N, int K, const float* bias, int act, const void* dref_h, int ldr,
                 int dact, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, void* stream);
void cham_gemm_h2_launch_counts(long long* out8, int reset);
int cham_gemm_h2_set_nt_wide(int on);
/* TILE-BLOCKED planes (round 6).  A two-plane matrix [rows, ld] may be stored as [row tiles of 256][ld / 32 column blocks][256 rows][32
 * columns] (element (r, c) of a plane at ((r / 256) * (ld / 32) + c / 32) * 8192 + (r % 256) * 32 + c % 32; ld % 32 == 0), with
 * ceil(rows / 256) row tiles allocated per plane and the rows beyond the matrix ZERO: a (256-row x 32-column) block is then 16 KB contiguous,
 * and what the NT plane GEMMs (CAR forward / dgrad over the candidate rows, nar_model.py:384-388) fetch per LDS-DMA request - 16 rows x 32 k -
 * is 1 KB of whole 128-byte lines where the row-major operand gives sixteen half lines (HBM over-fetch 1.24-1.29 x and an L1 line-fill
 * limit in round 5).  The 16-byte pieces the kernels move are the same element sets in both layouts, so results are BIT-IDENTICAL.
 *   cham_gemm_h2b = cham_gemm_h2 + a_tiles / b_tiles (> 0: that operand is tile-blocked with this many row tiles allocated per plane, plane
 *   stride >= tiles * 256 * ld; 0: row-major) + dref_blocked (the saved activation's h plane likewise).  NT: A only (ld == K, K % 32 == 0,
 *   the 64-byte-piece kernel); TN: A and B independently.  cham_gemm_h2_launch_counts out8[3] / out8[4]: NT / TN launches with a blocked operand.
 *   cham_combine_fwd_h2b = cham_combine_fwd_h2 + blocked (the planes of the PreCAR output are written tile-blocked);
 *   cham_dm_mulpred_h2_blk = cham_dm_mulpred_h2 (ds1_scale_rec = w_scale_rec = NULL) / cham_dm_mulpred_h2h (both given) with the planes of
 *   the gradient at the CAR tanh written tile-blocked. */
int cham_h2b_block_elements(void);      /* elements from one block to the next: 8192 (+ the build's padding, 0 by default) */
int cham_gemm_h2b(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride, int ldb,
                  const float* b_scale, int tn, float* C, int ldc, int M, int N, int K, const float* bias, int act, const void* dref_h, int ldr,
                  int dact, int accumulate, float* workspace, size_t workspace_bytes, int splits_hint, int a_tiles, int b_tiles,
                  int dref_blocked, void* stream);
int cham_combine_fwd_h2b(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1p,
                         long long plane_stride, const void* scale_rec, int blocked, void* stream);
int cham_dm_mulpred_h2_blk(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const void* ds1_scale_rec,
                           const void* w_scale_rec, const float* Z2c, const float* pred, int C, int BT, int N, void* dZ2p,
                           long long out_plane_stride, const void* out_scale_rec, float* dpred_pre, float* col_part, void* stream);
/* producers of two-plane matrices (csrc/scorer.hip, csrc/dm_fused.hip): cham_combine_fwd_p3 / cham_mulpred_bwd_p3 / cham_dm_mulpred_p3
 * with the output written as (h, l) fp16 planes x the scale of `scale_rec` (filled BEFORE the call: max|U| + max|V| for the PreCAR output,
 * rownorm(dS1) x rownorm(Ws1) for the gradient at the CAR tanh) */
int cham_combine_fwd_h2(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1p,
                        long long plane_stride, const void* scale_rec, void* stream);
int cham_mulpred_bwd_h2(const float* dM, const float* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* dZ2p,
                        long long plane_stride, float* col_part, const void* scale_rec, void* stream);
int cham_dm_mulpred_h2(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const float* Z2c, const float* pred,
                       int C, int BT, int N, void* dZ2p, long long out_plane_stride, const void* out_scale_rec, float* dpred_pre,
                       float* col_part, void* stream);
/* cham_dm_mulpred_h2 with the kernel's own products (dS1 Ws1^T, K = 128) on two fp16 planes as well (round 5): Wh = plane 0 of
 * cham_split2h(Ws1 [C, K]) under the scale of w_scale_rec (planes w_plane_stride elements apart), dS1 split in registers with the scale of
 * ds1_scale_rec (a record covering max |dS1|: cham_h2_scale_rownorm2's second record) - three v_mfma_f32_32x32x16_f16 products instead of
 * six bf16 ones; same outputs, same shape limits. */
int cham_dm_mulpred_h2h(const float* dS1, int lds1, int K, const void* Wh, long long w_plane_stride, const void* ds1_scale_rec,
                        const void* w_scale_rec, const float* Z2c, const float* pred, int C, int BT, int N, void* dZ2p,
                        long long out_plane_stride, const void* out_scale_rec, float* dpred_pre, float* col_part, void* stream);
/* the bf16 configuration's twin (BASELINE configs[2]): every matrix a single bf16 array (dS1 [BT*(1+N), K] row stride lds1 ELEMENTS, Ws1b = the
 * bf16 shadow of Ws1 [C, K] as stored, Z2c and the output dZ2c [BT*(1+N), C]); replaces cham_gemm_b16(dS1, Ws1) + cham_mulpred_bwd_b16 and,
 * through col_part (sums of the STORED, bf16-rounded rows), the cham_colsum_b16 pass of the b2 gradient.  Same shape limits. */
int cham_dm_mulpred_b16(const void* dS1, int lds1, int K, const void* Ws1b, const void* Z2c, const float* pred, int C, int BT, int N, void* dZ2c,
                        float* dpred_pre, float* col_part, void* stream);

/* producers of plane-resident matrices (csrc/scorer.hip): the candidate rows of the PreCAR output leaky(U[b,t] + V[item])
 * (nar_model.py:356-405) and the gradient at the CAR tanh (autodiff of nar_model.py:478-495: dM * pred * (1 - Z2^2)) written as three
 * bf16 planes `plane_stride` elements apart; cham_mulpred_bwd_p3 also writes dpred_pre (as cham_mulpred_bwd) and, when col_part !=
 * NULL, col_part[bt] = the sum of position bt's 1 + N gradient rows (the b2 bias gradient = column sum of col_part). */
int cham_combine_fwd_p3(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1p,
                        long long plane_stride, void* stream);
int cham_mulpred_bwd_p3(const float* dM, const float* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* dZ2p,
                        long long plane_stride, float* col_part, void* stream);

/* Scorer layer-1 dgrad fused with the `cand (.) pred` backward (csrc/dm_fused.hip): what cham_gemm_f32x3(dS1, Ws1^T) followed by
 * cham_mulpred_bwd_p3 compute - the autodiff of matching_dense_layer_1 (nar_model.py:447-473), of cand (.) pred (:478-495) and of the
 * CAR tanh (:384-388) - without materialising the [B*T*(1+N), C] matrix between them: dS1 [BT*(1+N), K = 128] (row stride lds1), Wp =
 * plane 0 of cham_split3(Ws1 [C, K]) (planes w_plane_stride elements apart), Z2c [BT*(1+N), C], pred [BT, C] -> the three planes of the
 * gradient at the CAR tanh (dZ2p, planes out_plane_stride elements apart), dpred_pre [BT, C] and, when col_part != NULL, col_part[bt] =
 * the sum of position bt's rows.  Takes K == 128, C % 64 == 0 and 32 <= 1 + N <= 256; -EINVAL otherwise (the caller keeps the two
 * separate calls). */
int cham_dm_mulpred_p3(const float* dS1, int lds1, int K, const void* Wp, long long w_plane_stride, const float* Z2c, const float* pred,
                       int C, int BT, int N, void* dZ2p, long long out_plane_stride, float* dpred_pre, float* col_part, void* stream);

/* bf16-RESIDENT GEMMs of the bf16 configuration (csrc/gemm_b16.hip): the matrices with one row per candidate live in HBM as bf16
 * (weights: a bf16 shadow of the fp32 master copy); fp32 accumulation / bias / activation.
 *   transA = 0, transB = 1 (NT): A [M, lda], B [N, ldb] bf16, k contiguous; C bf16 (out_f32 = 0) or fp32 [M, ldc];
 *     epilogue + bias (fp32) and act, or x act'(dref) with dref = the saved bf16 activation [M, ldr] (dgrad, bf16 out);
 *   transA = 1, transB = 0 (TN): A stored [K, lda >= M], B stored [K, ldb >= N] bf16; C fp32 (+)=; split-K through `workspace`
 *     (splits_hint 1 = none, 0 = automatic; fixed-order reduction).
 * K % 8 == 0 (NT), M % 8 == 0 and N % 8 == 0 (TN), N % 4 == 0, leading dimensions % 8 == 0, 16-byte aligned bases. */
int cham_gemm_b16(const void* A, int lda, int transA, const void* B, int ldb, int transB, void* C, int ldc, int out_f32, int M,
                  int N, int K, const float* bias, int act, const void* dref, int ldr, int dact, int accumulate, float* workspace,
                  size_t workspace_bytes, int splits_hint, void* stream);
/* test / tuning aids: tile variant for N > 64 (0 = 128x128, 1 = 256x128 / 8 waves, 2 = 256x128 / 4 waves, -1 = automatic) and
 * launches per tile instance since the last reset (out8[0..4] = 128x128, 256x128/8, 256x128/4, 256x64, 256x32) */
void cham_gemm_b16_set_variant(int variant);
void cham_gemm_b16_launch_counts(long long* out8, int reset);

/* tuning hook (bench / autotune only): selects the tile configuration used for N > 64 */
void cham_gemm_set_variant(int variant);
/* test aid: launches per tile instance since the last reset - out16[0..4] = fp32 128x128, 256x128, 256x256, 256x64, 256x32;
 * out16[5] = the small-output TN kernel (round 6: plain C = A^T B with M <= 128 and K >= 512 - the scorer's layer-2 / layer-3 weight
 * gradients, the user-context share of the PreCAR kernel's - runs as fp32 FMAs on the VALU with 6-12 KB of LDS, so that it becomes
 * resident BESIDE the one-workgroup-per-CU plane GEMMs instead of behind them; cham_gemm_f32x3 hands such shapes over too);
 * out16[8..12] = the same tiles of the bf16 kernels; out16[14] / out16[15] = epilogue variant / K-splits of the last launch.
 * Parity tests assert that a shape ran on the instance it is meant to cover; bench.py names the kernel symbol of a timed launch. */
void cham_gemm_launch_counts(long long* out16, int reset);

/* --- PreCAR combine (factorised nar_model.py:356-405): Z1[row] = leaky(U[u(row)] + V[v(row)]) and its backward.
 * rows [row_begin, row_begin+row_count) of the BT + BT*(1+N) CAR rows (the BT clicked-input rows come first: they are
 * combined ahead of the candidates so that the recurrent branch can start early) */
int cham_combine_fwd(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, float* Z1,
                     long row_begin, long row_count, void* stream);
size_t cham_combine_bwd_workspace_bytes(int C, int BT, int N, int pmax);
int cham_combine_bwd(const float* dpre, int C, int BT, int N, int pmax, const int32_t* neg_slot, float* dU, float* dV,
                     float* workspace, size_t workspace_bytes, void* stream);
/* GROUP SUMS from the CAR dgrad's epilogue (round 6; the reference's tf.gradients sums d(tanh pre-activation) over a click's candidates when
 * it differentiates the tiled user context, nar_model.py:356-405).  The dgrad over the candidate rows (NT two-plane GEMM x leaky'(saved
 * activation), M = BT (N + 1) rows in groups of group_rows = N + 1 >= 32 consecutive rows) also writes, per 128-row chunk q and k-th group
 * of that chunk (group (128 q) / group_rows + k), the column sums of the chunk's rows of the group to
 * groupsum[(q * (127 / group_rows + 2) + k) * N + column];
 * cham_combine_bwd_gs = cham_combine_bwd with dU built from those pieces in chunk order: the 1 GB of candidate-row gradients is read once
 * (by the slot sums) instead of twice.  C / dV are bit-identical to the plain entry points'; dU is summed in another, equally fixed order.
 * cham_gemm_h2_launch_counts out8[5]: dgrad launches that produced group sums. */
size_t cham_gemm_h2_groupsum_bytes(int M, int N, int group_rows);
int cham_gemm_h2_dgrad_gs(const void* A, long long a_plane_stride, int lda, const float* a_scale, const void* B, long long b_plane_stride, int ldb,
                          const float* b_scale, float* C, int ldc, int M, int N, int K, const void* dref_h, int ldr, int a_tiles, int dref_blocked,
                          int group_rows, float* groupsum, size_t groupsum_bytes, void* stream);
int cham_combine_bwd_gs(const float* dpre, int C, int BT, int N, int pmax, const int32_t* neg_slot, float* dU, float* dV, float* workspace,
                        size_t workspace_bytes, const float* groupsum, size_t groupsum_bytes, void* stream);

/* --- bf16 configuration (BASELINE configs[2]): the matrices with one row per candidate are bf16 in HBM.  Twins of the entry points
 * above / below that read or write such matrices (same arithmetic in fp32 registers; only the storage type differs), plus the
 * two helpers the configuration adds:
 *   cham_mul_rows_b16: Mc[row] = bf16(Z2c[row] * pred[row / NC]) - the `cand (.) pred` product of nar_model.py:478-495 as a bf16 GEMM
 *     operand (the fp32 path fuses it into the GEMM staging as a row scale);
 *   cham_cast_b16: bf16 shadow of an fp32 weight [R, Cc]: dst = bf16(W) and / or dstT = bf16(W)^T (either may be NULL). */
int cham_combine_fwd_b16(const float* U, const float* V, int C, int BT, int N, int pmax, const int32_t* neg_slot, void* Z1c,
                         void* stream);
int cham_combine_bwd_b16(const float* dpre_in, const void* dpre_cand, int C, int BT, int N, int pmax, const int32_t* neg_slot,
                         float* dU, float* dV, float* workspace, size_t workspace_bytes, void* stream);
int cham_mulpred_bwd_b16(void* dM, const void* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* stream);
int cham_mul_rows_b16(const void* Z2c, const float* pred, int C, int BT, int NC, void* Mc, void* stream);
int cham_score_softmax_fwd_b16(const void* S3, int K3, const float* w4, const float* b4, int BT, int N, float tau,
                               const uint8_t* mask, float* logits, float* probs, float* nll, float novelty_reg_factor,
                               const int64_t* neg_ids, const float* pop_norm, float* nov_aux, void* stream);
int cham_score_softmax_bwd_b16(const void* S3, int K3, const float* w4, const float* probs, const uint8_t* mask, int BT, int N,
                               float tau, float sum_mask, float* ds, void* dS3, float novelty_reg_factor, const int64_t* neg_ids,
                               const float* pop_norm, const float* logits, const float* nov_aux, void* stream);
int cham_colsum_b16(const void* X, int ld, int R, int F, const float* w, float* out, int accumulate, float* workspace,
                    size_t workspace_bytes, void* stream);
int cham_cast_b16(const float* W, int R, int Cc, void* dst, void* dstT, void* stream);
/* dst[i] = (float)src[i] for a contiguous bf16 array of n elements (n % 4 == 0): the bf16 configuration with dropout_keep_prob < 1 */
int cham_upcast_b16(const void* src, size_t n, float* dst, void* stream);

/* --- K3 recurrent cell time steps: nar_model.py:1308-1361.
 * cell_kind 0 = UGRNN (tf.contrib.rnn.UGRNNCell, the reference's cell, :1317): xproj [B,T,2Hp] = x W_x + b (gate | candidate),
 *   Wh [Hp,2Hp]; saves hprev, G (gate), Cc (candidate); R / RH unused (NULL).
 * cell_kind 1 = GRU (tf.nn.rnn_cell.GRUCell, commented alternative at :1315; BASELINE config 4): xproj [B,T,3Hp] (r | u | c),
 *   Wh = W_gh [Hp,2Hp] followed by W_ch [Hp,Hp]; saves hprev, G (= u), Cc, R (= r) and RH (= r*h_prev, the wgrad operand).
 * backward: WhT = transpose(Wh) [2Hp,Hp] (GRU: transpose(W_gh) then transpose(W_ch)); dxproj = dL/d(pre-activations). */
int cham_rnn_fwd(int cell_kind, const float* xproj, const float* Wh, const int32_t* seq_len, int B, int T, int Hp, float* out,
                 float* hprev, float* G, float* Cc, float* R, float* RH, void* stream);
int cham_rnn_bwd(int cell_kind, const float* dout, const float* WhT, const int32_t* seq_len, int B, int T, int Hp,
                 const float* hprev, const float* G, const float* Cc, const float* R, float* dxproj, void* stream);
/* UGRNN time steps for Hp == 256 with the recurrent weights resident in LDS: eight cooperating workgroups per 32 sessions, each owning 32
 * hidden units (its columns of W_h in LDS for the whole sequence), exchanging their slice of h_t (forward) / of the gate and candidate
 * gradients (backward) through L2 with agent-scope release / acquire every step (csrc/rnn_coop.hip).  Same contract and outputs as
 * cham_rnn_fwd / cham_rnn_bwd with cell_kind 0 up to the fp32 summation order of the recurrent product; the backward takes W_h AS STORED
 * [Hp, 2Hp].  For steps whose critical path is the recurrent chain (ragged batches, the 32-session shard of a strong-scaling rank): ~8 us
 * per time step instead of ~33.  workspace: cham_rnn_coop_workspace_bytes(B, Hp) bytes, 256-byte aligned, zero-initialised once, one per
 * stream.  -EINVAL for Hp != 256 or B > 1024 (the caller keeps cham_rnn_fwd / _bwd).  cham_rnn_coop_timeouts: 1 if a workgroup ever gave
 * up a (bounded) spin on this workspace - synchronises the stream; tests and bench.py assert 0. */
size_t cham_rnn_coop_workspace_bytes(int B, int Hp);
int cham_ugrnn_fwd_coop(const float* xproj, const float* Wh, const int32_t* seq_len, int B, int T, int Hp, float* out, float* hprev, float* G,
                        float* Cc, void* workspace, size_t workspace_bytes, void* stream);
int cham_ugrnn_bwd_coop(const float* dout, const float* Wh, const int32_t* seq_len, int B, int T, int Hp, const float* hprev, const float* G,
                        const float* Cc, float* dxproj, void* workspace, size_t workspace_bytes, void* stream);
int cham_rnn_coop_timeouts(const void* workspace, int B, int Hp, void* stream);
/* step-wise fallback for rnn_units beyond the fused kernels' LDS budget (UGRNN Hp > 512; hypertuning goes to 1024,
 * nar_mlengine_hypertuning.yaml:28-33): the caller computes zh = h_{t-1} W_h (forward) / carry_next = direct + dzs W_h^T
 * (backward) with cham_gemm_f32 per time step; these do the UGRNN gate arithmetic, length masking and state carry */
int cham_ugrnn_point_fwd(const float* xproj, const float* zh, const int32_t* seq_len, int B, int T, int t, int Hp, float* h,
                         float* out, float* hprev, float* G, float* Cc, void* stream);
int cham_ugrnn_point_bwd(const float* dout, const float* carry, const int32_t* seq_len, int B, int T, int t, int Hp,
                         const float* hprev, const float* G, const float* Cc, float* dxproj, float* dzs, float* direct,
                         void* stream);
int cham_transpose_f32(const float* in, int rows, int cols, float* out, void* stream);
/* valid-position compaction (the mask of nar_model.py:231 applied as a row selection instead of a multiply): rows are
 * `words` 32-bit words wide; gather: dst[i] = src[pos[i]], scatter: dst[pos[i]] = src[i] (other rows of dst untouched) */
int cham_rows_gather(const void* src, const int32_t* pos, long n_rows, int words, void* dst, void* stream);
int cham_rows_scatter(const void* src, const int32_t* pos, long n_rows, int words, void* dst, void* stream);

/* --- K5 scoring tail + sampled softmax + masked NLL: nar_model.py:478-517, 639-667 */
int cham_mulpred_bwd(float* dM, const float* Z2c, const float* pred, int C, int BT, int N, float* dpred_pre, void* stream);
/* novelty_reg_factor > 0 adds the novelty regulariser of nar_model.py:673-683 to the per-click loss:
 * - factor * sum_n softmax(s_neg / tau)_n * (-log2 pop_norm[neg_ids[n]])  (neg_ids [BT,N], pop_norm [n_items], nov_aux [BT,3]
 * scratch carried from forward to backward; all three may be NULL when the factor is 0) */
int cham_score_softmax_fwd(const float* S3, int K3, const float* w4, const float* b4, int BT, int N, float tau,
                           const uint8_t* mask, float* logits, float* probs, float* nll, float novelty_reg_factor,
                           const int64_t* neg_ids, const float* pop_norm, float* nov_aux, void* stream);
int cham_score_softmax_bwd(const float* S3, int K3, const float* w4, const float* probs, const uint8_t* mask, int BT, int N,
                           float tau, float sum_mask, float* ds, float* dS3, float novelty_reg_factor, const int64_t* neg_ids,
                           const float* pop_norm, const float* logits, const float* nov_aux, void* stream);

/* evaluation: rank_items_by_predicted_prob, nar_model.py:777-794 (tf.nn.top_k over 1+N: descending, lowest index wins
 * ties).  pred_ids/pred_probs [BT, 1+N]; label_rank[bt] = 0-based rank of the positive, -1 for padded clicks - the input of
 * HitRate@n / MRR@n (metrics.py:40-66, 109-134; TF twins nar_model.py:826-835, 859-885) */
int cham_rank_items(const float* probs, const int64_t* label_next, const int64_t* neg_ids, const uint8_t* mask, int BT, int N,
                    int64_t* pred_ids, float* pred_probs, int32_t* label_rank, void* stream);

/* --- device-resident recent-clicks state: ClickedItemsState.update_items_state, clicked_items_state.py:187-250, fed by the
 * hook's flattening nar_model.py:1635-1646.  aci [B,T+1] = concat(item_clicked, label_last_item), event_ts [B,T] (both
 * already in HBM); buf_ids / buf_ts [buffer_size] newest first, zero padded; recent_pop [n_items] int32; pop_norm [n_items]
 * float32 = float32(max(recent_pop / (sum + 1), 1 / for_norm)) computed in float64; articles_pop [n_items] int64;
 * n_valid = device int32[2]: {rows retained in the buffer, clicks counted into recent_pop (the denominator of pop_norm is their
 * number + 1, clicked_items_state.py:242-246)}.  Bit-identical to the reference class. */
size_t cham_state_workspace_bytes(int B, int buffer_size);
int cham_state_update(const int64_t* aci, const int64_t* event_ts, int B, int T, double buffer_hours, int64_t* buf_ids,
                      int64_t* buf_ts, int buffer_size, int32_t* recent_pop, float* pop_norm, int64_t* articles_pop, int n_items,
                      int for_norm, int32_t* n_valid, void* workspace, size_t workspace_bytes, void* stream);

/* --- K7 regularisation loss, loss finalisation, TF Adam, bias-gradient column sums: nar_model.py:655, 660-667, 708-722 */
int cham_sumsq_partial(const float* params, size_t n_reg, float* partial, void* stream);
int cham_loss_finalize(const float* nll, int BT, float sum_mask, const float* sumsq_partial, float lambda, float* loss,
                       void* stream);
int cham_adam_tf(float* params, const float* grads, float* m, float* v, size_t n, size_t n_reg, float lambda, float lr_t,
                 float beta1, float beta2, float eps, void* stream);
/* gradient accumulation over session micro-batches of ONE optimizer step (row shards of the batch share pool, denominators
 * and weights - SURVEY.md 8e - so their flat gradient buffers add): acc = (first ? 0 : acc) + x;
 * loss_acc = [sum(xe) + reg, sum(xe), reg] */
int cham_accumulate(float* acc, const float* x, size_t n, int first, void* stream);
int cham_loss_accumulate(float* acc, const float* loss, int first, void* stream);
size_t cham_colsum_workspace_bytes(int R, int F);
int cham_colsum(const float* X, int ld, int R, int F, const float* w, float* out, int accumulate, float* workspace,
                size_t workspace_bytes, void* stream);


/* --- STEP SCALARS in device memory (round 6): a graph-capturable step.  The reference executes a training step as ONE session.run
 * (nar_model.py:1434-1470); the launch parameters of this library's step that change from one optimizer step to the next - the sampler
 * key (global step), the batch's max event time stamp (:235), sum(mask) (:664) and Adam's bias-corrected learning rate (:708) - can
 * live in one 32-byte DEVICE record {uint32 step, uint32 step_next, int64 max_ts, int64 reserved, float sum_mask, float lr_t} instead of
 * travelling by value: the `*_dev` entry points below are their by-value namesakes with that record in place of the scalar, the same
 * arithmetic on the same values (bit-identical results).  A step enqueued through them carries no per-step host value, so it can be
 * captured in a hipGraph once and replayed: the host then submits [copy the batch into the step's input slot, cham_step_scalars_set,
 * graph launch] instead of ~135 launches (chameleon_recsys_amd/nar/nar_model.py GraphedTrainStep).
 *   cham_step_scalars_set: a one-thread kernel taking the values by value (nothing on the host to keep alive), stream-ordered in front
 *   of the step; fields: bit 0 = the sampler keys, bit 1 = max_ts + sum_mask, bit 2 = lr_t (fields not selected keep their values).
 *   cham_neg_sample_dev: which = 0 -> .step, 1 -> .step_next (the next batch's negatives, drawn behind the current step). */
int cham_step_scalars_bytes(void);
int cham_step_scalars_set(void* rec, uint32_t step, uint32_t step_next, int64_t max_ts, float sum_mask, float lr_t, int fields, void* stream);
int cham_neg_sample_dev(const int64_t* aci, int Bg, int T1, const int64_t* buffer, int buf_size, uint32_t seed, const void* scalars, int which,
                        int row_begin, int row_count, int N, int n_from_buffer, int64_t* neg_ids, int32_t* neg_slot, int64_t* pool, int32_t* canon,
                        int32_t* meta, void* workspace, size_t workspace_bytes, void* stream);
int cham_step_ints_dev(const int64_t* ic_rows, const int64_t* ln_rows, const int64_t* pool, const int64_t* ets_rows, const void* scalars, int BT,
                       int pmax, const int32_t* seq_len_in, int B, const uint8_t* mask_in, int64_t* ids_all, int64_t* ref_ts, int32_t* seq_len,
                       uint8_t* mask, void* stream);
int cham_norm_stats_from_buffer_dev(const int64_t* buffer_ids, int n_prefix, const void* scalars, const int64_t* created, const float* pop_norm,
                                    float* scratch, float* stats, void* stream);
int cham_score_softmax_bwd_dev(const float* S3, int K3, const float* w4, const float* probs, const uint8_t* mask, int BT, int N, float tau,
                               const void* scalars, float* ds, float* dS3, float novelty_reg_factor, const int64_t* neg_ids, const float* pop_norm,
                               const float* logits, const float* nov_aux, void* stream);
int cham_score_softmax_bwd_b16_dev(const void* S3, int K3, const float* w4, const float* probs, const uint8_t* mask, int BT, int N, float tau,
                                   const void* scalars, float* ds, void* dS3, float novelty_reg_factor, const int64_t* neg_ids,
                                   const float* pop_norm, const float* logits, const float* nov_aux, void* stream);
int cham_loss_finalize_dev(const float* nll, int BT, const void* scalars, const float* sumsq_partial, float lambda, float* loss, void* stream);
int cham_adam_tf_dev(float* params, const float* grads, float* m, float* v, size_t n, size_t n_reg, float lambda, const void* scalars,
                     float beta1, float beta2, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
