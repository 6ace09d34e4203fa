/*
 * chameleon_tfrecord.h - C ABI of libchameleon_tfrecord.so: host-side codec of the NAR session files
 * (GZIP TFRecord of tf.train.SequenceExample).  Plain C++17 + zlib, no TensorFlow.
 *
 * The reference (gabrielspmoreira/chameleon_recsys) has no FFI; each entry point replaces a cluster of TF calls,
 * cited as file:line relative to the reference root.  All pointers are HOST pointers; the caller owns every
 * output buffer.  Return codes: 0 ok, 1 end of data, -22 bad argument, -5 I/O error, -74 CRC mismatch,
 * -71 malformed protobuf, -61 a configured feature is missing / has the wrong type or arity.
 * dtypes: 0 = int64, 1 = float32, 2 = bytes (context features only).
 */
#ifndef CHAMELEON_TFRECORD_H
#define CHAMELEON_TFRECORD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* masked CRC-32C of the TFRecord framing (TF spec; not in the reference tree) */
uint32_t cham_crc32c_masked(const uint8_t* data, uint64_t n);

/* --- input_fn: nar_module/nar/datasets.py:100-143 make_dataset + :35-82 parse_sequence_example -----------------
 * files are read in the given order, once (no shuffle / repeat, :124-143); names/dtypes = n_ctx context features
 * followed by n_seq sequence features; "session_size" (context) and "item_clicked" (sequence) are required
 * (nar_model.py:22-23).  A background thread decodes `prefetch` batches ahead (datasets.py:142 prefetch(1)). */
void* cham_sessions_open(const char* const* files, int n_files, const char* const* names, const int32_t* dtypes, int n_ctx,
                         int n_seq, int batch_size, int truncate_session_length, int check_crc, int prefetch, int* err);
/* advance to the next batch; B rows, T = max(min(len, truncate)) - 1 padded input length (datasets.py:134-135) */
int cham_sessions_next(void* h, int* B, int* T);
/* context feature i -> out[B] (int64 / float); session_size already min(session_size, truncate) (:56-57) */
int cham_sessions_ctx(void* h, int i, void* out);
/* bytes context feature (Adressa user_id): returns the total byte count; blob + offsets[B+1] filled when non-NULL */
int64_t cham_sessions_ctx_bytes(void* h, int i, char* blob, int64_t* offsets);
/* sequence feature i -> out[B, T], truncated, last element dropped (:60-63, :72-74), zero padded */
int cham_sessions_seq(void* h, int i, void* out);
/* label_next_item[B, T] = item_clicked[1:], label_last_item[B, 1] = item_clicked[-1:] (:67-69) */
int cham_sessions_labels(void* h, int64_t* label_next_item, int64_t* label_last_item);
void cham_sessions_close(void* h);

/* --- raw record stream: tf.data.TFRecordDataset(path, 'GZIP'), datasets.py:124 -------------------------------- */
void* cham_tfr_open(const char* path);
int cham_tfr_next(void* h, const uint8_t** data, uint64_t* len, int check_crc);
void cham_tfr_close(void* h);

/* --- writer: nar_module/nar/tf_records_management.py:12-32 (TFRecordWriter GZIP + SequenceExample builders) ----- */
void* cham_tfw_open(const char* path, int gzip_level);
int cham_tfw_write_record(void* h, const uint8_t* data, uint64_t len);
/* one session: ctx values by dtype from ctx_i64 / ctx_f32 / ctx_bytes (NUL-terminated), sequences row-major
 * seq_i64[n_seq][len] / seq_f32[n_seq][len]; one Feature per time step (make_sequential_feature, :12-19) */
int cham_tfw_write_session(void* h, const char* const* names, const int32_t* dtypes, int n_ctx, int n_seq, const int64_t* ctx_i64,
                           const float* ctx_f32, const char* const* ctx_bytes, const int64_t* seq_i64, const float* seq_f32, int len);
int cham_tfw_close(void* h);

#ifdef __cplusplus
}
#endif
#endif
