// libchameleon_tfrecord.so - host-side session-file codec of the NAR input pipeline (plain C++17 + zlib, no TF).
//
// Replaces, for the NAR training path, the TensorFlow pieces behind the reference's input_fn and writer:
//   * tf.data.TFRecordDataset(path, compression_type='GZIP')            nar_module/nar/datasets.py:124
//   * tf.parse_single_sequence_example(...) + truncate / shift / labels  nar_module/nar/datasets.py:35-82
//   * dataset.padded_batch(batch_size, zeros) + prefetch(1)              nar_module/nar/datasets.py:134-142
//   * tf.python_io.TFRecordWriter(GZIP) + SequenceExample builders       nar_module/nar/tf_records_management.py:12-32
// File format: one GZIP stream per file; TFRecord framing
//   u64le length | u32le masked_crc32c(length) | data | u32le masked_crc32c(data);
// payload = tf.train.SequenceExample protobuf (context: Features = 1, feature_lists: FeatureLists = 2), each time
// step one Feature holding exactly one value.
//
// A reader handle owns a background thread that inflates + decodes + pads the NEXT batch while the caller (the
// GPU step) consumes the current one (the reference's prefetch(1)).  C ABI: include/chameleon_tfrecord.h.
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

enum { DT_INT64 = 0, DT_FLOAT = 1, DT_BYTES = 2 };
enum {
    TFR_OK = 0,
    TFR_EOF = 1,
    TFR_ERR_ARG = -22,
    TFR_ERR_IO = -5,
    TFR_ERR_CRC = -74,      // EBADMSG
    TFR_ERR_PROTO = -71,    // EPROTO: malformed protobuf
    TFR_ERR_MISSING = -61,  // ENODATA: a configured feature is absent / has the wrong type or arity
};

// ------------------------------------------------------------------------------------------------ crc32c
uint32_t g_crc_table[8][256];
std::once_flag g_crc_once;
void crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFF];
}
uint32_t crc32c(const uint8_t* p, size_t n) {
    std::call_once(g_crc_once, crc_init);
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_table[7][lo & 0xFF] ^ g_crc_table[6][(lo >> 8) & 0xFF] ^ g_crc_table[5][(lo >> 16) & 0xFF] ^ g_crc_table[4][lo >> 24] ^
            g_crc_table[3][hi & 0xFF] ^ g_crc_table[2][(hi >> 8) & 0xFF] ^ g_crc_table[1][(hi >> 16) & 0xFF] ^ g_crc_table[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc_table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
inline uint32_t masked_crc(const uint8_t* p, size_t n) {
    const uint32_t c = crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// ------------------------------------------------------------------------------------------------ protobuf wire
struct Span { const uint8_t* p; const uint8_t* e; };

inline bool rd_varint(Span& s, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64 && s.p < s.e; shift += 7) {
        const uint8_t b = *s.p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}
inline bool rd_len(Span& s, Span& out) {
    uint64_t n;
    if (!rd_varint(s, n) || n > (uint64_t)(s.e - s.p)) return false;
    out.p = s.p; out.e = s.p + n; s.p += n;
    return true;
}
inline bool skip_field(Span& s, int wt) {
    uint64_t v; Span t;
    switch (wt) {
        case 0: return rd_varint(s, v);
        case 1: if (s.e - s.p < 8) return false; s.p += 8; return true;
        case 2: return rd_len(s, t);
        case 5: if (s.e - s.p < 4) return false; s.p += 4; return true;
        default: return false;
    }
}

// One decoded Feature: kind + values (bytes kept as a span into the record)
struct FeatureVal {
    int kind = -1;                  // DT_* or -1 = empty
    std::vector<int64_t> i64;
    std::vector<float> f32;
    std::vector<Span> bytes;
    void clear() { kind = -1; i64.clear(); f32.clear(); bytes.clear(); }
};

bool parse_feature(Span s, FeatureVal& out) {
    out.clear();
    while (s.p < s.e) {
        uint64_t tag;
        if (!rd_varint(s, tag)) return false;
        const int field = (int)(tag >> 3), wt = (int)(tag & 7);
        if (wt != 2 || field < 1 || field > 3) { if (!skip_field(s, wt)) return false; continue; }
        Span list;
        if (!rd_len(s, list)) return false;
        out.kind = field == 1 ? DT_BYTES : (field == 2 ? DT_FLOAT : DT_INT64);
        while (list.p < list.e) {
            uint64_t t2;
            if (!rd_varint(list, t2)) return false;
            const int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
            if (f2 != 1) { if (!skip_field(list, w2)) return false; continue; }
            if (field == 1) {                       // BytesList: repeated bytes value = 1
                Span b;
                if (w2 != 2 || !rd_len(list, b)) return false;
                out.bytes.push_back(b);
            } else if (field == 2) {                // FloatList: packed (LEN) or fixed32
                if (w2 == 2) {
                    Span pk;
                    if (!rd_len(list, pk) || ((pk.e - pk.p) & 3)) return false;
                    for (; pk.p < pk.e; pk.p += 4) { float f; memcpy(&f, pk.p, 4); out.f32.push_back(f); }
                } else if (w2 == 5) {
                    if (list.e - list.p < 4) return false;
                    float f; memcpy(&f, list.p, 4); list.p += 4; out.f32.push_back(f);
                } else return false;
            } else {                                // Int64List: packed varints or single varint
                if (w2 == 2) {
                    Span pk;
                    if (!rd_len(list, pk)) return false;
                    while (pk.p < pk.e) { uint64_t v; if (!rd_varint(pk, v)) return false; out.i64.push_back((int64_t)v); }
                } else if (w2 == 0) {
                    uint64_t v;
                    if (!rd_varint(list, v)) return false;
                    out.i64.push_back((int64_t)v);
                } else return false;
            }
        }
    }
    return true;
}

// map<string, X> entry: key = 1, value = 2
bool parse_map_entry(Span s, Span& key, Span& val) {
    key = Span{nullptr, nullptr}; val = Span{nullptr, nullptr};
    while (s.p < s.e) {
        uint64_t tag;
        if (!rd_varint(s, tag)) return false;
        const int field = (int)(tag >> 3), wt = (int)(tag & 7);
        if (wt == 2 && field == 1) { if (!rd_len(s, key)) return false; }
        else if (wt == 2 && field == 2) { if (!rd_len(s, val)) return false; }
        else if (!skip_field(s, wt)) return false;
    }
    return key.p != nullptr;
}

struct FeatureSpec { std::string name; int dtype; };

struct Schema {
    std::vector<FeatureSpec> ctx, seq;
    int idx_session_size = -1, idx_item_clicked = -1;
    int find(const std::vector<FeatureSpec>& v, const Span& key) const {
        const size_t n = key.e - key.p;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].name.size() == n && memcmp(v[i].name.data(), key.p, n) == 0) return (int)i;
        return -1;
    }
};

// One parsed session (already truncated; sequences keep their FULL truncated length L, the [:-1] / [1:] shifts are
// applied when the batch is padded).
struct Session {
    std::vector<int64_t> ctx_i64;                 // per ctx feature (0 for non-int)
    std::vector<float> ctx_f32;
    std::vector<std::string> ctx_bytes;
    std::vector<std::vector<int64_t>> seq_i64;    // per seq feature
    std::vector<std::vector<float>> seq_f32;
    int len = 0;                                  // truncated sequence length of item_clicked
};

int parse_session(const uint8_t* data, size_t n, const Schema& sc, int truncate, Session& out, FeatureVal& tmp) {
    out.ctx_i64.assign(sc.ctx.size(), 0);
    out.ctx_f32.assign(sc.ctx.size(), 0.f);
    out.ctx_bytes.assign(sc.ctx.size(), std::string());
    out.seq_i64.assign(sc.seq.size(), {});
    out.seq_f32.assign(sc.seq.size(), {});
    std::vector<char> seen_ctx(sc.ctx.size(), 0), seen_seq(sc.seq.size(), 0);
    Span s{data, data + n};
    while (s.p < s.e) {
        uint64_t tag;
        if (!rd_varint(s, tag)) return TFR_ERR_PROTO;
        const int field = (int)(tag >> 3), wt = (int)(tag & 7);
        if (wt != 2 || (field != 1 && field != 2)) { if (!skip_field(s, wt)) return TFR_ERR_PROTO; continue; }
        Span body;
        if (!rd_len(s, body)) return TFR_ERR_PROTO;
        while (body.p < body.e) {                  // Features.feature / FeatureLists.feature_list: map entries, field 1
            uint64_t t2;
            if (!rd_varint(body, t2)) return TFR_ERR_PROTO;
            if ((t2 >> 3) != 1 || (t2 & 7) != 2) { if (!skip_field(body, (int)(t2 & 7))) return TFR_ERR_PROTO; continue; }
            Span entry, key, val;
            if (!rd_len(body, entry) || !parse_map_entry(entry, key, val)) return TFR_ERR_PROTO;
            if (field == 1) {                      // context
                const int i = sc.find(sc.ctx, key);
                if (i < 0) continue;               // unconfigured features are ignored (datasets.py:39-42)
                if (!parse_feature(val, tmp)) return TFR_ERR_PROTO;
                if (tmp.kind != sc.ctx[i].dtype) return TFR_ERR_MISSING;
                if (tmp.kind == DT_INT64) { if (tmp.i64.size() != 1) return TFR_ERR_MISSING; out.ctx_i64[i] = tmp.i64[0]; }
                else if (tmp.kind == DT_FLOAT) { if (tmp.f32.size() != 1) return TFR_ERR_MISSING; out.ctx_f32[i] = tmp.f32[0]; }
                else { if (tmp.bytes.size() != 1) return TFR_ERR_MISSING; out.ctx_bytes[i].assign((const char*)tmp.bytes[0].p, tmp.bytes[0].e - tmp.bytes[0].p); }
                seen_ctx[i] = 1;
            } else {                               // feature_lists
                const int i = sc.find(sc.seq, key);
                if (i < 0) continue;
                if (sc.seq[i].dtype == DT_BYTES) return TFR_ERR_ARG;      // no bytes sequence feature on the NAR path
                Span fl = val;                     // FeatureList: repeated Feature feature = 1
                int count = 0;
                while (fl.p < fl.e) {
                    uint64_t t3;
                    if (!rd_varint(fl, t3)) return TFR_ERR_PROTO;
                    if ((t3 >> 3) != 1 || (t3 & 7) != 2) { if (!skip_field(fl, (int)(t3 & 7))) return TFR_ERR_PROTO; continue; }
                    Span fe;
                    if (!rd_len(fl, fe)) return TFR_ERR_PROTO;
                    if (count >= truncate) { ++count; continue; }          // datasets.py:60-63 truncation
                    if (!parse_feature(fe, tmp)) return TFR_ERR_PROTO;
                    // FixedLenSequenceFeature(shape=[]): exactly one value per step
                    if (tmp.kind != sc.seq[i].dtype) return TFR_ERR_MISSING;
                    if (tmp.kind == DT_INT64) { if (tmp.i64.size() != 1) return TFR_ERR_MISSING; out.seq_i64[i].push_back(tmp.i64[0]); }
                    else { if (tmp.f32.size() != 1) return TFR_ERR_MISSING; out.seq_f32[i].push_back(tmp.f32[0]); }
                    ++count;
                }
                seen_seq[i] = 1;
            }
        }
    }
    for (size_t i = 0; i < seen_ctx.size(); ++i) if (!seen_ctx[i]) return TFR_ERR_MISSING;
    for (size_t i = 0; i < seen_seq.size(); ++i) if (!seen_seq[i]) return TFR_ERR_MISSING;
    out.len = (int)(sc.seq[sc.idx_item_clicked].dtype == DT_INT64 ? out.seq_i64[sc.idx_item_clicked].size() : 0);
    for (size_t i = 0; i < sc.seq.size(); ++i) {
        const size_t li = sc.seq[i].dtype == DT_INT64 ? out.seq_i64[i].size() : out.seq_f32[i].size();
        if ((int)li != out.len) return TFR_ERR_MISSING;      // padded_batch would still work, the model would not
    }
    if (out.len < 1) return TFR_ERR_MISSING;
    // datasets.py:56-57 session_size <- min(session_size, truncate)
    if (out.ctx_i64[sc.idx_session_size] > truncate) out.ctx_i64[sc.idx_session_size] = truncate;
    return TFR_OK;
}

// ------------------------------------------------------------------------------------------------ gz record stream
struct RecordReader {
    gzFile gz = nullptr;
    std::vector<uint8_t> buf;
    int open(const char* path) {
        gz = gzopen(path, "rb");
        if (!gz) return TFR_ERR_IO;
        gzbuffer(gz, 1 << 20);
        return TFR_OK;
    }
    void close() { if (gz) { gzclose(gz); gz = nullptr; } }
    // like next(), but the record is left with its 4-byte data CRC appended and that CRC is NOT verified (worker threads do it)
    int next_raw(bool check_len_crc) {
        uint8_t hdr[12];
        const int got = gzread(gz, hdr, 12);
        if (got == 0) return TFR_EOF;
        if (got != 12) return TFR_ERR_IO;
        uint64_t len; uint32_t lcrc;
        memcpy(&len, hdr, 8); memcpy(&lcrc, hdr + 8, 4);
        if (check_len_crc && masked_crc(hdr, 8) != lcrc) return TFR_ERR_CRC;
        if (len > (1ull << 31)) return TFR_ERR_CRC;
        buf.resize(len + 4);
        size_t off = 0;
        while (off < len + 4) {
            const unsigned want = (unsigned)std::min<size_t>(len + 4 - off, 1u << 30);
            const int r = gzread(gz, buf.data() + off, want);
            if (r <= 0) return TFR_ERR_IO;
            off += r;
        }
        return TFR_OK;
    }
    // TFR_OK + record in buf, TFR_EOF, or an error
    int next(bool check_crc) {
        uint8_t hdr[12];
        const int got = gzread(gz, hdr, 12);
        if (got == 0) return TFR_EOF;
        if (got != 12) return TFR_ERR_IO;
        uint64_t len; uint32_t lcrc;
        memcpy(&len, hdr, 8); memcpy(&lcrc, hdr + 8, 4);
        if (check_crc && masked_crc(hdr, 8) != lcrc) return TFR_ERR_CRC;
        if (len > (1ull << 31)) return TFR_ERR_CRC;
        buf.resize(len + 4);
        size_t off = 0;
        while (off < len + 4) {
            const unsigned want = (unsigned)std::min<size_t>(len + 4 - off, 1u << 30);
            const int r = gzread(gz, buf.data() + off, want);
            if (r <= 0) return TFR_ERR_IO;
            off += r;
        }
        uint32_t dcrc;
        memcpy(&dcrc, buf.data() + len, 4);
        if (check_crc && masked_crc(buf.data(), len) != dcrc) return TFR_ERR_CRC;
        buf.resize(len);
        return TFR_OK;
    }
};

// ------------------------------------------------------------------------------------------------ batches
struct Batch {
    int B = 0, T = 0;
    std::vector<Session> sessions;
};

// Raw records of one batch: the inflate thread only splits the stream (and checks the 12-byte header CRC, which guards the
// length); the data CRC and the protobuf decode run on the worker pool.
struct RawBatch {
    uint64_t seq = 0;
    int file_idx = 0;
    std::vector<uint8_t> bytes;            // records back to back: data followed by its 4-byte masked CRC
    std::vector<size_t> off;               // start of record i; off.back() = bytes.size()
};

// tf.data pipeline of datasets.py:118-142: TFRecordDataset(files, 'GZIP') -> map(parse, num_parallel_calls = cpu_count) ->
// padded_batch -> prefetch.  GZIP inflate is sequential per stream, so ONE thread inflates and cuts the record stream into
// batches of raw records; `n_workers` threads verify the data CRC and decode the SequenceExamples of whole batches in
// parallel; a reorder buffer hands the batches out in stream order (a batch may span two files, like the reference's).
struct SessionReader {
    Schema sc;
    std::vector<std::string> files;
    int batch_size = 128, truncate = 20, check_crc = 1, prefetch = 2, n_workers = 1;
    std::thread th_reader;
    std::vector<std::thread> th_workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_space, cv_get;
    std::deque<std::unique_ptr<RawBatch>> work;              // inflated, not yet decoded
    std::map<uint64_t, std::unique_ptr<Batch>> ready;        // decoded, waiting for their turn
    uint64_t next_out = 0, n_raw = 0;                        // next batch the consumer takes / number of raw batches produced
    int in_decode = 0;
    bool reader_done = false, stop = false;
    int error = TFR_OK;
    uint64_t error_seq = ~0ull;                              // batches before this one are still delivered
    std::string error_file;
    std::unique_ptr<Batch> cur;
    std::atomic<long long> ns_inflate{0}, ns_decode{0}, ns_reader_wait{0}, ns_batcher_wait{0};      // CHAM_TFRECORD_STATS=1 prints them at close

    void fail(int err, uint64_t seq, const std::string& file) {      // mu held
        if (seq < error_seq) { error = err; error_seq = seq; error_file = file; }
    }
    static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    bool push_raw(std::unique_ptr<RawBatch> rb) {
        const long long t0 = now_ns();
        std::unique_lock<std::mutex> lk(mu);
        // bound the batches in flight (raw + being decoded + decoded): prefetch depth + one per worker
        cv_space.wait(lk, [&] { return stop || (int)(work.size() + ready.size()) + in_decode < prefetch + n_workers; });
        ns_reader_wait += now_ns() - t0;
        if (stop) return false;
        rb->seq = n_raw++;
        work.push_back(std::move(rb));
        cv_work.notify_one();
        return true;
    }
    // ---- stage 1: inflate.  One gzip stream is sequential, but the files of a chunk (hourly files) are independent streams:
    // up to n_inflate files are inflated concurrently, each into its own bounded queue of record chunks; the batcher below walks
    // the files IN ORDER and cuts the concatenated record stream into batches (a batch may span two files, like the reference's
    // TFRecordDataset(files) -> padded_batch).
    struct FileStream {
        std::deque<std::unique_ptr<RawBatch>> chunks;      // up to batch_size records each
        bool done = false;
        int err = TFR_OK;
    };
    int n_inflate = 1;
    std::deque<FileStream> streams;            // (deque: FileStream holds move-only chunks; never reallocated after start())
    std::vector<std::thread> th_inflate;
    std::mutex mu_in;
    std::condition_variable cv_in_put, cv_in_get;
    size_t next_file = 0;

    void inflate_loop() {
        for (;;) {
            size_t fi;
            {
                std::lock_guard<std::mutex> lk(mu_in);
                if (stop || next_file >= files.size()) return;
                fi = next_file++;
            }
            RecordReader rr;
            int err = rr.open(files[fi].c_str());
            auto ch = std::make_unique<RawBatch>();
            auto flush = [&](bool last) -> bool {
                std::unique_lock<std::mutex> lk(mu_in);
                cv_in_put.wait(lk, [&] { return stop || streams[fi].chunks.size() < 4; });
                if (stop) return false;
                if (ch->off.size() > 1) streams[fi].chunks.push_back(std::move(ch));
                if (last) { streams[fi].done = true; streams[fi].err = err; }
                cv_in_get.notify_all();
                ch = std::make_unique<RawBatch>();
                return true;
            };
            while (err == TFR_OK) {
                const long long t0 = now_ns();
                const int r = rr.next_raw(check_crc != 0);
                ns_inflate += now_ns() - t0;
                if (r == TFR_EOF) break;
                if (r != TFR_OK) { err = r; break; }
                if (ch->off.empty()) ch->off.push_back(0);
                ch->bytes.insert(ch->bytes.end(), rr.buf.begin(), rr.buf.end());
                ch->off.push_back(ch->bytes.size());
                if ((int)ch->off.size() - 1 == batch_size && !flush(false)) { rr.close(); return; }
            }
            rr.close();
            if (!flush(true)) return;
        }
    }
    // ---- stage 2: batcher (cheap: byte-range appends)
    void read_loop() {
        auto rb = std::make_unique<RawBatch>();
        int err = TFR_OK;
        std::string err_file;
        for (size_t fi = 0; fi < files.size() && err == TFR_OK; ++fi) {
            for (;;) {
                std::unique_ptr<RawBatch> ch;
                bool file_done = false;
                {
                    std::unique_lock<std::mutex> lk(mu_in);
                    const long long t0 = now_ns();
                    cv_in_get.wait(lk, [&] { return stop || !streams[fi].chunks.empty() || streams[fi].done; });
                    ns_batcher_wait += now_ns() - t0;
                    if (stop) return;
                    if (!streams[fi].chunks.empty()) { ch = std::move(streams[fi].chunks.front()); streams[fi].chunks.pop_front(); cv_in_put.notify_all(); }
                    else { file_done = true; if (streams[fi].err != TFR_OK) { err = streams[fi].err; err_file = files[fi]; } }
                }
                if (file_done) break;
                const int n = (int)ch->off.size() - 1;
                int i = 0;
                while (i < n) {
                    if (rb->off.empty()) { rb->off.push_back(0); rb->file_idx = (int)fi; }
                    const int have = (int)rb->off.size() - 1;
                    const int take = std::min(n - i, batch_size - have);
                    const size_t b0 = ch->off[i], b1 = ch->off[i + take], base = rb->bytes.size();
                    rb->bytes.insert(rb->bytes.end(), ch->bytes.begin() + b0, ch->bytes.begin() + b1);
                    for (int k = 1; k <= take; ++k) rb->off.push_back(base + (ch->off[i + k] - b0));
                    i += take;
                    if ((int)rb->off.size() - 1 == batch_size) {
                        if (!push_raw(std::move(rb))) return;
                        rb = std::make_unique<RawBatch>();
                    }
                }
            }
        }
        if (err == TFR_OK && rb->off.size() > 1) { if (!push_raw(std::move(rb))) return; }     // last batch is short (no drop_remainder)
        std::lock_guard<std::mutex> lk(mu);
        if (err != TFR_OK) fail(err, n_raw, err_file);       // everything cut before the error is still delivered
        reader_done = true;
        cv_work.notify_all(); cv_get.notify_all();
    }
    void work_loop() {
        FeatureVal tmp;
        for (;;) {
            std::unique_ptr<RawBatch> rb;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !work.empty() || reader_done; });
                if (stop || (work.empty() && reader_done)) return;
                rb = std::move(work.front());
                work.pop_front();
                ++in_decode;
            }
            const long long t0 = now_ns();
            auto b = std::make_unique<Batch>();
            int err = TFR_OK;
            const int n = (int)rb->off.size() - 1;
            b->sessions.resize(n);
            int L = 0;
            for (int i = 0; i < n && err == TFR_OK; ++i) {
                const uint8_t* p = rb->bytes.data() + rb->off[i];
                const size_t len = rb->off[i + 1] - rb->off[i] - 4;
                uint32_t dcrc;
                memcpy(&dcrc, p + len, 4);
                if (check_crc && masked_crc(p, len) != dcrc) { err = TFR_ERR_CRC; break; }
                err = parse_session(p, len, sc, truncate, b->sessions[i], tmp);
                L = std::max(L, b->sessions[i].len);
            }
            b->B = n;
            b->T = L - 1;                                  // inputs drop their last element (datasets.py:72-74)
            ns_decode += now_ns() - t0;
            std::lock_guard<std::mutex> lk(mu);
            --in_decode;
            if (err != TFR_OK) fail(err, rb->seq, files[rb->file_idx]);
            else ready[rb->seq] = std::move(b);
            cv_get.notify_all();
        }
    }
    void start() {
        for (size_t i = 0; i < files.size(); ++i) streams.emplace_back();
        for (int i = 0; i < n_inflate; ++i) th_inflate.emplace_back([this] { inflate_loop(); });
        th_reader = std::thread([this] { read_loop(); });
        for (int i = 0; i < n_workers; ++i) th_workers.emplace_back([this] { work_loop(); });
    }
    // TFR_OK / TFR_EOF / error
    int next() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            auto it = ready.find(next_out);
            if (it != ready.end()) {
                cur = std::move(it->second);
                ready.erase(it);
                ++next_out;
                cv_space.notify_all();
                return TFR_OK;
            }
            if (error != TFR_OK && next_out >= error_seq) { cur.reset(); return error; }
            if (reader_done && next_out >= n_raw) { cur.reset(); return TFR_EOF; }
            cv_get.wait(lk);
        }
    }
    ~SessionReader() {
        { std::lock_guard<std::mutex> lk(mu); std::lock_guard<std::mutex> lk2(mu_in); stop = true; }
        cv_space.notify_all(); cv_work.notify_all(); cv_get.notify_all(); cv_in_put.notify_all(); cv_in_get.notify_all();
        for (auto& t : th_inflate) if (t.joinable()) t.join();
        if (th_reader.joinable()) th_reader.join();
        for (auto& t : th_workers) if (t.joinable()) t.join();
        if (getenv("CHAM_TFRECORD_STATS"))
            fprintf(stderr, "tfrecord reader: %llu batches; inflate %.1f ms over %d threads; batcher waited %.1f ms for records, %.1f ms for "
                            "queue space; decode %.1f ms over %d workers\n", (unsigned long long)n_raw, ns_inflate / 1e6, n_inflate,
                    ns_batcher_wait / 1e6, ns_reader_wait / 1e6, ns_decode / 1e6, n_workers);
    }
};

// ------------------------------------------------------------------------------------------------ writer
struct Writer { gzFile gz = nullptr; };

void put_varint(std::string& o, uint64_t v) {
    while (v >= 0x80) { o.push_back((char)((v & 0x7F) | 0x80)); v >>= 7; }
    o.push_back((char)v);
}
void put_len_field(std::string& o, int field, const std::string& body) {
    put_varint(o, ((uint64_t)field << 3) | 2);
    put_varint(o, body.size());
    o += body;
}
std::string enc_feature_i64(int64_t v) {          // Feature{int64_list{value:[v]}} (packed)
    std::string pk; put_varint(pk, (uint64_t)v);
    std::string list; put_len_field(list, 1, pk);
    std::string f; put_len_field(f, 3, list);
    return f;
}
std::string enc_feature_f32(float v) {
    std::string pk((const char*)&v, 4);
    std::string list; put_len_field(list, 1, pk);
    std::string f; put_len_field(f, 2, list);
    return f;
}
std::string enc_feature_bytes(const char* p, size_t n) {
    std::string list; put_len_field(list, 1, std::string(p, n));
    std::string f; put_len_field(f, 1, list);
    return f;
}
std::string enc_map_entry(const std::string& key, const std::string& val) {
    std::string e; put_len_field(e, 1, key); put_len_field(e, 2, val);
    return e;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

uint32_t cham_crc32c_masked(const uint8_t* data, uint64_t n) { return masked_crc(data, n); }

// ---- reader -----------------------------------------------------------------------------------------------
// names: n_ctx context feature names then n_seq sequence feature names; dtypes likewise (0 int64, 1 float, 2 bytes)
void* cham_sessions_open(const char* const* files, int n_files, const char* const* names, const int32_t* dtypes, int n_ctx,
                         int n_seq, int batch_size, int truncate_session_length, int check_crc, int prefetch, int* err) {
    int e = TFR_OK;
    std::unique_ptr<SessionReader> r(new SessionReader());
    if (!files || n_files <= 0 || !names || !dtypes || n_ctx <= 0 || n_seq <= 0 || batch_size <= 0 || truncate_session_length < 2) e = TFR_ERR_ARG;
    if (e == TFR_OK) {
        for (int i = 0; i < n_files; ++i) r->files.emplace_back(files[i]);
        for (int i = 0; i < n_ctx; ++i) {
            r->sc.ctx.push_back({names[i], dtypes[i]});
            if (r->sc.ctx.back().name == "session_size") r->sc.idx_session_size = i;
        }
        for (int i = 0; i < n_seq; ++i) {
            r->sc.seq.push_back({names[n_ctx + i], dtypes[n_ctx + i]});
            if (r->sc.seq.back().name == "item_clicked") r->sc.idx_item_clicked = i;
        }
        // required keys: nar_model.py:22-23, 219-233
        if (r->sc.idx_session_size < 0 || r->sc.idx_item_clicked < 0 || r->sc.ctx[r->sc.idx_session_size].dtype != DT_INT64 ||
            r->sc.seq[r->sc.idx_item_clicked].dtype != DT_INT64)
            e = TFR_ERR_ARG;
    }
    if (err) *err = e;
    if (e != TFR_OK) return nullptr;
    r->batch_size = batch_size; r->truncate = truncate_session_length; r->check_crc = check_crc;
    r->prefetch = prefetch > 0 ? prefetch : 1;
    // decode threads: CHAM_TFRECORD_THREADS, default min(this process's share of the hardware threads, 16) (datasets.py:118-120 uses
    // cpu_count() map calls).  Under data parallelism every rank of the node decodes the GLOBAL batch stream (the integer
    // tensors are replicated by reading the same files): the share is hardware threads / LOCAL_WORLD_SIZE (torchrun's count of ranks on
    // this node; WORLD_SIZE when that is absent), less the inflate threads, so 8 ranks do not start 8 x 20 threads on one host.
    // 16 decode + 4 inflate threads: 369 k full-length / 981 k G1-like sessions/s, profiles/r02_decode_throughput.json
    int nw = 0;
    if (const char* e = getenv("CHAM_TFRECORD_THREADS")) nw = atoi(e);
    if (nw <= 0) {
        int ranks = 1;
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = atoi(e);
        else if (const char* e2 = getenv("WORLD_SIZE")) ranks = atoi(e2);
        if (ranks < 1) ranks = 1;
        int share = (int)std::thread::hardware_concurrency() / ranks;
        nw = share > 20 ? 16 : (share * 4) / 5;          // (a fifth of the share is left to the inflate threads below)
        if (nw > 16) nw = 16;
        if (nw < 1) nw = 1;
    }
    r->n_workers = nw;
    int ni = 0;
    if (const char* e = getenv("CHAM_TFRECORD_INFLATE_THREADS")) ni = atoi(e);
    if (ni <= 0) ni = nw >= 8 ? 4 : (nw >= 4 ? 2 : 1);
    if (ni > n_files) ni = n_files;
    r->n_inflate = ni;
    SessionReader* raw = r.release();
    raw->start();
    return raw;
}

// advances to the next batch: 0 = ok (B, T set), 1 = end of data, < 0 error
int cham_sessions_next(void* h, int* B, int* T) {
    if (!h) return TFR_ERR_ARG;
    SessionReader* r = (SessionReader*)h;
    const int rc = r->next();
    if (rc == TFR_OK) { if (B) *B = r->cur->B; if (T) *T = r->cur->T; }
    return rc;
}

// context feature i of the current batch -> out[B] (int64 or float)
int cham_sessions_ctx(void* h, int i, void* out) {
    SessionReader* r = (SessionReader*)h;
    if (!r || !r->cur || i < 0 || i >= (int)r->sc.ctx.size() || !out) return TFR_ERR_ARG;
    const int dt = r->sc.ctx[i].dtype;
    if (dt == DT_BYTES) return TFR_ERR_ARG;
    for (int b = 0; b < r->cur->B; ++b) {
        if (dt == DT_INT64) ((int64_t*)out)[b] = r->cur->sessions[b].ctx_i64[i];
        else ((float*)out)[b] = r->cur->sessions[b].ctx_f32[i];
    }
    return TFR_OK;
}

// bytes context feature: total byte count (call with blob == NULL first), then blob + offsets[B+1]
int64_t cham_sessions_ctx_bytes(void* h, int i, char* blob, int64_t* offsets) {
    SessionReader* r = (SessionReader*)h;
    if (!r || !r->cur || i < 0 || i >= (int)r->sc.ctx.size() || r->sc.ctx[i].dtype != DT_BYTES) return TFR_ERR_ARG;
    int64_t tot = 0;
    for (int b = 0; b < r->cur->B; ++b) {
        const std::string& s = r->cur->sessions[b].ctx_bytes[i];
        if (blob) { memcpy(blob + tot, s.data(), s.size()); offsets[b] = tot; }
        tot += (int64_t)s.size();
    }
    if (blob) offsets[r->cur->B] = tot;
    return tot;
}

// sequence feature i -> out[B, T] zero padded, inputs = seq[:-1] (datasets.py:72-74)
int cham_sessions_seq(void* h, int i, void* out) {
    SessionReader* r = (SessionReader*)h;
    if (!r || !r->cur || i < 0 || i >= (int)r->sc.seq.size() || !out) return TFR_ERR_ARG;
    const int T = r->cur->T, dt = r->sc.seq[i].dtype;
    for (int b = 0; b < r->cur->B; ++b) {
        const Session& s = r->cur->sessions[b];
        const int n = s.len - 1;
        if (dt == DT_INT64) {
            int64_t* o = (int64_t*)out + (size_t)b * T;
            for (int t = 0; t < T; ++t) o[t] = t < n ? s.seq_i64[i][t] : 0;
        } else {
            float* o = (float*)out + (size_t)b * T;
            for (int t = 0; t < T; ++t) o[t] = t < n ? s.seq_f32[i][t] : 0.f;
        }
    }
    return TFR_OK;
}

// labels: label_next_item[B, T] = item_clicked[1:], label_last_item[B, 1] = item_clicked[-1:] (datasets.py:67-69)
int cham_sessions_labels(void* h, int64_t* label_next_item, int64_t* label_last_item) {
    SessionReader* r = (SessionReader*)h;
    if (!r || !r->cur || !label_next_item || !label_last_item) return TFR_ERR_ARG;
    const int T = r->cur->T, ic = r->sc.idx_item_clicked;
    for (int b = 0; b < r->cur->B; ++b) {
        const Session& s = r->cur->sessions[b];
        int64_t* o = label_next_item + (size_t)b * T;
        for (int t = 0; t < T; ++t) o[t] = (t + 1 < s.len) ? s.seq_i64[ic][t + 1] : 0;
        label_last_item[b] = s.seq_i64[ic][s.len - 1];
    }
    return TFR_OK;
}

void cham_sessions_close(void* h) { delete (SessionReader*)h; }

// ---- raw record access (format tests, tooling) -------------------------------------------------------------
void* cham_tfr_open(const char* path) {
    RecordReader* r = new RecordReader();
    if (r->open(path) != TFR_OK) { delete r; return nullptr; }
    return r;
}
int cham_tfr_next(void* h, const uint8_t** data, uint64_t* len, int check_crc) {
    RecordReader* r = (RecordReader*)h;
    if (!r) return TFR_ERR_ARG;
    const int rc = r->next(check_crc != 0);
    if (rc == TFR_OK) { *data = r->buf.data(); *len = r->buf.size(); }
    return rc;
}
void cham_tfr_close(void* h) { RecordReader* r = (RecordReader*)h; if (r) { r->close(); delete r; } }

// ---- writer ------------------------------------------------------------------------------------------------
void* cham_tfw_open(const char* path, int gzip_level) {
    char mode[8];
    snprintf(mode, sizeof mode, "wb%d", gzip_level >= 0 && gzip_level <= 9 ? gzip_level : 6);
    gzFile gz = gzopen(path, mode);
    if (!gz) return nullptr;
    Writer* w = new Writer();
    w->gz = gz;
    return w;
}
int cham_tfw_write_record(void* h, const uint8_t* data, uint64_t len) {
    Writer* w = (Writer*)h;
    if (!w || !data) return TFR_ERR_ARG;
    uint8_t hdr[12];
    memcpy(hdr, &len, 8);
    const uint32_t lc = masked_crc(hdr, 8), dc = masked_crc(data, len);
    memcpy(hdr + 8, &lc, 4);
    if (gzwrite(w->gz, hdr, 12) != 12) return TFR_ERR_IO;
    if (len && gzwrite(w->gz, data, (unsigned)len) != (int)len) return TFR_ERR_IO;
    if (gzwrite(w->gz, &dc, 4) != 4) return TFR_ERR_IO;
    return TFR_OK;
}
// One session as a SequenceExample (tf_records_management.py:12-19 make_sequential_feature: one Feature per step).
// ctx values: int64 ctx_i64[n_ctx] / float ctx_f32[n_ctx] / bytes ctx_bytes[n_ctx] (by dtype);
// seq values: seq_i64[n_seq][len] / seq_f32[n_seq][len] row-major.
int cham_tfw_write_session(void* h, const char* const* names, const int32_t* dtypes, int n_ctx, int n_seq, const int64_t* ctx_i64,
                           const float* ctx_f32, const char* const* ctx_bytes, const int64_t* seq_i64, const float* seq_f32, int len) {
    if (!h || !names || !dtypes || len < 0) return TFR_ERR_ARG;
    std::string ctx, fls;
    for (int i = 0; i < n_ctx; ++i) {
        std::string f;
        if (dtypes[i] == DT_INT64) f = enc_feature_i64(ctx_i64[i]);
        else if (dtypes[i] == DT_FLOAT) f = enc_feature_f32(ctx_f32[i]);
        else f = enc_feature_bytes(ctx_bytes[i], strlen(ctx_bytes[i]));
        put_len_field(ctx, 1, enc_map_entry(names[i], f));
    }
    for (int i = 0; i < n_seq; ++i) {
        std::string fl;
        for (int t = 0; t < len; ++t) {
            if (dtypes[n_ctx + i] == DT_INT64) put_len_field(fl, 1, enc_feature_i64(seq_i64[(size_t)i * len + t]));
            else if (dtypes[n_ctx + i] == DT_FLOAT) put_len_field(fl, 1, enc_feature_f32(seq_f32[(size_t)i * len + t]));
            else return TFR_ERR_ARG;
        }
        put_len_field(fls, 1, enc_map_entry(names[n_ctx + i], fl));
    }
    std::string ex;
    put_len_field(ex, 1, ctx);
    put_len_field(ex, 2, fls);
    return cham_tfw_write_record(h, (const uint8_t*)ex.data(), ex.size());
}
int cham_tfw_close(void* h) {
    Writer* w = (Writer*)h;
    if (!w) return TFR_ERR_ARG;
    const int rc = gzclose(w->gz);
    delete w;
    return rc == Z_OK ? TFR_OK : TFR_ERR_IO;
}

}  // extern "C"
