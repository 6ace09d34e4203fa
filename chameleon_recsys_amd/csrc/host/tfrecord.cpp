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
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
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

struct SessionReader {
    Schema sc;
    std::vector<std::string> files;
    int batch_size = 128, truncate = 20, check_crc = 1, prefetch = 2;
    // producer state
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<std::unique_ptr<Batch>> q;
    bool done = false, stop = false;
    int error = TFR_OK;
    std::string error_file;
    std::unique_ptr<Batch> cur;

    void produce() {
        RecordReader rr;
        FeatureVal tmp;
        auto b = std::make_unique<Batch>();
        int err = TFR_OK;
        for (size_t fi = 0; fi < files.size() && err == TFR_OK; ++fi) {
            if ((err = rr.open(files[fi].c_str())) != TFR_OK) { error_file = files[fi]; break; }
            for (;;) {
                const int r = rr.next(check_crc != 0);
                if (r == TFR_EOF) break;
                if (r != TFR_OK) { err = r; error_file = files[fi]; break; }
                b->sessions.emplace_back();
                const int pr = parse_session(rr.buf.data(), rr.buf.size(), sc, truncate, b->sessions.back(), tmp);
                if (pr != TFR_OK) { err = pr; error_file = files[fi]; break; }
                if ((int)b->sessions.size() == batch_size) {
                    if (!push(std::move(b))) { rr.close(); return; }
                    b = std::make_unique<Batch>();
                }
            }
            rr.close();
        }
        if (err == TFR_OK && !b->sessions.empty()) push(std::move(b));      // last batch is short (no drop_remainder)
        std::lock_guard<std::mutex> lk(mu);
        error = err; done = true;
        cv_get.notify_all();
    }
    bool push(std::unique_ptr<Batch> b) {
        int L = 0;
        for (auto& s : b->sessions) L = std::max(L, s.len);
        b->B = (int)b->sessions.size();
        b->T = L - 1;                                  // inputs drop their last element (datasets.py:72-74)
        std::unique_lock<std::mutex> lk(mu);
        cv_put.wait(lk, [&] { return stop || (int)q.size() < prefetch; });
        if (stop) return false;
        q.push_back(std::move(b));
        cv_get.notify_one();
        return true;
    }
    // TFR_OK / TFR_EOF / error
    int next() {
        std::unique_lock<std::mutex> lk(mu);
        cv_get.wait(lk, [&] { return !q.empty() || done; });
        if (!q.empty()) {
            cur = std::move(q.front());
            q.pop_front();
            cv_put.notify_one();
            return TFR_OK;
        }
        cur.reset();
        return error != TFR_OK ? error : TFR_EOF;
    }
    ~SessionReader() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_put.notify_all();
        if (th.joinable()) th.join();
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
    SessionReader* raw = r.release();
    raw->th = std::thread([raw] { raw->produce(); });
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
