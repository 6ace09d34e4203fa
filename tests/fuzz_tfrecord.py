"""Mutation fuzzer of the host-side session codec (csrc/host/tfrecord.cpp) - test infrastructure, not a pytest module.

The codec parses UNTRUSTED bytes: a GZIP stream, TFRecord framing (length | masked crc32c | payload | masked crc32c) and a hand-rolled
tf.train.SequenceExample wire decode (the reference hands the same job to tf.data: /root/reference/nar_module/nar/datasets.py:35-82, 124).
Contract checked here: EVERY input ends in a clean end of data or in a _tfrecord.TFRecordError carrying one of the documented codes - never
another exception, a crash, a hang, or (in the sanitized build) an AddressSanitizer / UndefinedBehaviorSanitizer report.

Three mutation layers, each applied to valid files written by the codec's own writer:
  gzip     the compressed bytes are flipped / truncated / extended
  framing  the decompressed stream is mutated (lengths, CRCs, payload), re-compressed; read with and without CRC checking
  proto    one record's payload is mutated and RE-FRAMED WITH CORRECT CRCs, so the protobuf decoder itself sees the damage
           (byte flips, truncation, varint bombs, oversized length prefixes, wrong wire types, spliced payloads)

Run directly (`python tests/fuzz_tfrecord.py --cases 3000 --seed 1`) or through tests/test_tfrecord_fuzz.py, which also runs it against the
ASAN + UBSAN build in a child process (CHAM_TFRECORD_LIB + LD_PRELOAD=libasan)."""
import argparse
import gzip
import os
import struct
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from chameleon_recsys_amd import _tfrecord                                              # noqa: E402
from chameleon_recsys_amd.nar import config, datasets, synthetic                        # noqa: E402
from chameleon_recsys_amd.nar import tf_records_management as tfm                       # noqa: E402


def masked_crc(data):
    """TFRecord's masked CRC-32C through the library's own export (a pure function of the bytes)."""
    lib = _tfrecord.load()
    buf = bytes(data)
    return lib.cham_crc32c_masked(buf, len(buf))


def frame(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + bytes(payload) + struct.pack("<I", masked_crc(payload))


def split_records(raw):
    out, off = [], 0
    while off + 12 <= len(raw):
        (n,) = struct.unpack_from("<Q", raw, off)
        out.append(raw[off + 12: off + 12 + n])
        off += 12 + n + 4
    return out


def seed_corpus(tmp_dir):
    """Two valid session files (G1 schema, Adressa schema with a bytes context feature) -> [(config, decompressed stream)]."""
    corpus = []
    cfg = config.get_session_features_config_gcom(300)
    ss = synthetic.make_sessions(9, 12, 300, cfg, seed=11, hour_index=0, length_dist='g1')
    path = os.path.join(tmp_dir, "seed_g1.tfrecord.gz")
    tfm.save_rows_to_tf_record_file(ss, cfg, path)
    corpus.append((cfg, gzip.open(path, "rb").read()))
    cfg2 = config.get_session_features_config_adressa(200)
    cfg2['single_features']['user_id'] = {'type': 'categorical', 'dtype': 'bytes'}
    ss2 = synthetic.make_sessions(6, 10, 200, cfg2, seed=12, hour_index=0, length_dist='g1')
    for i, s in enumerate(ss2):
        s['user_id'] = ("user-%d" % i).encode()
    path2 = os.path.join(tmp_dir, "seed_adressa.tfrecord.gz")
    tfm.save_rows_to_tf_record_file(ss2, cfg2, path2)
    corpus.append((cfg2, gzip.open(path2, "rb").read()))
    return corpus


# ---- mutators: bytes -> bytes -------------------------------------------------------------------------------------------------------
_VARINT_BOMBS = [b"\xff" * 10 + b"\x01", b"\xff" * 11, b"\x80" * 12 + b"\x00", b"\xff\xff\xff\xff\x0f", b"\xff\xff\xff\xff\xff\xff\xff\xff\x7f",
                 b"\x80\x80\x80\x80\x80\x80\x80\x80\x80\x01"]


def mutate_bytes(rng, data):
    data = bytearray(data)
    if not data:
        return bytes(rng.integers(0, 256, int(rng.integers(0, 16)), dtype=np.uint8))
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 9))
        pos = int(rng.integers(0, len(data))) if data else 0
        if kind == 0 and data:                                     # flip one bit
            data[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1 and data:                                   # overwrite a byte
            data[pos] = int(rng.integers(0, 256))
        elif kind == 2:                                            # truncate
            del data[pos:]
        elif kind == 3:                                            # drop a span
            del data[pos:pos + int(rng.integers(1, 9))]
        elif kind == 4:                                            # insert random bytes
            data[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        elif kind == 5:                                            # insert a varint bomb
            data[pos:pos] = _VARINT_BOMBS[int(rng.integers(0, len(_VARINT_BOMBS)))]
        elif kind == 6 and data:                                   # a length-like byte becomes huge / zero
            data[pos] = (0xFF, 0x7F, 0x80, 0x00)[int(rng.integers(0, 4))]
        elif kind == 7 and len(data) > 8:                          # duplicate a span somewhere else
            a = int(rng.integers(0, len(data) - 4)); n = int(rng.integers(1, min(64, len(data) - a)))
            data[pos:pos] = data[a:a + n]
        elif kind == 8 and len(data) >= 8:                         # overwrite 8 bytes with an extreme little-endian integer
            v = (0, 1, 2 ** 31, 2 ** 32 - 1, 2 ** 63 - 1, 2 ** 64 - 1, 2 ** 40)[int(rng.integers(0, 7))]
            p8 = int(rng.integers(0, len(data) - 7))
            data[p8:p8 + 8] = struct.pack("<Q", v)
    return bytes(data)


def make_case(rng, raw, layer):
    """-> (gzip file bytes, check_crc)"""
    if layer == "gzip":
        comp = gzip.compress(raw, compresslevel=1)
        return mutate_bytes(rng, comp), True
    if layer == "framing":
        return gzip.compress(mutate_bytes(rng, raw), compresslevel=1), bool(rng.integers(0, 2))
    recs = split_records(raw)
    i = int(rng.integers(0, len(recs)))
    how = int(rng.integers(0, 4))
    if how == 0 and len(recs) > 1:                 # splice: the head of one payload, the tail of another
        j = int(rng.integers(0, len(recs)))
        cut = int(rng.integers(0, min(len(recs[i]), len(recs[j])) + 1))
        recs[i] = recs[i][:cut] + recs[j][cut:]
    elif how == 1:                                 # pure noise of a plausible size
        recs[i] = bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8))
    else:
        recs[i] = mutate_bytes(rng, recs[i])
    return gzip.compress(b"".join(frame(r) for r in recs), compresslevel=1), True


def run_case(cfg, file_bytes, check_crc, path, batch_size, trunc):
    """-> 'ok' | 'error:<code>'; raises on anything outside the contract."""
    with open(path, "wb") as fh:
        fh.write(file_bytes)
    try:
        ds = datasets.SessionDataset(path, cfg, batch_size=batch_size, truncate_sequence_length=trunc, check_crc=check_crc)
        try:
            n = 0
            while ds.advance():
                n += 1
                f, l = ds.features, ds.labels
                B, T = f['item_clicked'].shape
                assert l['label_next_item'].shape == (B, T) and l['label_last_item'].shape == (B, 1)
                assert 0 < B <= batch_size and 0 <= T <= trunc - 1, (B, T)
                if n > 10000:
                    raise AssertionError("reader does not terminate")
        finally:
            ds.close()
    except _tfrecord.TFRecordError as ex:
        msg = str(ex)
        code = int(msg[msg.rindex("code") + 4:].strip(" )"))
        if code not in _tfrecord.ERRORS:
            raise AssertionError("undocumented error code %d: %s" % (code, msg))
        return "error:%d" % code
    return "ok"


def run(cases, seed, verbose=False):
    rng = np.random.default_rng(seed)
    outcomes = {}
    with tempfile.TemporaryDirectory() as tmp:
        corpus = seed_corpus(tmp)
        path = os.path.join(tmp, "case.tfrecord.gz")
        for cfg, raw in corpus:        # the unmutated seeds decode cleanly
            assert run_case(cfg, gzip.compress(raw), True, path, 4, 8) == "ok"
        for i in range(cases):
            cfg, raw = corpus[int(rng.integers(0, len(corpus)))]
            layer = ("gzip", "framing", "proto", "proto")[int(rng.integers(0, 4))]
            data, crc = make_case(rng, raw, layer)
            res = run_case(cfg, data, crc, path, int(rng.integers(1, 9)), int(rng.integers(2, 12)))
            key = "%s/%s" % (layer, res)
            outcomes[key] = outcomes.get(key, 0) + 1
    if verbose:
        for k in sorted(outcomes):
            print("%-24s %d" % (k, outcomes[k]))
    return outcomes


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    out = run(a.cases, a.seed, verbose=True)
    errs = sum(v for k, v in out.items() if "error" in k)
    print("fuzz ok: %d cases, %d ended in a TFRecordError, %d decoded, library %s" % (a.cases, errs, a.cases - errs, _tfrecord.LIB_PATH))
