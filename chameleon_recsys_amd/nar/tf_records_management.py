"""Session TFRecord WRITER: GZIP TFRecord files of tf.train.SequenceExample, byte-compatible with what
nar_module/nar/tf_records_management.py:12-32 + preprocessing (nar_preprocess_gcom.py:75-108) emit, written by
the C++ codec (libchameleon_tfrecord.so) - TensorFlow is not needed.  Used by the synthetic-data generator."""
import ctypes

import numpy as np

from .. import _tfrecord
from .._tfrecord import check
from .utils import chunks, get_tf_dtype


def _schema(features_config):
    single, seq = features_config['single_features'], features_config['sequence_features']
    names = list(single.keys()) + list(seq.keys())
    dtypes = [get_tf_dtype(single[n]['dtype']) for n in single] + [get_tf_dtype(seq[n]['dtype']) for n in seq]
    c_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    c_dtypes = (ctypes.c_int32 * len(names))(*dtypes)
    return names, dtypes, c_names, c_dtypes, len(single), len(seq)


class SessionTFRecordWriter:
    """with SessionTFRecordWriter(path, features_config) as w: w.write(session_dict)"""

    def __init__(self, export_filename, features_config, gzip_level=6):
        self.lib = _tfrecord.load()
        self.names, self.dtypes, self.c_names, self.c_dtypes, self.n_ctx, self.n_seq = _schema(features_config)
        self.h = self.lib.cham_tfw_open(export_filename.encode(), gzip_level)
        if not self.h:
            raise _tfrecord.TFRecordError("cannot open %s for writing" % export_filename)

    def write(self, session):
        n_ctx, n_seq = self.n_ctx, self.n_seq
        ctx_i64 = np.zeros(n_ctx, np.int64); ctx_f32 = np.zeros(n_ctx, np.float32)
        ctx_bytes = (ctypes.c_char_p * n_ctx)()
        for i in range(n_ctx):
            v = session[self.names[i]]
            if self.dtypes[i] == _tfrecord.DT_INT64:
                ctx_i64[i] = int(v)
            elif self.dtypes[i] == _tfrecord.DT_FLOAT:
                ctx_f32[i] = float(v)
            else:
                ctx_bytes[i] = v if isinstance(v, bytes) else str(v).encode()
        length = len(session[self.names[n_ctx]])
        seq_i64 = np.zeros((n_seq, length), np.int64); seq_f32 = np.zeros((n_seq, length), np.float32)
        for i in range(n_seq):
            v = np.asarray(session[self.names[n_ctx + i]])
            if v.shape[0] != length:
                raise ValueError("sequence feature %s has %d steps, expected %d" % (self.names[n_ctx + i], v.shape[0], length))
            if self.dtypes[n_ctx + i] == _tfrecord.DT_INT64:
                seq_i64[i] = v
            else:
                seq_f32[i] = v
        check(self.lib.cham_tfw_write_session(self.h, self.c_names, self.c_dtypes, n_ctx, n_seq, ctx_i64.ctypes.data,
                                              ctx_f32.ctypes.data, ctx_bytes, seq_i64.ctypes.data, seq_f32.ctypes.data, length),
              "cham_tfw_write_session")

    def close(self):
        if self.h:
            h, self.h = self.h, None
            check(self.lib.cham_tfw_close(h), "cham_tfw_close")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def save_rows_to_tf_record_file(rows, features_config, export_filename):
    """tf_records_management.py:22-32 (rows are session dicts; the schema comes from features_config instead of a
    make_sequence_example_fn building protobuf objects)."""
    with SessionTFRecordWriter(export_filename, features_config) as w:
        for row in rows:
            w.write(row)


def export_sessions_to_tf_records(sessions, features_config, output_path, examples_by_file=1000):
    """tf_records_management.py:34-42 export_dataframe_to_tf_records: '*' in output_path -> 4-digit chunk index."""
    template = output_path.replace('*', '{0:04d}')
    out = []
    for i, chunk in enumerate(chunks(sessions, examples_by_file)):
        save_rows_to_tf_record_file(chunk, features_config, template.format(i))
        out.append(template.format(i))
    return out
