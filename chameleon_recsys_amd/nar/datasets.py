"""input_fn of the NAR Estimator: GZIP TFRecord session files -> (features, labels) batches.

Mirror of nar_module/nar/datasets.py:
  * ``prepare_dataset_iterator(files, features_config, batch_size=128, truncate_session_length=20)`` (:166-179)
    returns ``(features, labels)`` - in the reference these are symbolic ``iterator.get_next()`` tensors that take
    a new value at every session.run; here they are dict objects whose numpy arrays are replaced at every
    ``advance()`` of the shared ``SessionDataset`` (the Estimator loop calls it once per step and stops at the
    end of data exactly like TF's OutOfRangeError).
  * parse / truncate / shift / zero-pad semantics of parse_sequence_example (:35-82) and padded_batch (:134-135)
    live in the C++ codec (csrc/host/tfrecord.cpp), which also prefetches the next batch on its own thread (:142).
"""
import ctypes

import numpy as np

from .. import _tfrecord
from .._tfrecord import check
from .utils import get_tf_dtype


class OutOfRangeError(Exception):
    """End of the input files (tf.errors.OutOfRangeError)."""


class BatchDict(dict):
    """The ``next_element`` handle: a dict of numpy arrays re-bound at every step; ``.dataset`` is the iterator."""
    dataset = None


class SessionDataset:
    def __init__(self, path, features_config, batch_size=128, truncate_sequence_length=20, check_crc=True, prefetch=2):
        self.lib = _tfrecord.load()
        files = [path] if isinstance(path, (str, bytes)) else list(path)
        if not files:
            raise ValueError("no input files")
        self.files = [f.decode() if isinstance(f, bytes) else str(f) for f in files]
        single, seq = features_config['single_features'], features_config['sequence_features']
        self.ctx_names, self.seq_names = list(single.keys()), list(seq.keys())
        self.ctx_dtypes = [get_tf_dtype(single[n]['dtype']) for n in self.ctx_names]
        self.seq_dtypes = [get_tf_dtype(seq[n]['dtype']) for n in self.seq_names]
        for req in ('session_size',):
            if req not in single:
                raise ValueError("features_config['single_features'] must contain %r" % req)
        for req in ('item_clicked', 'event_timestamp'):               # nar_model.py:22-23
            if req not in seq:
                raise ValueError("features_config['sequence_features'] must contain %r" % req)
        names = self.ctx_names + self.seq_names
        c_files = (ctypes.c_char_p * len(self.files))(*[f.encode() for f in self.files])
        c_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        c_dtypes = (ctypes.c_int32 * len(names))(*(self.ctx_dtypes + self.seq_dtypes))
        err = ctypes.c_int(0)
        self.h = self.lib.cham_sessions_open(c_files, len(self.files), c_names, c_dtypes, len(self.ctx_names), len(self.seq_names),
                                             int(batch_size), int(truncate_sequence_length), int(bool(check_crc)), int(prefetch),
                                             ctypes.byref(err))
        if not self.h:
            check(err.value or -22, "cham_sessions_open")
        self.features, self.labels = BatchDict(), BatchDict()
        self.features.dataset = self.labels.dataset = self
        self.batch_size = batch_size
        self._peeked = None          # (features, labels) of the batch after the current one, decoded ahead by peek()

    # ---- iteration
    def advance(self):
        """Loads the next batch into ``self.features`` / ``self.labels``; returns False at the end of the data."""
        nxt, self._peeked = (self._peeked if self._peeked is not None else self._load()), None
        if nxt is None:
            return False
        for k, v in nxt[0].items():
            self.features[k] = v
        for k, v in nxt[1].items():
            self.labels[k] = v
        return True

    def peek(self):
        """The batch the next advance() will install, decoded now (the same array objects), or None at the end of the data.
        Lets the training loop upload it and draw its negatives while the current step is still running."""
        if self._peeked is None:
            self._peeked = self._load()
        return self._peeked

    def _load(self):
        if self.h is None:
            return None
        B, T = ctypes.c_int(0), ctypes.c_int(0)
        rc = self.lib.cham_sessions_next(self.h, ctypes.byref(B), ctypes.byref(T))
        if rc == _tfrecord.EOF:
            self.close()
            return None
        check(rc, "cham_sessions_next")
        B, T = B.value, T.value
        f, l = {}, {}
        for i, (n, dt) in enumerate(zip(self.ctx_names, self.ctx_dtypes)):
            if dt == _tfrecord.DT_BYTES:
                tot = check(self.lib.cham_sessions_ctx_bytes(self.h, i, None, None), "cham_sessions_ctx_bytes")
                blob = ctypes.create_string_buffer(max(1, tot)); off = np.zeros(B + 1, np.int64)
                self.lib.cham_sessions_ctx_bytes(self.h, i, blob, off.ctypes.data)
                raw = blob.raw
                f[n] = np.array([raw[off[b]:off[b + 1]] for b in range(B)], dtype=object)
            else:
                a = np.empty(B, np.int64 if dt == _tfrecord.DT_INT64 else np.float32)
                check(self.lib.cham_sessions_ctx(self.h, i, a.ctypes.data), "cham_sessions_ctx")
                f[n] = a
        for i, (n, dt) in enumerate(zip(self.seq_names, self.seq_dtypes)):
            a = np.empty((B, T), np.int64 if dt == _tfrecord.DT_INT64 else np.float32)
            check(self.lib.cham_sessions_seq(self.h, i, a.ctypes.data), "cham_sessions_seq")
            f[n] = a
        nxt, last = np.empty((B, T), np.int64), np.empty((B, 1), np.int64)
        check(self.lib.cham_sessions_labels(self.h, nxt.ctypes.data, last.ctypes.data), "cham_sessions_labels")
        l['label_next_item'], l['label_last_item'] = nxt, last
        return f, l

    def get_next(self):
        if not self.advance():
            raise OutOfRangeError()
        return self.features, self.labels

    def __iter__(self):
        return self

    def __next__(self):
        if not self.advance():
            raise StopIteration
        return dict(self.features), dict(self.labels)

    def close(self):
        if self.h is not None:
            h, self.h = self.h, None
            self.lib.cham_sessions_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_dataset(path, features_config, batch_size=128, num_map_threads=None, truncate_sequence_length=20):
    """datasets.py:100-143 (num_map_threads is accepted for signature compatibility: one decode thread sustains
    > 100k sessions/s, the GPU step consumes ~10k)."""
    return SessionDataset(path, features_config, batch_size=batch_size, truncate_sequence_length=truncate_sequence_length)


def prepare_dataset_iterator(files, features_config, batch_size=128, truncate_session_length=20):
    """datasets.py:166-179: returns the (features, labels) ``next_element`` pair."""
    ds = make_dataset(files, features_config, batch_size=batch_size, truncate_sequence_length=truncate_session_length)
    return ds.features, ds.labels
