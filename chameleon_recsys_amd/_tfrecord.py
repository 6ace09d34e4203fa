"""ctypes binding of libchameleon_tfrecord.so (include/chameleon_tfrecord.h) - the host-side session-file codec."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int32, c_int64, c_uint32, c_uint64, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHAM_TFRECORD_LIB: another build of the same codec (the ASAN / UBSAN one of build.build_host(sanitize=True), tests only)
LIB_PATH = os.environ.get("CHAM_TFRECORD_LIB") or os.path.join(_HERE, "libchameleon_tfrecord.so")

DT_INT64, DT_FLOAT, DT_BYTES = 0, 1, 2
OK, EOF = 0, 1
ERRORS = {-22: "bad argument", -5: "I/O error (missing / truncated file?)", -74: "TFRecord CRC mismatch",
          -71: "malformed SequenceExample protobuf", -61: "a configured feature is missing or has the wrong type / arity"}

_SIGNATURES = {
    "cham_crc32c_masked": (c_uint32, [c_void_p, c_uint64]),
    "cham_sessions_open": (c_void_p, [POINTER(c_char_p), c_int, POINTER(c_char_p), POINTER(c_int32), c_int, c_int, c_int, c_int,
                                      c_int, c_int, POINTER(c_int)]),
    "cham_sessions_next": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "cham_sessions_ctx": (c_int, [c_void_p, c_int, c_void_p]),
    "cham_sessions_ctx_bytes": (c_int64, [c_void_p, c_int, c_void_p, c_void_p]),
    "cham_sessions_seq": (c_int, [c_void_p, c_int, c_void_p]),
    "cham_sessions_labels": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cham_sessions_close": (None, [c_void_p]),
    "cham_tfr_open": (c_void_p, [c_char_p]),
    "cham_tfr_next": (c_int, [c_void_p, POINTER(POINTER(c_uint8)), POINTER(c_uint64), c_int]),
    "cham_tfr_close": (None, [c_void_p]),
    "cham_tfw_open": (c_void_p, [c_char_p, c_int]),
    "cham_tfw_write_record": (c_int, [c_void_p, c_void_p, c_uint64]),
    "cham_tfw_write_session": (c_int, [c_void_p, POINTER(c_char_p), POINTER(c_int32), c_int, c_int, c_void_p, c_void_p,
                                       POINTER(c_char_p), c_void_p, c_void_p, c_int]),
    "cham_tfw_close": (c_int, [c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


class TFRecordError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TFRecordError("host codec %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc < 0:
        raise TFRecordError("%s: %s (code %d)" % (what, ERRORS.get(rc, "error"), rc))
    return rc
