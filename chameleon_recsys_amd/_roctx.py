"""roctx ranges around the stages of a step ("rocprofv3 counters + roctx ranges around K0-K7").  Off by default: with
CHAM_ROCTX=1 the ranges are emitted through the ROCm tools extension library (librocprofiler-sdk-roctx.so / libroctx64.so) and show up in a
`rocprofv3 --marker-trace --kernel-trace` run as host-side spans (which stage ENQUEUED a kernel; the kernels themselves are stream-ordered
and may execute later).  Never a dependency of the product path: without the variable, or without the library, every call is a no-op."""
import ctypes
import os

_lib = None
_on = os.environ.get("CHAM_ROCTX", "0") == "1"
if _on:
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        for d in ("", "/opt/rocm/lib/"):
            try:
                _lib = ctypes.CDLL(d + name)
                break
            except OSError:
                _lib = None
        if _lib is not None:
            break
    if _lib is not None:
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePop.restype = ctypes.c_int


def enabled():
    return _lib is not None


def push(name):
    """Opens a nested range; returns its depth (>= 0) or -1 when tracing is off."""
    return _lib.roctxRangePushA(name.encode()) if _lib is not None else -1


def pop():
    return _lib.roctxRangePop() if _lib is not None else -1


class range_:
    """`with _roctx.range_("CAR forward"): ...` - costs one attribute test when tracing is off."""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _lib is not None:
            _lib.roctxRangePushA(self.name.encode())

    def __exit__(self, *a):
        if _lib is not None:
            _lib.roctxRangePop()
        return False
