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


_depth = 0          # ranges this module has open (single host thread drives a step)


def push(name):
    """Opens a nested range; returns its depth (>= 0) or -1 when tracing is off."""
    global _depth
    if _lib is None:
        return -1
    _depth += 1
    return _lib.roctxRangePushA(name.encode())


def pop():
    global _depth
    if _lib is None or _depth == 0:
        return -1
    _depth -= 1
    return _lib.roctxRangePop()


def depth():
    return _depth


def unwind(to_depth):
    """Closes every range opened since depth() returned `to_depth` - the `finally` of a function that pushes / pops stage ranges across many
    calls that may raise (NARModuleModel.forward: an error mid-step must not leave the range stack unbalanced for the rest of the process)."""
    while _depth > to_depth and _lib is not None:
        pop()


class range_:
    """`with _roctx.range_("CAR forward"): ...` - costs one attribute test when tracing is off."""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        push(self.name)

    def __exit__(self, *a):
        pop()
        return False
