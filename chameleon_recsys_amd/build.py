"""Builds libchameleon_nar.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libchameleon_nar.so")
SOURCES = ["gemm.hip", "gemm_x3.hip", "gemm_p3.hip", "gemm_h2.hip", "dm_fused.hip", "gemm_b16.hip", "sampler.hip", "features.hip", "scorer.hip", "rnn.hip", "rnn_coop.hip", "optim.hip", "state.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("CHAM_BUILD_DEFINES", "").split()          # extra hipcc flags (development aid)


HOST_LIB = os.path.join(HERE, "libchameleon_tfrecord.so")
HOST_SOURCES = [os.path.join(CSRC, "host", "tfrecord.cpp")]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread"]


def _stamp(files=None, flags=FLAGS):
    h = hashlib.sha256()
    if files is None:
        files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if os.path.isfile(os.path.join(CSRC, f))]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode()); h.update(fh.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


HOST_LIB_SANITIZED = os.path.join(HERE, "libchameleon_tfrecord_san.so")
SANITIZE_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def build_host(force=False, verbose=True, sanitize=False):
    """libchameleon_tfrecord.so: the host-side session TFRecord codec (plain C++17 + zlib, g++).
    sanitize=True: the same sources as libchameleon_tfrecord_san.so under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; test
    infrastructure - tests/test_tfrecord_fuzz.py loads it through CHAM_TFRECORD_LIB in a child process with libasan preloaded)."""
    lib = HOST_LIB_SANITIZED if sanitize else HOST_LIB
    flags = [f for f in HOST_FLAGS if f != "-O2"] + SANITIZE_FLAGS if sanitize else HOST_FLAGS
    stamp_file = lib + ".stamp"
    stamp = _stamp(HOST_SOURCES, flags)
    if not force and os.path.exists(lib) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return lib
    cmd = [os.environ.get("CXX", "g++")] + flags + HOST_SOURCES + ["-o", lib, "-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return lib


def sanitizer_runtime():
    """Path of libasan.so for LD_PRELOAD (python itself is not instrumented), or None."""
    try:
        out = subprocess.check_output([os.environ.get("CXX", "g++"), "-print-file-name=libasan.so"]).decode().strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return os.path.realpath(out) if os.path.sep in out and os.path.exists(out) else None


def _src_stamp(src):
    """Hash of one translation unit: the source, the shared header(s) it includes and the flags."""
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, src)] + sorted(os.path.join(CSRC, x) for x in os.listdir(CSRC) if x.endswith(".h")):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode()); h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compiles the sources whose hash changed (one hipcc process per file, in parallel) and links libchameleon_nar.so."""
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        st, sf = _src_stamp(s), o + ".stamp"
        if not force and os.path.exists(o) and os.path.exists(sf) and open(sf).read() == st:
            continue
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, sf, st, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, sf, st, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
        if verbose and out.strip():
            print(out.decode())
        with open(sf, "w") as fh:
            fh.write(st)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_host(force="--force" in sys.argv)
