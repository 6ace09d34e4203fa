"""Philox4x32-10 counter-based generator (Salmon et al., SC'11), numpy-vectorised.

TEST INFRASTRUCTURE (see oracle/__init__.py).

TF's ``tf.random_shuffle`` streams (graph seed 42, op-id derived op seeds; nar_model.py:1229,
1241, 1300, nar_trainer_gcom.py:343) cannot be reproduced without TF, so "bit-exact negative
sample indices" is defined between this oracle and the HIP sampler.  Both draw one 32-bit word

    rand32(q, j, b, stage; seed, step) = Philox4x32-10(ctr=(q, j, b, stage), key=(seed, step))[0]

* ``q``     element index inside the list being shuffled,
* ``j``     click column (0 for the two batch-level shuffles),
* ``b``     GLOBAL session row (so data-parallel shards draw the same numbers),
* ``stage`` 0 = recent-buffer shuffle, 1 = candidate-pool shuffle, 2 = per-click shuffle,
* ``seed``  tf_random_seed (42), ``step`` the global step (0-based, one per batch).

A "shuffle" of a list is defined as the stable ascending sort of the 64-bit keys
``(rand32 << 32) | q`` - distinct by construction, i.e. a uniformly random permutation.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

STAGE_BUFFER = 0
STAGE_POOL = 1
STAGE_CLICK = 2


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable unsigned ints < 2**32; returns the 4 output words (uint64 arrays
    holding 32-bit values)."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0           # 32x32 -> 64 bit, fits uint64
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def rand32(q, j, b, stage, seed, step):
    return philox4x32_10(q, j, b, stage, seed, step)[0]


def sort_keys(q, j, b, stage, seed, step):
    """64-bit shuffle keys ``(rand32 << 32) | q``."""
    q = np.asarray(q, dtype=np.uint64)
    return (rand32(q, j, b, stage, seed, step) << np.uint64(32)) | (q & MASK32)
