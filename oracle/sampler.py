"""Negative sampler oracle (integer; bit-exact contract with the HIP sampler).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates nar_module/nar/nar_model.py:265-276 and :1220-1304 (TF graph) - the reference's own numpy
clone is nar/benchmarks/candidate_sampling.py:7-90.  Where the two disagree we follow the TF
graph: ``tf.setdiff1d`` keeps the ORDER and REPETITION of the candidate pool
(nar_model.py:1259), while the numpy clone's ``np.setdiff1d(..., assume_unique=True)`` sorts.

Every ``tf.random_shuffle`` is replaced by the keyed stable sort of oracle/philox.py.
"""
import numpy as np

from . import philox

INF_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)


def _shuffled_take(values, keys, limit):
    """``shuffle(values[valid])[:limit]``: valid = key != INF; order = ascending key."""
    valid = keys != INF_KEY
    order = np.argsort(keys, kind="stable")
    order = order[: int(valid.sum())][:limit]
    return values[order]


def sample_from_recent_buffer(buffer_ids, sample_size, seed, step):
    """nar_model.py:1220-1233 get_sample_from_recently_clicked_items_buffer.

    q = position in the (un-compacted) buffer."""
    buffer_ids = np.asarray(buffer_ids, dtype=np.int64)
    q = np.arange(buffer_ids.shape[0], dtype=np.uint64)
    keys = philox.sort_keys(q, 0, 0, philox.STAGE_BUFFER, seed, step)
    keys = np.where(buffer_ids != 0, keys, INF_KEY)
    return _shuffled_take(buffer_ids, keys, sample_size)


def candidate_pool(all_clicked_items, buffer_sample, num_neg, sample_size, seed, step, factor=20):
    """nar_model.py:1281-1300 get_batch_negative_samples (pool part).

    pool = shuffle(concat(batch non-zero ids WITH repetition (:1289-1293), buffer sample))[:20*N].
    q = position in the un-compacted concatenation [all_clicked_items.ravel() ; buffer-sample slots
    0..sample_size-1] (slots beyond the actual sample length are empty)."""
    flat = np.asarray(all_clicked_items, dtype=np.int64).reshape(-1)
    slots = np.zeros(sample_size, dtype=np.int64)
    slots[: len(buffer_sample)] = buffer_sample
    cat = np.concatenate([flat, slots])
    q = np.arange(cat.shape[0], dtype=np.uint64)
    keys = philox.sort_keys(q, 0, 0, philox.STAGE_POOL, seed, step)
    keys = np.where(cat != 0, keys, INF_KEY)
    return _shuffled_take(cat, keys, num_neg * factor)


def canonical_slots(pool):
    """canon[q] = smallest q' with pool[q'] == pool[q]."""
    first = {}
    canon = np.empty(len(pool), dtype=np.int32)
    for q, v in enumerate(pool.tolist()):
        canon[q] = first.setdefault(v, q)
    return canon


def neg_items_click(pool, canon, valid_mask, num_neg, b_global, j, seed, step):
    """nar_model.py:1239-1254 get_neg_items_click.

    shuffle(valid) -> distinct values in order of first occurrence (tf.unique +
    unsorted_segment_min, :1244-1249) -> first N -> zero-pad (:1252).
    First-occurrence order of the distinct values == ascending order of each value's MIN key."""
    P = len(pool)
    q = np.arange(P, dtype=np.uint64)
    keys = philox.sort_keys(q, j, b_global, philox.STAGE_CLICK, seed, step)
    keys = np.where(valid_mask, keys, INF_KEY)
    minkey = np.full(P, INF_KEY, dtype=np.uint64)
    np.minimum.at(minkey, canon, keys)
    order = np.argsort(minkey, kind="stable")
    n_distinct = int((minkey != INF_KEY).sum())
    order = order[:n_distinct][:num_neg]
    win_q = (minkey[order] & philox.MASK32).astype(np.int64)   # pool position that won
    ids = np.zeros(num_neg, dtype=np.int64)
    slots = np.full(num_neg, -1, dtype=np.int32)
    ids[: len(order)] = pool[win_q]
    slots[: len(order)] = canon[win_q]
    return ids, slots


def batch_negative_samples(all_clicked_items, buffer_ids, num_neg, sample_size, seed, step,
                           rows=None, return_aux=False):
    """Full sampler: nar_model.py:265-276 -> [B, T, N] int64 (last position dropped, :275).

    ``all_clicked_items`` is the GLOBAL batch [B, T+1]; ``rows`` optionally restricts the output to a
    slice of session rows (data-parallel shard), RNG keyed by the global row index."""
    aci = np.asarray(all_clicked_items, dtype=np.int64)
    B, T1 = aci.shape
    buf_sample = sample_from_recent_buffer(buffer_ids, sample_size, seed, step)
    pool = candidate_pool(aci, buf_sample, num_neg, sample_size, seed, step)
    canon = canonical_slots(pool)
    rows = range(B) if rows is None else rows
    out = np.zeros((len(rows), T1 - 1, num_neg), dtype=np.int64)
    out_slots = np.full((len(rows), T1 - 1, num_neg), -1, dtype=np.int32)
    for i, b in enumerate(rows):
        # nar_model.py:1257-1259: ordered setdiff of the pool against this session's ids
        valid = ~np.isin(pool, aci[b])
        for j in range(T1 - 1):
            if aci[b, j] == 0:          # :1262-1263 padded click -> zeros
                continue
            ids, sl = neg_items_click(pool, canon, valid, num_neg, b, j, seed, step)
            out[i, j], out_slots[i, j] = ids, sl
    if return_aux:
        return out, dict(pool=pool, canon=canon, slots=out_slots, buf_sample=buf_sample)
    return out
