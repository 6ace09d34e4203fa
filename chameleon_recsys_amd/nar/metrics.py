"""Streaming accuracy metrics of the NAR evaluation path: HitRate@N and MRR@N.

Same classes / method names / results as nar_module/nar/metrics.py:23-66 (StreamingMetric, MRR) and :109-134
(HitRate) of the reference, vectorised over the [B, T, K] prediction tensor instead of the reference's Python
double loop.  ``predictions[b, t]`` = item ids ranked by predicted probability (nar_model.py:777-794), ``labels[b, t]``
= next clicked item (0 = padding, skipped).  Pinned against the reference classes: tests/golden/metrics_hitrate_mrr.npz.
"""
import numpy as np


class StreamingMetric:
    name = 'undefined'

    def __init__(self, topn):
        self.topn = topn
        self.reset()

    def reset(self):
        pass

    def add(self, predictions, labels):
        pass

    def result(self):
        pass


def _first_hit_rank(predictions, labels, topn):
    """rank (0-based) of the first position within the top-n where prediction == label, -1 if none; valid = label != 0."""
    predictions = np.asarray(predictions)[..., :topn]
    labels = np.asarray(labels)
    hit = predictions == labels[..., None]
    any_hit = hit.any(axis=-1)
    rank = np.where(any_hit, hit.argmax(axis=-1), -1)
    return rank, labels != 0


class HitRate(StreamingMetric):
    name = 'hitrate_at_n'

    def reset(self):
        self.hitrate_total = 0
        self.hitrate_matches = 0

    def add(self, predictions, labels):
        rank, valid = _first_hit_rank(predictions, labels, self.topn)
        self.hitrate_total += int(valid.sum())
        self.hitrate_matches += int(((rank >= 0) & valid).sum())

    def result(self):
        return self.hitrate_matches / float(self.hitrate_total)


class MRR(StreamingMetric):
    name = 'mrr_at_n'

    def reset(self):
        self.mrr_results = []

    def add(self, predictions, labels):
        rank, valid = _first_hit_rank(predictions, labels, self.topn)
        rr = np.where(rank >= 0, 1.0 / (1.0 + np.maximum(rank, 0)), 0.0)
        self.mrr_results.extend(rr[valid].tolist())       # row-major == the reference's loop order

    def result(self):
        return np.mean(self.mrr_results)
