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


class HitRateBySessionPosition(StreamingMetric):
    """HitRate@N per session position (nar_module/nar/metrics.py:136-168): key = 1-based click column; also the average recent
    normalised popularity of the labels at that position and the number of clicks counted there."""
    name = 'hitrate_at_n_by_pos'

    def reset(self):
        self.hitrate_matches_by_session_pos = {}
        self.hitrate_total_by_session_pos = {}
        self.norm_pop_by_pos = {}

    def add(self, predictions, labels, labels_norm_pop):
        rank, valid = _first_hit_rank(predictions, labels, self.topn)
        labels_norm_pop = np.asarray(labels_norm_pop, dtype=np.float64)
        for col in np.flatnonzero(valid.any(axis=0)):
            v = valid[:, col]
            key = int(col) + 1
            self.hitrate_total_by_session_pos[key] = self.hitrate_total_by_session_pos.get(key, 0) + int(v.sum())
            self.norm_pop_by_pos[key] = self.norm_pop_by_pos.get(key, 0) + float(labels_norm_pop[:, col][v].sum())
            hits = int(((rank[:, col] >= 0) & v).sum())
            if hits:
                self.hitrate_matches_by_session_pos[key] = self.hitrate_matches_by_session_pos.get(key, 0) + hits

    def result(self):
        tot = self.hitrate_total_by_session_pos
        hitrate = {k: self.hitrate_matches_by_session_pos.get(k, 0) / float(tot[k]) for k in tot}
        avg_pop = {k: self.norm_pop_by_pos.get(k, 0) / float(tot[k]) for k in tot}
        return hitrate, avg_pop, tot
