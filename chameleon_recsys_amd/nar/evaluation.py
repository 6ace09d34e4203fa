"""Evaluation helpers of the NAR hook: nar_module/nar/evaluation.py of the reference for the metrics this build carries
(HitRate, MRR, HitRateBySessionPosition) + the item cold-start analysis state (:50-90).  The novelty / diversity / coverage metrics
of the reference (metrics.py:172-778) are out of scope."""
import numpy as np

from .metrics import HitRateBySessionPosition


def update_metrics(preds, labels, labels_norm_pop, preds_norm_pop, clicked_items, streaming_metrics, recommender=''):
    """evaluation.py:12-26."""
    for metric in streaming_metrics:
        if metric.name == HitRateBySessionPosition.name:
            metric.add(preds, labels, labels_norm_pop)
        else:
            metric.add(preds, labels)


def compute_metrics_results(streaming_metrics, recommender=''):
    """evaluation.py:28-46 (same result keys)."""
    results = {}
    for metric in streaming_metrics:
        if metric.name == HitRateBySessionPosition.name:
            recall_by_session_pos, avg_norm_pop_by_session_pos, hitrate_total_by_session_pos = metric.result()
            for key in recall_by_session_pos:
                results['{}_{}_{:02d}'.format(metric.name, recommender, key)] = recall_by_session_pos[key]
                if recommender == 'chameleon':
                    results['{}_{}_{:02d}'.format('clicks_at_pos', recommender, key)] = hitrate_total_by_session_pos[key]
                    results['{}_{}_{:02d}'.format('avg_norm_pop_by_pos', recommender, key)] = avg_norm_pop_by_session_pos[key]
        else:
            results['{}_{}'.format(metric.name, recommender)] = metric.result()
    return results


class ColdStartAnalysisState:
    """evaluation.py:50-90: for every item, the number of steps between its first click and its first appearance in a top-n
    recommendation list."""

    def __init__(self):
        self.items_num_steps_before_first_rec = dict()
        self.unique_clicked_items_count = 0

    def update_items_num_steps_before_first_rec(self, batch_rec_items, items_first_click_step, step):
        ids = np.unique(np.asarray(batch_rec_items).reshape(-1))
        self.unique_clicked_items_count = len(items_first_click_step)
        for item_id in ids[ids != 0].tolist():
            if item_id in items_first_click_step and item_id not in self.items_num_steps_before_first_rec:
                elapsed_steps = step - items_first_click_step[item_id]
                assert elapsed_steps >= 0
                self.items_num_steps_before_first_rec[item_id] = elapsed_steps

    def get_statistics(self):
        if len(self.items_num_steps_before_first_rec) > 0:
            values = np.array(list(self.items_num_steps_before_first_rec.values()))
            return {'min': np.min(values), '01%': np.percentile(values, 1), '10%': np.percentile(values, 10),
                    '25%': np.percentile(values, 25), '50%': np.percentile(values, 50), '75%': np.percentile(values, 75),
                    '90%': np.percentile(values, 90), '99%': np.percentile(values, 99), 'max': np.max(values),
                    'mean': np.mean(values), 'std': np.std(values), 'uniqueRecommendedItemsCount': len(values),
                    'uniqueClickedItemsCount': self.unique_clicked_items_count}
        return {'uniqueClickedItemsCount': 0}
