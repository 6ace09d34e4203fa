"""Recent-clicks state oracle (host-side, integer + float64).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates nar_module/nar/clicked_items_state.py:187-250 (buffer + recent popularity; the
co-occurrence / cold-start parts are baseline-only and out of scope) and the hook's
batch -> (ids, timestamps) flattening, nar_module/nar/nar_model.py:1635-1646.
Pinned against the reference class itself: tests/golden/state_trace.npz (oracle/make_golden.py).
"""
import numpy as np


def batch_clicks_for_state(item_clicked, label_last_item, event_timestamp):
    """nar_model.py:1635-1646.  Returns (ids_nonzero, ts_nonzero), row-major order."""
    items = np.concatenate([item_clicked, label_last_item], axis=1)
    flat = items.reshape(-1)
    nz = np.nonzero(flat)
    last_ts = np.max(event_timestamp, axis=1).reshape(-1, 1)          # :1642
    ts = np.concatenate([event_timestamp, last_ts], axis=1).reshape(-1)
    return flat[nz], ts[nz]


class ClickedItemsStateOracle:
    def __init__(self, recent_clicks_buffer_hours, recent_clicks_buffer_max_size,
                 recent_clicks_for_normalization, num_items):
        self.hours = recent_clicks_buffer_hours
        self.max_size = recent_clicks_buffer_max_size
        self.for_norm = recent_clicks_for_normalization
        self.num_items = num_items
        self.articles_pop = np.zeros(num_items, dtype=np.int64)
        self.buffer = np.zeros((self.max_size, 2), dtype=np.int64)     # (id, ts), newest first
        self._update_pop_norm(np.zeros(num_items, dtype=np.int64))

    def _update_pop_norm(self, recent_pop):
        # clicked_items_state.py:242-246
        self.articles_recent_pop = recent_pop
        self.articles_recent_pop_norm = np.maximum(recent_pop / (recent_pop.sum() + 1),
                                                   [1.0 / self.for_norm])

    def update_items_state(self, ids, ts):
        # clicked_items_state.py:206-223
        batch = np.hstack([ids.reshape(-1, 1), ts.reshape(-1, 1)])[::-1]
        thr = np.min(ts) - int(self.hours * 1000 * 60 * 60)            # :225-228
        kept = self.buffer[self.buffer[:, 1] >= thr]
        buf = np.vstack([batch, kept])[: self.max_size]
        if buf.shape[0] < self.max_size:
            buf = np.vstack([buf, np.zeros((self.max_size - buf.shape[0], 2), dtype=np.int64)])
        self.buffer = buf
        # :231-240
        b_ids = buf[:, 0]
        self._update_pop_norm(np.bincount(b_ids[b_ids != 0], minlength=self.num_items).astype(np.int64))
        # :248-250
        self.articles_pop += np.bincount(ids, minlength=self.num_items)

    def get_recent_clicks_buffer(self):
        return self.buffer[:, 0]

    def get_articles_recent_pop_norm(self):
        return self.articles_recent_pop_norm
