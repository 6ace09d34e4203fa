"""Recent-clicks state (host side): recent-clicks ring buffer + recent popularity -> ``articles_recent_pop_norm``.

Mirrors nar_module/nar/clicked_items_state.py:10-250 for the parts the NAR training step consumes
(update_items_state :187-193, buffer :206-228, recent pop :231-246, global pop :248-250, snapshot around eval
:49-79) and the item cold-start bookkeeping (:43-45, 97-104, 196-203).  The co-occurrence matrix (:252-255, benchmarks only) is
out of scope.
Vectorised numpy (np.bincount instead of collections.Counter); results are bit-identical to the reference class
(tests/golden/state_trace.npz).
"""
import numpy as np

MILISECS_BY_HOUR = 1000 * 60 * 60



_LANE_STREAMS = {}


def lane_stream(device, role, priority=0):
    """The process-wide HIP stream of a lane (side / aux / upload / state) on a device.  Every runtime and state object of a process
    shares them: torch hands out pool streams round-robin and HIP multiplexes streams onto a few hardware queues, so per-object streams
    made WHICH lanes share a queue - and with it how well they overlap - depend on how many objects the process had created before
    (profiles/r03_notes.md section 7).  Work of different objects on one lane is merely stream-ordered."""
    import torch
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), role, int(priority))
    st = _LANE_STREAMS.get(key)
    if st is None:
        st = _LANE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=int(priority))
    return st


class _ColdStartBookkeeping:
    """Item cold-start analysis (--eval_cold_start): the step an item was clicked first and, through evaluation.ColdStartAnalysisState,
    how many steps later it first showed up in a top-n recommendation list (clicked_items_state.py:43-45, 97-104, 196-203; snapshot /
    restore :57-59, 75-79).  Host-side dictionaries in both state classes - it only runs when the flag is set."""

    def _reset_cold_start(self):
        from .evaluation import ColdStartAnalysisState
        self.current_step = 0
        self.items_first_click_step = dict()
        self.cold_start_state = ColdStartAnalysisState()

    def _save_cold_start(self):
        from copy import deepcopy
        self._cold_chkp = (deepcopy(self.items_first_click_step), deepcopy(self.cold_start_state), self.current_step)

    def _restore_cold_start(self):
        self.items_first_click_step, self.cold_start_state, self.current_step = self._cold_chkp
        del self._cold_chkp

    def increment_current_step(self):
        self.current_step += 1

    def get_current_step(self):
        return self.current_step

    def get_cold_start_state(self):
        return self.cold_start_state

    def update_items_first_click_step(self, batch_clicked_items):
        step = self.get_current_step()
        for item_id in set(int(x) for x in batch_clicked_items) - {0}:
            if item_id not in self.items_first_click_step:
                self.items_first_click_step[item_id] = step


class ClickedItemsState(_ColdStartBookkeeping):

    def __init__(self, recent_clicks_buffer_hours, recent_clicks_buffer_max_size, recent_clicks_for_normalization, num_items):
        self.recent_clicks_buffer_hours = recent_clicks_buffer_hours
        self.recent_clicks_buffer_max_size = recent_clicks_buffer_max_size
        self.recent_clicks_for_normalization = recent_clicks_for_normalization
        self.num_items = num_items
        self.reset_state()

    def reset_state(self):
        self.articles_pop = np.zeros(shape=[self.num_items], dtype=np.int64)
        self.articles_recent_pop = np.zeros(shape=[self.num_items], dtype=np.int64)
        self._update_recent_pop_norm(self.articles_recent_pop)
        # two columns (article_id, click_timestamp), newest first
        self.pop_recent_clicks_buffer = np.zeros(shape=[self.recent_clicks_buffer_max_size, 2], dtype=np.int64)
        self._reset_cold_start()

    def save_state_checkpoint(self):
        self.articles_pop_chkp = np.copy(self.articles_pop)
        self.pop_recent_clicks_buffer_chkp = np.copy(self.pop_recent_clicks_buffer)
        self.articles_recent_pop_chkp = np.copy(self.articles_recent_pop)
        self._save_cold_start()

    def restore_state_checkpoint(self):
        self.articles_pop = self.articles_pop_chkp
        self.pop_recent_clicks_buffer = self.pop_recent_clicks_buffer_chkp
        # NB the reference does not restore articles_recent_pop(_norm) (clicked_items_state.py:61-79): it is recomputed
        # from the restored buffer at the next update.  We keep that behaviour.
        self._restore_cold_start()
        del self.articles_pop_chkp, self.pop_recent_clicks_buffer_chkp, self.articles_recent_pop_chkp

    def get_articles_pop(self):
        return self.articles_pop

    def get_articles_recent_pop(self):
        return self.articles_recent_pop

    def get_articles_recent_pop_norm(self):
        return self.articles_recent_pop_norm

    def get_recent_clicks_buffer(self):
        return self.pop_recent_clicks_buffer[:, 0]

    def update_items_state(self, batch_clicked_items, batch_clicked_timestamps):
        self._update_recently_clicked_items_buffer(batch_clicked_items, batch_clicked_timestamps)
        self._update_recent_pop_items()
        self._update_pop_items(batch_clicked_items)

    def _update_recently_clicked_items_buffer(self, batch_clicked_items, batch_clicked_timestamps):
        batch = np.hstack([batch_clicked_items.reshape(-1, 1), batch_clicked_timestamps.reshape(-1, 1)])[::-1]
        self.truncate_last_hours_recent_clicks_buffer(np.min(batch_clicked_timestamps))
        buf = np.vstack([batch, self.pop_recent_clicks_buffer])[:self.recent_clicks_buffer_max_size]
        if buf.shape[0] < self.recent_clicks_buffer_max_size:
            buf = np.vstack([buf, np.zeros(shape=[self.recent_clicks_buffer_max_size - buf.shape[0], 2], dtype=np.int64)])
        self.pop_recent_clicks_buffer = buf

    def truncate_last_hours_recent_clicks_buffer(self, reference_timestamp):
        thr = reference_timestamp - int(self.recent_clicks_buffer_hours * MILISECS_BY_HOUR)
        self.pop_recent_clicks_buffer = self.pop_recent_clicks_buffer[self.pop_recent_clicks_buffer[:, 1] >= thr]

    def _update_recent_pop_items(self):
        ids = self.pop_recent_clicks_buffer[:, 0]
        self.articles_recent_pop = np.bincount(ids[ids != 0], minlength=self.num_items).astype(np.int64)
        self._update_recent_pop_norm(self.articles_recent_pop)

    def _update_recent_pop_norm(self, articles_recent_pop):
        min_norm_pop = 1.0 / self.recent_clicks_for_normalization
        self.articles_recent_pop_norm = np.maximum(articles_recent_pop / (articles_recent_pop.sum() + 1), [min_norm_pop])

    def _update_pop_items(self, batch_items_nonzero):
        self.articles_pop += np.bincount(batch_items_nonzero, minlength=self.num_items)


def batch_clicks_for_state(clicked_items, last_item_label, clicked_timestamps):
    """ItemsStateUpdaterHook.after_run, nar_model.py:1635-1646: flatten [clicks ; last label], drop padding; the last
    label re-uses the max timestamp of its session."""
    batch_clicked_items = np.concatenate([clicked_items, last_item_label], axis=1).reshape(-1)
    nz = np.nonzero(batch_clicked_items)
    last_ts = np.max(clicked_timestamps, axis=1).reshape(-1, 1)
    ts = np.concatenate([clicked_timestamps, last_ts], axis=1).reshape(-1)
    return batch_clicked_items[nz], ts[nz]


class DeviceClickedItemsState(_ColdStartBookkeeping):
    """The same state kept in HBM: recent-clicks ring buffer (ids, timestamps), recent popularity histogram,
    ``articles_recent_pop_norm`` (float32, what the graph is fed) and global popularity, updated by the HIP kernels of
    csrc/state.hip straight from the batch tensors the step already has on the device - no host round trip per step.
    Same public methods as ClickedItemsState (the getters download); bit-identical results (tests/test_state_gpu.py).

    The update only depends on the batch's ids / timestamps, not on the training step, so it runs on its own stream as soon as
    the step has finished READING the state (NARModuleModel calls note_consumed() after its last state read; the next
    feed_state waits for `updated_event`): ~0.25 ms of small kernels leave the critical path.  CHAM_STATE_ASYNC=0 keeps the
    update on the caller's stream."""
    is_device = True

    def __init__(self, recent_clicks_buffer_hours, recent_clicks_buffer_max_size, recent_clicks_for_normalization, num_items,
                 device='cuda:0'):
        import torch
        from .. import _lib
        self.lib = _lib.load()
        self.torch = torch
        self.device = torch.device(device)
        self.recent_clicks_buffer_hours = recent_clicks_buffer_hours
        self.recent_clicks_buffer_max_size = recent_clicks_buffer_max_size
        self.recent_clicks_for_normalization = recent_clicks_for_normalization
        self.num_items = num_items
        self._ws = None
        import os
        self.stream = lane_stream(self.device, "state") if os.environ.get("CHAM_STATE_ASYNC", "1") == "1" else None
        self.updated_event, self.consumed_event, self._consumed_key = None, None, None
        self.reset_state()

    def reset_state(self):
        t, dev, n, m = self.torch, self.device, self.num_items, self.recent_clicks_buffer_max_size
        self.buf_ids = t.zeros(m, dtype=t.int64, device=dev)
        self.buf_ts = t.zeros(m, dtype=t.int64, device=dev)
        self.recent_pop = t.zeros(n, dtype=t.int32, device=dev)
        self.articles_pop = t.zeros(n, dtype=t.int64, device=dev)
        # _update_recent_pop_norm(zeros): max(0 / 1, 1 / for_norm)
        self.pop_norm = t.full((n,), float(np.float32(1.0 / self.recent_clicks_for_normalization)), dtype=t.float32, device=dev)
        self.n_valid = t.zeros(2, dtype=t.int32, device=dev)      # {rows retained in the buffer, clicks counted into recent_pop}
        self.n_updates = 0
        self._reset_cold_start()
        self._after_host_side_change()

    # ---- stream ordering
    def _after_host_side_change(self):
        """State tensors were (re)created on the caller's stream: later updates on the state stream must come after that."""
        if self.stream is not None:
            self.stream.wait_stream(self.torch.cuda.current_stream())
            self.consumed_event = None

    def sync_to_current(self):
        """Every reader of the state tensors on another stream calls this first."""
        if self.updated_event is not None:
            self.torch.cuda.current_stream().wait_event(self.updated_event)

    def note_consumed(self, aci):
        """Called by the step after its last read of the state; `aci` identifies the batch whose update may now start."""
        if self.stream is not None:
            ev = self.torch.cuda.Event()
            ev.record()
            self.consumed_event, self._consumed_key = ev, aci.data_ptr()

    # ---- updates
    def update_from_device_batch(self, aci, event_ts):
        """aci [B, T+1] int64 = concat(item_clicked, label_last_item), event_ts [B, T] int64 - device tensors of the GLOBAL batch."""
        from .._lib import check, ptr
        t = self.torch
        B, T1 = aci.shape
        need = self.lib.cham_state_workspace_bytes(B, self.recent_clicks_buffer_max_size)
        if self._ws is None or self._ws.numel() < need:
            self._ws = t.empty(need, dtype=t.uint8, device=self.device)
        st = t.cuda.current_stream()
        if self.stream is not None:
            if self.consumed_event is not None and self._consumed_key == aci.data_ptr():
                self.stream.wait_event(self.consumed_event)        # the step that uploaded this batch is done reading the state
            else:
                self.stream.wait_stream(st)                        # unknown producer of the inputs: fully ordered
            self.consumed_event = None
            st = self.stream
            aci.record_stream(st); event_ts.record_stream(st)     # (caching allocator: these are read on the state stream)
        with t.cuda.stream(st):
            check(self.lib.cham_state_update(ptr(aci), ptr(event_ts), B, T1 - 1, float(self.recent_clicks_buffer_hours),
                                             ptr(self.buf_ids), ptr(self.buf_ts), self.recent_clicks_buffer_max_size,
                                             ptr(self.recent_pop), ptr(self.pop_norm), ptr(self.articles_pop), self.num_items,
                                             self.recent_clicks_for_normalization, ptr(self.n_valid), ptr(self._ws), self._ws.numel(),
                                             st.cuda_stream), "cham_state_update")
            if self.stream is not None:
                self.updated_event = t.cuda.Event()
                self.updated_event.record()
        self.n_updates += 1

    def update_items_state(self, batch_clicked_items, batch_clicked_timestamps):
        """Host-array entry point of the reference class (flattened non-zero ids / timestamps)."""
        t = self.torch
        ids = np.ascontiguousarray(batch_clicked_items, dtype=np.int64).reshape(-1)
        ts = np.ascontiguousarray(batch_clicked_timestamps, dtype=np.int64).reshape(-1, 1)
        aci = np.stack([ids, np.zeros_like(ids)], axis=1)          # one "session" per click: [id, no label]
        self.update_from_device_batch(t.from_numpy(aci).to(self.device), t.from_numpy(ts).to(self.device))

    # ---- the reference's getters (download)
    def get_recent_clicks_buffer(self):
        self.sync_to_current()
        return self.buf_ids.cpu().numpy()

    def get_articles_recent_pop_norm(self):
        self.sync_to_current()
        return self.pop_norm.cpu().numpy()

    def get_articles_recent_pop(self):
        self.sync_to_current()
        return self.recent_pop.cpu().numpy().astype(np.int64)

    def get_articles_pop(self):
        self.sync_to_current()
        return self.articles_pop.cpu().numpy()

    @property
    def pop_recent_clicks_buffer(self):
        self.sync_to_current()
        return np.stack([self.buf_ids.cpu().numpy(), self.buf_ts.cpu().numpy()], axis=1)

    # ---- snapshot around evaluation (clicked_items_state.py:49-79: buffer + global pop; pop_norm is NOT restored)
    def save_state_checkpoint(self):
        self.sync_to_current()
        self._chkp = (self.articles_pop.clone(), self.buf_ids.clone(), self.buf_ts.clone(), self.n_valid.clone(), self.n_updates)
        self._save_cold_start()

    def restore_state_checkpoint(self):
        self.sync_to_current()          # the update still in flight writes the tensors that are being replaced
        self.articles_pop, self.buf_ids, self.buf_ts, self.n_valid, self.n_updates = self._chkp
        del self._chkp
        self._restore_cold_start()
        self._after_host_side_change()
