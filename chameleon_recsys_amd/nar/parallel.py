"""Data-parallel NAR training: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference is single-worker by design (README.md:252).  The step shards by SESSION ROWS:
every rank sees the global batch's integer tensors (ids, timestamps, sizes - a few hundred KB), so the candidate
pool (nar_model.py:1286-1300), max_event_timestamp (:235), the normalisation statistics and the loss denominator
sum(mask) (:664) are computed redundantly and identically; each rank runs sampling / features / CAR / RNN / scorer
for its own rows only (negative sampling is keyed by the GLOBAL row index, so the result does not depend on the
sharding), and the flat gradient buffer is summed with ONE all-reduce per step (dense grads ~12 MB + the small
embedding tables; CHAM_DP_MODE=sharded reduce-scatters the whole buffer to owner ranks instead, =sparse / sparse_rs exchange only
the touched rows of the item table - see DataParallelNAR).  The state update after the step is applied identically on every rank from the
replicated ids - no communication.
"""
import logging

import numpy as np
import torch
import torch.distributed as dist


def shard_rows(n_rows, rank, world):
    """Contiguous row range of `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def slice_batch(features, labels, begin, end):
    f = {k: np.asarray(v)[begin:end] for k, v in features.items()}
    l = {k: np.asarray(v)[begin:end] for k, v in labels.items()}
    return f, l


class DataParallelNAR:
    """Wraps a NARModuleModel: ``upload(global_features, global_labels)`` -> this rank's device batch;
    gradients are summed over ranks (the local loss is already divided by the GLOBAL sum(mask)).

    mode "allreduce" (default): ONE all-reduce of the flat gradient buffer, every rank runs the full Adam.
    mode "sharded" (env CHAM_DP_MODE=sharded; the large-catalog layout): reduce-scatter of the flat gradients, each rank runs
    TF-Adam on its contiguous 1/world slice of (weights, m, v) - the embedding tables are >90 % of it - and the updated
    slices are all-gathered.  Same result bit for bit (Adam is elementwise); optimizer HBM traffic per rank / world.
    (Rounds 1-3 also had "hybrid" = all-reduce of the dense gradients + a DENSE reduce-scatter of the embedding-table region to owner
    ranks: at config 5 that is the whole 7.5 GB table per step where "sparse_rs" moves the touched rows - removed in round 4, SURVEY.md
    8e's C1 + C2 is "sparse_rs".)
    mode "sparse" (large catalogs, BASELINE config 5): the trainable ITEM table's data gradient only touches the rows of the
    global batch's clicked ids + the candidate pool + the pad item - a list every rank derives identically from the replicated
    integers, so no index exchange is needed: every rank packs those rows (duplicates allowed) into a compact [L, dim] buffer,
    ONE all-reduce of that buffer replaces the all-reduce of the whole [n_items, dim] table (5 M x 378 floats = 7.5 GB ->
    ~86 k rows = 130 MB), and the summed rows are written back; the L2 term of the other rows is local (added inside the Adam
    kernel).  Everything else in the flat buffer is all-reduced densely.
    mode "sparse_rs": as "sparse", with the touched rows exchanged as SURVEY.md 8e C2 words it - a REDUCE-SCATTER of the packed row
    buffer (rank r receives the sums of the r-th 1/world of the list: the rows it "owns" this step) followed by an all-gather of the
    summed chunks, the dense remainder in its own all-reduce.  Adam stays replicated: TF's dense Adam moves every row of the table
    every step (L2 term + momentum, nar_model.py:740/917), so an owner-only update would have to ship the 7.5 GB table back - the
    summed gradient rows are what is worth moving.  On a ring the two forms move the same bytes; kept as the literal C2 collective
    pair and as the RCCL reduce_scatter_tensor / all_gather_into_tensor coverage at the row granularity."""

    def __init__(self, model, process_group=None, mode=None):
        import os
        self.model = model
        self.pg = process_group
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.mode = mode or os.environ.get("CHAM_DP_MODE", "allreduce")
        if self.mode not in ("allreduce", "sharded", "sparse", "sparse_rs"):
            raise ValueError("CHAM_DP_MODE must be 'allreduce', 'sharded', 'sparse' or 'sparse_rs'")
        rt = model.rt
        rt.dp_rank, rt.dp_world = self.rank, self.world
        rt.dp_pg = process_group
        rt.dp_mode = self.mode
        self._early, self._early_work, self._early_bytes = None, None, 0
        # dtype of the dense gradient exchange ("12-13 MB fp32 / 6.5 MB bf16"): CHAM_DP_GRAD_DTYPE = f32 | bf16 | auto
        # (default: bf16 exactly when the runtime computes in bf16 - BASELINE configs[2]).  bf16: the flat gradients are rounded to bf16
        # (nearest even) into a communication buffer, summed by the collective in bf16 and widened back into the fp32 buffer Adam reads -
        # identical values on every rank.  The touched embedding rows of the sparse modes stay fp32 (a few MB of sparse sums).
        want = os.environ.get("CHAM_DP_GRAD_DTYPE", "auto")
        if want not in ("auto", "f32", "bf16"):
            raise ValueError("CHAM_DP_GRAD_DTYPE must be 'auto', 'f32' or 'bf16'")
        if want == "bf16" and self.mode == "sharded":
            raise ValueError("CHAM_DP_MODE=sharded exchanges fp32 (the owner rank's Adam reads the reduce-scatter's output): CHAM_DP_GRAD_DTYPE=bf16 "
                             "is for the allreduce / sparse / sparse_rs modes")
        self.comm_bf16 = self.mode != "sharded" and (want == "bf16" or (want == "auto" and getattr(rt, 'gemm_dtype', 'f32') == 'bf16'))
        self._comm16, self._early16 = None, None
        if self.comm_bf16 and self.world > 1 and self.rank == 0:
            logging.getLogger(__name__).info("data-parallel dense gradients are exchanged and summed in bf16 (CHAM_DP_GRAD_DTYPE=%s, "
                                             "gemm_dtype=%s); CHAM_DP_GRAD_DTYPE=f32 keeps the exchange in fp32", want,
                                             getattr(rt, 'gemm_dtype', 'f32'))
        self.last_exchange_bytes = 0       # payload this rank handed to the collectives of the last step (all modes; bookkeeping only)
        # how often each collective ran (tests assert the reduce-scatter / all-gather branches really executed with world > 1;
        # bench.py's "dp" object reports them)
        self.collective_calls = {"all_reduce": 0, "reduce_scatter": 0, "all_gather": 0, "broadcast": 0}
        # time_exchange = True: HIP events on the step's stream around the synchronous part of the exchange (everything but the early
        # bucket's overlapped flight) - the collectives make the current stream wait for RCCL's, so the pair brackets them; read with
        # exchange_ms() (bench.py's "dp" object).  Off by default: no events, no synchronisation.
        self.time_exchange = False
        self._exchange_events = []
        # CHAM_DP_FORCE=1: install the exchange hooks for a process group of ONE rank too - every collective of every mode then runs
        # (on RCCL when the group's backend is "nccl") and must leave the step bit-identical to the plain single-process one
        # (tests/test_dp_rccl_gpu.py; bench.py's dp_self_exchange_ms)
        self.active = self.world > 1 or (os.environ.get("CHAM_DP_FORCE", "0") == "1" and dist.is_initialized())
        rt.dp_active = self.active
        if self.active:
            L = getattr(rt, 'layout', None)
            # Early bucket: the session-FC and scorer kernels [Wf1 .. Ws4] are contiguous in the flat buffer and their gradients
            # are final long before the tail of the backward pass (PreCAR backward beside the W2 weight gradient): their
            # all-reduce is issued from the side lane as soon as they are written and overlaps with the rest of the backward
            # (allreduce / sparse modes; matters for strong scaling, where a 32-row step is ~2 ms).  CHAM_DP_EARLY_BUCKET=0: off.
            ents = getattr(L, 'entries', None) or {}
            if self.mode in ("allreduce", "sparse", "sparse_rs") and os.environ.get("CHAM_DP_EARLY_BUCKET", "1") == "1" and 'Wf1' in ents and 'Ws4' in ents:
                a, b = L.entries['Wf1'].offset, L.entries['Ws4'].offset + L.entries['Ws4'].size
                names = [e.name for e in L.entries.values() if a <= e.offset < b]
                if names == ['Wf1', 'Wf2', 'Ws1', 'Ws2', 'Ws3', 'Ws4']:
                    self._early = (a, b)
                    rt.dp_early_bucket = self._issue_early_bucket
            rt.dp_gather_slots = self._gather_slots
            self._via_host = False
            if self.mode in ("sharded", "sparse_rs"):
                self._probe_tensor_collectives(rt.flat.device)
            if self.mode == "sharded":
                if rt.flat.numel() % self.world:
                    raise ValueError("flat parameter buffer (%d) does not split over %d ranks" % (rt.flat.numel(), self.world))
                rt.dp_sharded = self._sharded_step
                self._grad_slice = torch.empty(rt.flat.numel() // self.world, dtype=rt.flat.dtype, device=rt.flat.device)
            elif self.mode in ("sparse", "sparse_rs") and 'items_embedding' in rt.layout.entries:
                e = rt.layout.entries['items_embedding']
                self._item = (e.offset, e.shape[0], e.shape[1])
                self._compact, self._rs = None, None
                rt.dp_allreduce = self._sparse_allreduce
            else:
                rt.dp_allreduce = self._allreduce
            # identical initial weights on every rank (a write to rt.flat that bypasses load_logical_weights: bump the version the
            # bf16 / plane shadows of the weights are keyed on)
            dist.broadcast(rt.flat, src=0, group=self.pg)
            self.collective_calls["broadcast"] += 1
            rt.weights_version = getattr(rt, 'weights_version', 0) + 1

    # ---- early bucket (see __init__)
    COMM16_CHUNK = 1 << 25          # elements of the bf16 communication buffer (64 MB): larger ranges go through it in pieces

    def _comm_buffer(self, flat_grads, n):
        """bf16 communication buffer of at most COMM16_CHUNK elements, allocated once (round 5 imaged the WHOLE flat gradient
        buffer - 3.8 GB at configs[4] - where a bounded staging buffer does)."""
        need = min(int(n), self.COMM16_CHUNK)
        if self._comm16 is None or self._comm16.numel() < need or self._comm16.device != flat_grads.device:
            self._comm16 = torch.empty(need, dtype=torch.bfloat16, device=flat_grads.device)
        return self._comm16

    def _probe_tensor_collectives(self, device):
        """reduce_scatter_tensor / all_gather_into_tensor on a 4-element tensor of the runtime's device, once, on every rank (a
        collective): RCCL takes device tensors, gloo takes CPU tensors (torch 2.10) - whether a gloo group takes DEVICE tensors (the
        two-ranks-on-one-GPU tests) is found out here rather than assumed.  When it does not, the same two collectives run on host
        staging copies (self._via_host): the chunking / owner-slice arithmetic around them is the same code either way."""
        self._via_host = False
        try:
            x = torch.ones(self.world * 2, dtype=torch.float32, device=device)
            y = torch.empty(2, dtype=torch.float32, device=device)
            dist.reduce_scatter_tensor(y, x, op=dist.ReduceOp.SUM, group=self.pg)
            dist.all_gather_into_tensor(x, y, group=self.pg)
            ok = bool((x.cpu() == float(self.world)).all())
        except (RuntimeError, NotImplementedError, ValueError):
            ok = False
        if not ok:
            if torch.device(device).type == "cpu":
                raise RuntimeError("backend %r of the data-parallel group runs neither reduce_scatter_tensor nor all_gather_into_tensor: "
                                   "CHAM_DP_MODE=%s needs them" % (dist.get_backend(self.pg), self.mode))
            self._via_host = True

    def _reduce_scatter(self, out, inp):
        self.collective_calls["reduce_scatter"] += 1
        if self._via_host:
            h_out, h_in = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.reduce_scatter_tensor(h_out, h_in, op=dist.ReduceOp.SUM, group=self.pg)
            out.copy_(h_out)
        else:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.pg)

    def _all_gather(self, out, inp):
        self.collective_calls["all_gather"] += 1
        if self._via_host:
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h_out, inp.cpu(), group=self.pg)
            out.copy_(h_out)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.pg)

    def _all_reduce(self, t, async_op=False):
        self.collective_calls["all_reduce"] += 1
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)

    def _reduce_range(self, flat_grads, a, b, async_op=False, buf16=None):
        """all-reduce(SUM) of flat_grads[a:b] in the exchange dtype; returns (work or None, bytes handed to the collective).  In bf16 the
        range is rounded (nearest even) into a bf16 buffer, summed there by the collective and widened back into the fp32 buffer Adam
        reads.  With `buf16` (the early bucket's own buffer, async) the CALLER widens with _widen_range once the collective has finished;
        without it the range goes through the bounded staging buffer piece by piece, synchronously, and is widened here."""
        if not self.comm_bf16:
            return self._all_reduce(flat_grads[a:b], async_op=async_op), 4 * (b - a)
        if buf16 is not None:
            buf16.copy_(flat_grads[a:b])           # fp32 -> bf16, round to nearest even
            return self._all_reduce(buf16, async_op=async_op), 2 * (b - a)
        assert not async_op
        stage = self._comm_buffer(flat_grads, b - a)
        for c in range(a, b, self.COMM16_CHUNK):
            d = min(b, c + self.COMM16_CHUNK)
            piece = stage[:d - c]
            piece.copy_(flat_grads[c:d])
            self._all_reduce(piece)
            flat_grads[c:d].copy_(piece)
        return None, 2 * (b - a)

    def _widen_range(self, flat_grads, a, b, buf16):
        if self.comm_bf16:
            flat_grads[a:b].copy_(buf16)

    def _issue_early_bucket(self, flat_grads):
        """Called by the backward pass on the lane that produced the bucket, right after its last gradient was written."""
        a, b = self._early
        if self.comm_bf16 and (self._early16 is None or self._early16.numel() != b - a or self._early16.device != flat_grads.device):
            self._early16 = torch.empty(b - a, dtype=torch.bfloat16, device=flat_grads.device)      # (its own small buffer: the sparse modes never build the whole-buffer image)
        self._early_work, self._early_bytes = self._reduce_range(flat_grads, a, b, async_op=True, buf16=self._early16 if self.comm_bf16 else None)
        self._early_grads = flat_grads

    def _wait_early_bucket(self):
        if self._early_work is not None:
            self._early_work.wait()            # (stream-ordered for RCCL: the current stream waits for the collective)
            self._early_work = None
            self._widen_range(self._early_grads, *self._early, buf16=self._early16)
            return True
        return False

    def _ranges_without_early(self, a, b, early_done):
        """[a, b) minus the early bucket when that one was already reduced."""
        if not early_done:
            return [(a, b)]
        ea, eb = self._early
        out = []
        if a < min(b, ea):
            out.append((a, min(b, ea)))
        if max(a, eb) < b:
            out.append((max(a, eb), b))
        return out

    def _timed(fn):
        def wrapped(self, *a, **k):
            if not self.time_exchange or not torch.cuda.is_available():
                return fn(self, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(self, *a, **k)
            e1.record()
            self._exchange_events.append((e0, e1))
            return out
        return wrapped

    def exchange_ms(self, reset=True):
        """Per-step durations (ms) of the exchange recorded while time_exchange was set (synchronises the device)."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._exchange_events]
        if reset:
            self._exchange_events = []
        return out

    @_timed
    def _allreduce(self, flat_grads):
        early = self._wait_early_bucket()
        self.last_exchange_bytes = self._early_bytes if early else 0
        for a, b in self._ranges_without_early(0, flat_grads.numel(), early):
            _, nbytes = self._reduce_range(flat_grads, a, b)
            self.last_exchange_bytes += nbytes

    # ---- the touched rows of the item table <-> the packed [L, dim] exchange buffer (HIP: cham_rows_gather / cham_rows_scatter on the
    # ---- current stream).  There is no CPU form of the product path: the world-2/3 gloo tests install numpy stand-ins for these two.
    def _rows_gather(self, table, ids, rows):
        from .._lib import check, ptr
        if not table.is_cuda:
            raise RuntimeError("cham_rows_gather is a HIP kernel: the item-table gradient must live on a ROCm device")
        L, dim = rows.shape
        check(self.model.rt.lib.cham_rows_gather(ptr(table), ptr(ids), L, dim, ptr(rows), torch.cuda.current_stream().cuda_stream), "cham_rows_gather")

    def _rows_scatter(self, rows, ids, table):
        from .._lib import check, ptr
        if not table.is_cuda:
            raise RuntimeError("cham_rows_scatter is a HIP kernel: the item-table gradient must live on a ROCm device")
        L, dim = rows.shape
        check(self.model.rt.lib.cham_rows_scatter(ptr(rows), ptr(ids), L, dim, ptr(table), torch.cuda.current_stream().cuda_stream), "cham_rows_scatter")

    @_timed
    def _sparse_allreduce(self, flat_grads):
        """ONE collective for everything but the early bucket: [flat buffer without the item table | touched item-table rows] packed
        into a contiguous communication buffer (round 1 issued three: prefix, suffix, rows)."""
        rt = self.model.rt
        off, n, dim = self._item
        # GLOBAL clicked ids + candidate pool + pad item of this step: int32 row indices in a buffer of the StepPlan, written by
        # forward().  (Round 2 finding: building the list here from temporaries - torch.cat(...).to(int32), freed when this function
        # returns - gave GPU memory-access faults with 8 ranks and the asynchronous state update; nothing is allocated or freed
        # around the collective any more.  profiles/r02_notes.md)
        ids = rt.dp_touched
        L = ids.numel()
        early = self._wait_early_bucket()
        ranges = self._ranges_without_early(0, off, early) + self._ranges_without_early(off + n * dim, flat_grads.numel(), early)
        n_dense = sum(b - a for a, b in ranges)
        need = n_dense + L * dim
        if self._compact is None or self._compact.numel() < need:
            self._compact = torch.empty(need, dtype=flat_grads.dtype, device=flat_grads.device)
        comm = self._compact[:need]
        table = flat_grads[off:off + n * dim]
        o = 0
        for a, b in ranges:
            comm[o:o + b - a].copy_(flat_grads[a:b]); o += b - a
        rows = comm[n_dense:].view(L, dim)
        self._rows_gather(table, ids, rows)
        dense_done = False
        if self.comm_bf16 and n_dense:          # dense remainder in bf16 (its own collective(s)), the touched rows fp32
            self._reduce_range(comm, 0, n_dense)
            dense_done = True
        if self.mode == "sparse_rs":
            if not dense_done and n_dense:
                self._all_reduce(comm[:n_dense])
            # C2: the row list padded to a multiple of the world size (pad rows: zeros, never written back);
            # rank r receives the sums of rows [r * per, (r + 1) * per) - the rows it owns this step - and the summed chunks are
            # all-gathered.  Runs on RCCL and on gloo alike (round 5 substituted an all-reduce on gloo: these lines had only ever
            # executed with a world of one).
            per = -(-L // self.world)
            if self._rs is None or self._rs[0].numel() < per * self.world * dim or self._rs[0].device != comm.device:
                self._rs = (torch.zeros(per * self.world * dim, dtype=comm.dtype, device=comm.device),
                            torch.empty(per * dim, dtype=comm.dtype, device=comm.device))
            packed, mine = self._rs[0][:per * self.world * dim], self._rs[1][:per * dim]
            packed[:L * dim].copy_(rows.view(-1))
            packed[L * dim:].zero_()
            self._reduce_scatter(mine, packed)       # C2: this rank's rows, summed
            self._all_gather(packed, mine)
            rows.view(-1).copy_(packed[:L * dim])
        elif dense_done:
            self._all_reduce(comm[n_dense:])
        else:
            self._all_reduce(comm)
        self.last_exchange_bytes = (2 if dense_done else 4) * n_dense + 4 * L * dim + (self._early_bytes if early else 0)
        self.last_touched_rows = L
        o = 0
        for a, b in ranges:
            flat_grads[a:b].copy_(comm[o:o + b - a]); o += b - a
        # duplicate ids carry identical sums: concurrent writes of the same value
        self._rows_scatter(rows, ids, table)

    # ---- checkpoints: in the sharded mode a rank's Adam slots are only valid on its own slice
    def _gather_slots(self, m, v):
        """Full (m, v) on every rank, for NARRuntime.state_dict(): the owned slices all-gathered (no-op in the replicated modes)."""
        if not self.active or self.mode in ("allreduce", "sparse", "sparse_rs"):
            return m, v
        n = m.numel() // self.world
        a = self.rank * n
        out = []
        for x in (m, v):
            full = x.clone()
            if n > 0:
                self._all_gather(full[:n * self.world], x[a:a + n].clone())
            out.append(full)
        return out[0], out[1]

    @_timed
    def _sharded_step(self, flat_grads, flat_params, adam):
        """reduce-scatter of the flat gradients -> Adam on this rank's contiguous 1/world slice -> all-gather of the updated slices
        (RCCL and gloo alike)."""
        n = flat_params.numel() // self.world
        a = self.rank * n
        self._reduce_scatter(self._grad_slice, flat_grads)
        adam(a, a + n, self._grad_slice, 0)
        self._all_gather(flat_params, flat_params[a:a + n].clone())
        self.last_exchange_bytes = 4 * flat_grads.numel() + 4 * flat_params.numel()

    def upload(self, global_features, global_labels):
        n = np.asarray(global_features['item_clicked']).shape[0]
        b, e = shard_rows(n, self.rank, self.world)
        f, l = slice_batch(global_features, global_labels, b, e)
        # sequence tensors keep the GLOBAL padded length T (every rank must agree on T for the sampler keys)
        return self.model.upload_batch(f, l, global_features, global_labels, row_begin=b)

    def global_loss(self):
        """[total, xe, reg]: xe summed over ranks (each rank holds its rows' share / global sum(mask))."""
        loss = self.model.total_loss.clone()
        if self.active:
            xe = loss[1:2].clone()
            self._all_reduce(xe)
            loss[1] = xe[0]
            loss[0] = xe[0] + loss[2]
        return loss
