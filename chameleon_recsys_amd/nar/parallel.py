"""Data-parallel NAR training: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference is single-worker by design (README.md:252).  The step shards by SESSION ROWS:
every rank sees the global batch's integer tensors (ids, timestamps, sizes - a few hundred KB), so the candidate
pool (nar_model.py:1286-1300), max_event_timestamp (:235), the normalisation statistics and the loss denominator
sum(mask) (:664) are computed redundantly and identically; each rank runs sampling / features / CAR / RNN / scorer
for its own rows only (negative sampling is keyed by the GLOBAL row index, so the result does not depend on the
sharding), and the flat gradient buffer is summed with ONE all-reduce per step (dense grads ~12 MB + the small
embedding tables; large catalogs would switch the tables to a sparse row exchange).  The state update after the
step is applied identically on every rank from the replicated ids - no communication.
"""
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
    gradients are all-reduced (SUM - the local loss is already divided by the GLOBAL sum(mask))."""

    def __init__(self, model, process_group=None):
        self.model = model
        self.pg = process_group
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        rt = model.rt
        rt.dp_rank, rt.dp_world = self.rank, self.world
        if self.world > 1:
            rt.dp_allreduce = self._allreduce
            # identical initial weights on every rank
            dist.broadcast(rt.flat, src=0, group=self.pg)

    def _allreduce(self, flat_grads):
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.pg)

    def upload(self, global_features, global_labels):
        n = np.asarray(global_features['item_clicked']).shape[0]
        b, e = shard_rows(n, self.rank, self.world)
        f, l = slice_batch(global_features, global_labels, b, e)
        # sequence tensors keep the GLOBAL padded length T (every rank must agree on T for the sampler keys)
        return self.model.upload_batch(f, l, global_features, global_labels, row_begin=b)

    def global_loss(self):
        """[total, xe, reg]: xe summed over ranks (each rank holds its rows' share / global sum(mask))."""
        loss = self.model.total_loss.clone()
        if self.world > 1:
            xe = loss[1:2].clone()
            dist.all_reduce(xe, op=dist.ReduceOp.SUM, group=self.pg)
            loss[1] = xe[0]
            loss[0] = xe[0] + loss[2]
        return loss
