"""Data-parallel semantics on CPU with the gloo backend, world_size 2: row sharding, sampler keyed by the
GLOBAL row, local losses divided by the GLOBAL sum(mask), gradients combined by ONE all-reduce(SUM) through
chameleon_recsys_amd.nar.parallel - checked against the single-process oracle on the full batch.  (The compute of each
shard is the CPU oracle here: the HIP step itself has no CPU path; its sharded launch parameters - row_begin, global
denominators - are covered by tests/test_step_gpu.py::test_row_shards_sum_to_full_batch on the GPU.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chameleon_recsys_amd.nar import parallel, synthetic
from tests import helpers as H


def test_shard_rows_partition():
    for n in (1, 7, 64, 255, 256):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import sampler as S
    from oracle.nar_oracle import NAROracle
    p = H.tiny_params(C=32, H=24, neg=5, batch_size=16, buffer_size=300, for_norm=50, neg_from_buffer=30, n_items=200, ace_dim=8)
    batches = synthetic.make_batches(3, 16, 8, 200, p['session_features_config'], seed=2, length_dist='g1')
    st = H.warm_state(p, batches[:2])
    f, l = batches[2]
    buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
    orc = NAROracle(p, seed=5)
    # ---- this rank's rows; negatives keyed by the global row index from the replicated global id tensor
    b, e = parallel.shard_rows(16, rank, world)
    fl, ll = parallel.slice_batch(f, l, b, e)
    aci = np.concatenate([f['item_clicked'], l['label_last_item']], 1)
    neg = S.batch_negative_samples(aci, buf, 5, 30, 42, 0, rows=range(b, e))
    out = orc.forward(fl, ll, buf, pop, 'train', neg_items=neg, global_max_ts=int(np.asarray(f['event_timestamp']).max()))
    # the oracle's forward normalises by the LOCAL sum(mask) and max timestamp: re-scale to the global denominator and check
    # the global scalars are what every rank would compute from the replicated ints
    g_mask = float(np.minimum(np.maximum(f['session_size'] - 1, 0), f['item_clicked'].shape[1]).sum())
    l_mask = float(out['mask'].sum())
    assert int(np.asarray(fl['event_timestamp']).max()) <= int(np.asarray(f['event_timestamp']).max())
    xe_share = out['xe_loss'] * (l_mask / g_mask)
    for v in orc.w.values():
        v.grad = None
    xe_share.backward()
    names = list(orc.w.keys())
    flat = torch.cat([(orc.w[k].grad if orc.w[k].grad is not None else torch.zeros_like(orc.w[k])).reshape(-1) for k in names])

    class _RT:          # the two attributes DataParallelNAR touches
        flat = torch.zeros(4)
        dp_rank = dp_world = dp_allreduce = None

    class _Model:
        rt = _RT()
        total_loss = torch.tensor([float(xe_share.detach()) + float(out['reg_loss'].detach()), float(xe_share.detach()), float(out['reg_loss'].detach())])
    dp = parallel.DataParallelNAR(_Model())
    assert (dp.rank, dp.world) == (rank, world) and _Model.rt.dp_allreduce is not None
    _Model.rt.dp_allreduce(flat)                       # ONE all-reduce(SUM) of the flat gradient buffer
    loss = dp.global_loss()
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), flat=flat.numpy(), loss=loss.numpy(), neg=neg)
    dist.destroy_process_group()


def test_two_rank_gradients_equal_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(str(tmp_path / "dp.npz"))
    torch.set_num_threads(1)
    from oracle.nar_oracle import NAROracle
    p = H.tiny_params(C=32, H=24, neg=5, batch_size=16, buffer_size=300, for_norm=50, neg_from_buffer=30, n_items=200, ace_dim=8)
    batches = synthetic.make_batches(3, 16, 8, 200, p['session_features_config'], seed=2, length_dist='g1')
    st = H.warm_state(p, batches[:2])
    f, l = batches[2]
    orc = NAROracle(p, seed=5)
    out = orc.forward(f, l, st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm(), 'train')
    assert np.array_equal(out['neg_items'].numpy()[:8], got['neg'])           # rank 0's rows of the global sample
    out['xe_loss'].backward()
    flat = torch.cat([(v.grad if v.grad is not None else torch.zeros_like(v)).reshape(-1) for v in orc.w.values()]).numpy()
    scale = np.abs(flat).max()
    assert np.abs(got['flat'] - flat).max() < 2e-5 * scale + 1e-7
    assert abs(got['loss'][1] - float(out['xe_loss'])) < 1e-5 and abs(got['loss'][0] - float(out['total_loss'])) < 1e-5


# ---- optimizer-step exchange modes: same parameters on every rank as the single-process step ---------------------------------------
# ---- Every collective of nar/parallel.py runs here with world 2 AND 3 on gloo - all_reduce, and the reduce_scatter_tensor /
# ---- all_gather_into_tensor pair of "sharded" and "sparse_rs" (SURVEY.md 8e C2).  The reference is single-worker (README.md:252): this
# ---- test is the specification.  The touched-row list of the sparse modes has duplicates and a length that no world size divides (the
# ---- zero-padded tail of the reduce-scatter); cham_rows_gather / cham_rows_scatter (HIP) are replaced by numpy stand-ins.
_N_TOTAL, _N_EMB = 4098, 1500            # flat parameter count (2 * 3 * 683: splits over 2 and 3 ranks), item table [0, _N_EMB)
_ITEM_DIM = 6                            # item table = [250, 6]
_TOUCHED = np.asarray([0, 7, 7, 19, 3, 249, 100, 101, 7, 55, 200, 19, 0], np.int32)      # 13 entries: not a multiple of 2 or 3


def _toy_adam(flat, m, lr=0.1):
    def adam(a, b, grads, g_off):         # elementwise update on the parameter range [a, b): what cham_adam_tf does per element
        g = grads[g_off:g_off + (b - a)]
        m[a:b] = 0.9 * m[a:b] + 0.1 * g
        flat[a:b] -= lr * m[a:b] / (g.abs() + 1.0)
    return adam


def _toy_grads(step, rank):
    """A rank's flat gradient: dense everywhere, the item table non-zero only in the touched rows (as the data gradient is)."""
    g = torch.randn(_N_TOTAL, generator=torch.Generator().manual_seed(100 + 10 * step + rank))
    table = g[:_N_EMB].view(-1, _ITEM_DIM)
    keep = torch.zeros(table.shape[0], dtype=torch.bool)
    keep[torch.from_numpy(_TOUCHED).long()] = True
    table[~keep] = 0.0
    return g


def _mode_worker(rank, world, port, out_dir, mode, grad_dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), CHAM_DP_GRAD_DTYPE=grad_dtype)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g0 = torch.Generator().manual_seed(0)

    class _E:
        def __init__(self, name, offset, size, shape=None):
            self.name, self.offset, self.size, self.shape = name, offset, size, shape or (size,)

    class _Layout:
        emb_end = _N_EMB
        # the session-FC / scorer kernels are contiguous (the early all-reduce bucket of parallel.DataParallelNAR)
        entries = {e.name: e for e in (_E('items_embedding', 0, _N_EMB, (_N_EMB // _ITEM_DIM, _ITEM_DIM)), _E('W2', _N_EMB, 500),
                                       _E('Wf1', 2000, 300), _E('Wf2', 2300, 200), _E('Ws1', 2500, 100), _E('Ws2', 2600, 60),
                                       _E('Ws3', 2660, 40), _E('Ws4', 2700, 20), _E('b1', 2720, _N_TOTAL - 2720))}

    class _RT:
        flat = torch.randn(_N_TOTAL, generator=g0) if rank == 0 else torch.zeros(_N_TOTAL)     # broadcast must replicate rank 0
        layout = _Layout()
        dp_rank = dp_world = dp_allreduce = dp_sharded = dp_early_bucket = dp_gather_slots = None
        dp_touched = torch.from_numpy(_TOUCHED)

    class _Model:
        rt = _RT()
    dp = parallel.DataParallelNAR(_Model(), mode=mode)
    # numpy stand-ins for the two HIP row kernels (the product methods refuse CPU tensors)
    with pytest.raises(RuntimeError, match="HIP kernel"):
        dp._rows_gather(torch.zeros(4), torch.zeros(1, dtype=torch.int32), torch.zeros(1, 4))

    def rows_gather(table, ids, rows):
        rows.numpy()[...] = table.view(-1, rows.shape[1]).numpy()[ids.numpy()]

    def rows_scatter(rows, ids, table):
        table.view(-1, rows.shape[1]).numpy()[ids.numpy()] = rows.numpy()
    dp._rows_gather, dp._rows_scatter = rows_gather, rows_scatter
    rt = _Model.rt
    m = torch.zeros(_N_TOTAL)
    adam = _toy_adam(rt.flat, m)
    summed = []
    for step in range(3):
        grads = _toy_grads(step, rank)
        if rt.dp_sharded is not None:
            rt.dp_sharded(grads, rt.flat, adam)
        else:
            assert rt.dp_early_bucket is not None and dp._early == (2000, 2720)
            if step != 1:                                  # (step 1: no early issue - the whole buffer goes in one collective)
                rt.dp_early_bucket(grads)                  # what the backward pass does once [Wf1 .. Ws4] are final
            rt.dp_allreduce(grads)
            summed.append(grads.numpy().copy())
            adam(0, _N_TOTAL, grads, 0)
    m_full, _ = rt.dp_gather_slots(m, m.clone())               # checkpoint path: full Adam slots on every rank
    np.savez(os.path.join(out_dir, "%s_rank%d.npz" % (mode, rank)), flat=rt.flat.numpy(), m=m.numpy(), m_full=m_full.numpy(),
             summed=np.asarray(summed), calls=np.asarray([dp.collective_calls[k] for k in ("all_reduce", "reduce_scatter", "all_gather")]),
             nbytes=dp.last_exchange_bytes, comm_bf16=dp.comm_bf16)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode,grad_dtype", [("allreduce", "f32"), ("sharded", "f32"), ("sparse", "f32"), ("sparse_rs", "f32"),
                                             ("sparse", "bf16"), ("sparse_rs", "bf16")])
def test_exchange_modes_match_single_process_step(tmp_path, mode, grad_dtype, world):
    mp.spawn(_mode_worker, args=(world, _free_port(), str(tmp_path), mode, grad_dtype), nprocs=world, join=True)
    got = [np.load(str(tmp_path / ("%s_rank%d.npz" % (mode, r)))) for r in range(world)]
    flat = torch.randn(_N_TOTAL, generator=torch.Generator().manual_seed(0))
    m = torch.zeros(_N_TOTAL)
    adam = _toy_adam(flat, m)
    bf16 = grad_dtype == "bf16"
    assert all(bool(g['comm_bf16']) == bf16 for g in got)
    n_item = _N_EMB
    for step in range(3):
        gs = [_toy_grads(step, r) for r in range(world)]
        g = gs[0].clone()
        for x in gs[1:]:          # rank order: what gloo's ring / RCCL need not follow - fp32 sums of 2-3 terms are compared with a tolerance
            g = g + x
        if bf16:                   # the dense part (everything but the item table) travels in bf16; the touched rows stay fp32
            d = gs[0][n_item:].bfloat16()
            for x in gs[1:]:
                d = d + x[n_item:].bfloat16()
            g[n_item:] = d.float()
        if mode != "sharded":      # the summed gradient buffer every rank hands to Adam
            for r in range(world):
                tol = (2.0 ** -7 if bf16 else 1e-6) * float(g.abs().max())
                assert np.abs(got[r]['summed'][step] - g.numpy()).max() <= tol, (mode, r, step)
                assert np.array_equal(got[r]['summed'][step], got[0]['summed'][step]), (mode, r, step)        # identical on every rank, bit for bit
        adam(0, _N_TOTAL, torch.from_numpy(got[0]['summed'][step]) if (bf16 and mode != "sharded") else g, 0)
    for r in range(world):
        assert np.allclose(got[r]['flat'], flat.numpy(), atol=2e-6), (mode, r)      # every rank ends with the full updated parameters
        assert np.array_equal(got[r]['flat'], got[0]['flat']), (mode, r)            # ... the same bits on every rank
    for r in range(world):      # NARRuntime.state_dict(): the gathered slots are complete on every rank in every mode
        assert np.allclose(got[r]['m_full'], m.numpy(), atol=2e-6), (mode, r)
    owned = np.zeros(_N_TOTAL)
    for r in range(world):
        nz = got[r]['m'] != 0
        owned += nz
        assert np.allclose(got[r]['m'][nz], m.numpy()[nz], atol=2e-6)
    touched = m.numpy() != 0
    if mode == "sharded":
        assert (owned[touched] == 1).all()                                          # every optimizer slot lives on exactly one rank
    else:
        assert (owned[touched] == world).all()
    # the branch that ran: (all_reduce, reduce_scatter, all_gather) calls over the three steps (+ the gather of the checkpoint path)
    calls = {k: int(v) for k, v in zip(("all_reduce", "reduce_scatter", "all_gather"), got[0]['calls'])}
    if mode == "sharded":
        assert calls == {"all_reduce": 0, "reduce_scatter": 3, "all_gather": 3 + 2}, calls
    elif mode == "sparse_rs":
        assert calls["reduce_scatter"] == 3 and calls["all_gather"] == 3 and calls["all_reduce"] >= 3, calls
    else:
        assert calls["reduce_scatter"] == 0 and calls["all_gather"] == 0 and calls["all_reduce"] >= 3, calls
    if mode in ("sparse", "sparse_rs"):      # bytes of the last step (early bucket issued): dense remainder + touched rows, never the whole table
        L, early = len(_TOUCHED), 2720 - 2000
        n_dense = _N_TOTAL - _N_EMB - early
        assert int(got[0]['nbytes']) == (2 if bf16 else 4) * (n_dense + early) + 4 * L * _ITEM_DIM


# ---- the checkpoint decision under data parallelism (estimator._checkpoint_due): rank 0's wall clock, broadcast on the data-parallel
# ---- group every CKPT_DECISION_EVERY steps - every rank takes the same decision at the same step whatever its own clock says
def _ckpt_worker(rank, world, port, out_dir):
    import time
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chameleon_recsys_amd.nar.estimator import Estimator, RunConfig
    est = Estimator(model_fn=None, config=RunConfig(save_checkpoints_secs=1000))
    pg = dist.new_group(list(range(world)))
    est._store['runtime'] = types.SimpleNamespace(dp_active=True, dp_pg=pg, device="cpu")
    every = Estimator.CKPT_DECISION_EVERY
    decisions = []
    for step in range(1, 3 * every + 1):
        if step == every + 1:                 # after the first decision point: rank 0's clock says "due", rank 1's never does
            est._last_ckpt_time = time.time() - (2000 if rank == 0 else 0)
        if step == 2 * every + 1:             # ... and the other way round
            est._last_ckpt_time = time.time() - (0 if rank == 0 else 2000)
        decisions.append(bool(est._checkpoint_due()))
    np.save(os.path.join(out_dir, "ckpt_rank%d.npy" % rank), np.asarray(decisions))
    dist.destroy_process_group()


def test_checkpoint_decision_is_rank0s_and_taken_every_k_steps(tmp_path):
    from chameleon_recsys_amd.nar.estimator import Estimator
    world, every = 2, Estimator.CKPT_DECISION_EVERY
    mp.spawn(_ckpt_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d0, d1 = (np.load(str(tmp_path / ("ckpt_rank%d.npy" % r))) for r in range(world))
    assert np.array_equal(d0, d1)                                         # same decision on every rank at every step
    want = np.zeros(3 * every, bool)
    want[2 * every - 1] = True                                            # step 2K: rank 0 is due (rank 1's own clock is not)
    assert np.array_equal(d0, want), np.flatnonzero(d0)                   # step 3K: only rank 1 would be due -> nobody saves


# ---- the cooperative-kernel time-out word under data parallelism (estimator._check_device_health / _checkpoint_due): the word
# ---- is per process; MAX-reduced over the data-parallel group so that EVERY rank raises - one rank raising alone leaves the others in the
# ---- next collective
def _health_worker(rank, world, port, out_dir):
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chameleon_recsys_amd.nar.estimator import Estimator, RunConfig
    est = Estimator(model_fn=None, config=RunConfig(save_checkpoints_secs=None))      # (the poll does not depend on checkpointing being on)
    pg = dist.new_group(list(range(world)))
    bad = {"now": False}
    est._store['runtime'] = types.SimpleNamespace(dp_active=True, dp_pg=pg, device="cpu", _rnn_coop_ws={64: None},
                                                  rnn_coop_timed_out=lambda: bad["now"])
    res = []
    if rank == 0:            # evaluate()'s form: local word only, no collective - a rank may evaluate alone; would hang otherwise
        est._check_device_health(collective=False)
    est._check_device_health(); res.append("ok")                          # nobody timed out: no rank raises
    bad["now"] = rank == 1                                                # rank 1's kernels gave up a spin
    try:
        est._check_device_health(); res.append("ok")
    except RuntimeError as ex:
        res.append("raised other" if "another data-parallel rank" in str(ex) else "raised own")
    every = Estimator.CKPT_DECISION_EVERY
    try:
        for _ in range(every):
            est._checkpoint_due()
        res.append("ok")
    except RuntimeError:
        res.append("poll raised")
    with open(os.path.join(out_dir, "health_rank%d.txt" % rank), "w") as fh:
        fh.write(",".join(res))
    dist.destroy_process_group()


def test_kernel_timeout_word_raises_on_every_rank(tmp_path):
    world = 2
    mp.spawn(_health_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (open(str(tmp_path / ("health_rank%d.txt" % r))).read().split(",") for r in range(world))
    assert r0 == ["ok", "raised other", "poll raised"], r0
    assert r1 == ["ok", "raised own", "poll raised"], r1


# ---- bf16 gradient exchange ("6.5 MB bf16"; CHAM_DP_GRAD_DTYPE, default bf16 exactly when the runtime computes in bf16 -
# ---- BASELINE configs[2]): the flat gradients go through the collectives rounded to bf16 - early bucket and remainder alike - and come back
# ---- widened into the fp32 buffer Adam reads, identical on every rank
def _bf16_worker(rank, world, port, out_dir, dtype_env, gemm_dtype):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), CHAM_DP_GRAD_DTYPE=dtype_env)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)

    class _E:
        def __init__(self, name, offset, size):
            self.name, self.offset, self.size = name, offset, size

    class _Layout:
        emb_end = _N_EMB
        entries = {n: _E(n, o, sz) for n, o, sz in (('emb', 0, _N_EMB), ('W2', _N_EMB, 500), ('Wf1', 2000, 300), ('Wf2', 2300, 200),
                                                    ('Ws1', 2500, 100), ('Ws2', 2600, 60), ('Ws3', 2660, 40), ('Ws4', 2700, 20),
                                                    ('b1', 2720, _N_TOTAL - 2720))}

    class _RT:
        flat = torch.zeros(_N_TOTAL)
        layout = _Layout()
        dp_rank = dp_world = dp_allreduce = dp_sharded = dp_early_bucket = dp_gather_slots = None
    _RT.gemm_dtype = gemm_dtype

    class _Model:
        rt = _RT()
    dp = parallel.DataParallelNAR(_Model(), mode="allreduce")
    outs = []
    for step in range(2):
        grads = torch.randn(_N_TOTAL, generator=torch.Generator().manual_seed(300 + 10 * step + rank))
        if step == 0:
            _Model.rt.dp_early_bucket(grads)
        _Model.rt.dp_allreduce(grads)
        outs.append(grads.numpy().copy())
    np.savez(os.path.join(out_dir, "bf16_rank%d.npz" % rank), g0=outs[0], g1=outs[1], comm_bf16=dp.comm_bf16, nbytes=dp.last_exchange_bytes)
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype_env,gemm_dtype,want_bf16", [("auto", "bf16", True), ("auto", "f32", False), ("bf16", "f32", True), ("f32", "bf16", False)])
def test_bf16_gradient_exchange(tmp_path, dtype_env, gemm_dtype, want_bf16):
    world = 2
    mp.spawn(_bf16_worker, args=(world, _free_port(), str(tmp_path), dtype_env, gemm_dtype), nprocs=world, join=True)
    got = [np.load(str(tmp_path / ("bf16_rank%d.npz" % r))) for r in range(world)]
    assert all(bool(g['comm_bf16']) == want_bf16 for g in got)
    for step, key in enumerate(("g0", "g1")):
        gs = [torch.randn(_N_TOTAL, generator=torch.Generator().manual_seed(300 + 10 * step + r)) for r in range(world)]
        if want_bf16:
            ref = (gs[0].bfloat16() + gs[1].bfloat16()).float()         # two-rank sum in bf16: one rounding of the sum of two roundings
        else:
            ref = gs[0] + gs[1]
        for r in range(world):
            assert np.array_equal(got[r][key], ref.numpy()), (key, r, float(np.abs(got[r][key] - ref.numpy()).max()))
        if want_bf16:       # ... and within bf16 resolution of the fp32 sum
            full = (gs[0] + gs[1]).numpy()
            assert np.abs(got[0][key] - full).max() <= 2.0 ** -7 * np.abs(full).max()
    # bytes handed to the collectives of the last step (no early bucket there): the whole flat buffer at 2 or 4 bytes per gradient
    assert int(got[0]['nbytes']) == (2 if want_bf16 else 4) * _N_TOTAL
