"""Data-parallel HIP step end to end on ONE GPU: two processes (both on cuda:0, gloo transport - NCCL/RCCL refuses two
ranks on one device) run DataParallelNAR for a few optimizer steps on a shared global batch stream; the result must equal
the single-process run on the full batches (row sharding, sampler keyed by the global row, global denominators, ONE
all-reduce of the flat gradient buffer, replicated state update)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chameleon_recsys_amd.nar import synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu
STEPS = 3


def _params():
    return H.tiny_params(C=128, H=100, neg=8, batch_size=48)


def _run(dp_world, rank, batches, p):
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.parallel import DataParallelNAR
    model, _ = H.make_pair(p, seed=7)
    dp = DataParallelNAR(model)
    st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'],
                                 p['recent_clicks_for_normalization'], 1000)
    for f, l in batches[:2]:          # warm state (identical on every rank)
        aci = np.concatenate([f['item_clicked'], l['label_last_item']], 1)
        st.update_from_device_batch(torch.from_numpy(aci).cuda(), torch.from_numpy(f['event_timestamp']).cuda())
    losses = []
    for f, l in batches[2:2 + STEPS]:
        model.feed_state(st, st)
        d = dp.upload(f, l)
        model.train_step(d)
        losses.append(dp.global_loss().cpu().numpy())
        st.update_from_device_batch(d['aci'], d['g_event_ts'])
    torch.cuda.synchronize()
    sd = model.rt.state_dict()        # checkpoint image: a collective in the sharded / hybrid modes (gathers the Adam slots)
    return (np.stack(losses), model.rt.flat.cpu().numpy(), model.rt.m.cpu().numpy(), getattr(dp, 'emb_sharded', model.rt.layout.emb_end),
            sd['m'].numpy(), sd['dp_mode'])


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(CHAM_DP_MODE=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _params()
    batches = synthetic.make_batches(2 + STEPS, 48, 8, 1000, p['session_features_config'], length_dist='g1', seed=6)
    losses, flat, m, E, m_ckpt, ckpt_mode = _run(world, rank, batches, p)
    assert ckpt_mode == mode
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=losses, flat=flat, m=m, E=E, m_ckpt=m_ckpt)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "sharded", "hybrid", "sparse", "sparse_rs"])
def test_two_rank_hip_training_equals_single_process(gpu, tmp_path, mode):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0, r1 = np.load(str(tmp_path / "rank0.npz")), np.load(str(tmp_path / "rank1.npz"))
    assert np.array_equal(r0['flat'], r1['flat'])                                           # replicas stay bit-identical
    assert np.allclose(r0['losses'], r1['losses'], atol=1e-7)
    n = r0['m'].shape[0] // 2
    if mode == "sharded":       # Adam slots live on the rank that owns the parameter slice
        assert not r0['m'][n:].any() and not r1['m'][:n].any()
        m_dp = np.concatenate([r0['m'][:n], r1['m'][n:]])
    elif mode == "hybrid":      # embedding-table slots on the owning rank, dense slots replicated
        E = int(r0['E']); h = E // 2
        assert E > 0 and not r0['m'][h:E].any() and not r1['m'][:h].any() and np.array_equal(r0['m'][E:], r1['m'][E:])
        m_dp = np.concatenate([r0['m'][:h], r1['m'][h:E], r0['m'][E:]])
    else:
        assert np.array_equal(r0['m'], r1['m'])
        m_dp = r0['m']
    p = _params()
    batches = synthetic.make_batches(2 + STEPS, 48, 8, 1000, p['session_features_config'], length_dist='g1', seed=6)
    losses, flat, m, _, _, _ = _run(1, 0, batches, p)
    # a checkpoint written by ANY rank holds the complete Adam slots, whatever the exchange mode
    assert np.array_equal(r0['m_ckpt'], m_dp) and np.array_equal(r1['m_ckpt'], m_dp)
    assert np.abs(losses - r0['losses']).max() < 2e-5, (losses, r0['losses'])
    # the sum of two row shards' gradients differs from the single-process gradient in the last bits (fp32 summation order);
    # every kernel of the step is deterministic (no float atomics), so that is the only difference
    from chameleon_recsys_amd.nar.nar_model import NARRuntime
    H.assert_flat_close(NARRuntime(p).layout, r0['flat'], m_dp, flat, m, p['lr'], n_steps=STEPS, m_tol=1e-4)
