"""RCCL on the one GPU a test box has: a process group of ONE rank on the "nccl" backend (= RCCL on ROCm) drives DataParallelNAR with
the world > 1 guard lifted (CHAM_DP_FORCE=1), so every collective of every exchange mode executes on RCCL, stream-ordered with the
step's two lanes - all_reduce in two buckets (the early one asynchronously from the side lane), reduce_scatter_tensor,
all_gather_into_tensor, the packed sparse exchange, the all_gather of the Adam slots for a checkpoint - i.e. the branches of
nar/parallel.py that the two-process tests (tests/test_dp_gpu.py, gloo: RCCL refuses two ranks on one device) cannot reach.  With one
rank every collective is the identity, so weights, Adam slots and losses must be BIT-identical to the plain single-process step."""
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


def _params(**over):
    kw = dict(C=128, H=100, neg=8, batch_size=48)
    kw.update(over)
    return H.tiny_params(**kw)


# (C = 128: the small-tile kernels; C = 256: the plane-resident CAR GEMMs with their W2 plane shadows / the bf16 LDS-DMA core with bf16 shadows)
ARITH = [("f32", 128), ("f32", 256), ("bf16", 256)]


def _train(p, batches, dp_mode, round_grads_bf16=False):
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.parallel import DataParallelNAR
    model, _ = H.make_pair(p, seed=7)
    dp = DataParallelNAR(model, mode=dp_mode) if dp_mode else None
    if round_grads_bf16:      # what a bf16 exchange with ONE rank must equal: the plain step with its gradients rounded to bf16 before Adam
        model.rt.dp_allreduce = lambda g: g.copy_(g.bfloat16().float())
    st = DeviceClickedItemsState(p['recent_clicks_buffer_hours'], p['recent_clicks_buffer_max_size'],
                                 p['recent_clicks_for_normalization'], 1000)
    for f, l in batches[:2]:
        aci = np.concatenate([f['item_clicked'], l['label_last_item']], 1)
        st.update_from_device_batch(torch.from_numpy(aci).cuda(), torch.from_numpy(f['event_timestamp']).cuda())
    losses = []
    for f, l in batches[2:2 + STEPS]:
        model.feed_state(st, st)
        d = dp.upload(f, l) if dp else model.upload_batch(f, l)
        model.train_step(d)
        losses.append((dp.global_loss() if dp else model.total_loss).cpu().numpy().copy())
        st.update_from_device_batch(d['aci'], d['g_event_ts'])
    torch.cuda.synchronize()
    sd = model.rt.state_dict()
    info = (dp.active, dp.comm_bf16, dp.last_exchange_bytes, int(model.rt.flat.numel())) if dp else (False, False, 0, 0)
    return np.stack(losses), model.rt.flat.cpu().numpy(), sd['m'].numpy(), sd['v'].numpy(), info


def _worker(rank, port, out_dir):
    # (CHAM_DP_GRAD_DTYPE=f32: the bit-identity below is about the collectives' plumbing; the bf16 exchange has its own case at the end)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CHAM_DP_FORCE="1", CHAM_DP_GRAD_DTYPE="f32")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    out = {}
    for dtype, C in ARITH:
        p = _params(C=C, gemm_dtype=dtype)
        batches = synthetic.make_batches(2 + STEPS, 48, 8, 1000, p['session_features_config'], length_dist='g1', seed=6)
        ref = _train(p, batches, None)
        for mode in ("allreduce", "sharded", "sparse", "sparse_rs"):
            losses, flat, m, v, info = _train(p, batches, mode)
            assert info[0] and not info[1], "CHAM_DP_FORCE did not install the exchange hooks (or the exchange is not fp32)"
            out["%s/C%d/%s" % (dtype, C, mode)] = (bool(np.array_equal(losses, ref[0])), bool(np.array_equal(flat, ref[1])), bool(np.array_equal(m, ref[2])),
                                                   bool(np.array_equal(v, ref[3])), float(np.abs(losses - ref[0]).max()))
    # BASELINE configs[2] as SURVEY 8e words it: bf16 compute AND the dense gradients exchanged in bf16 (CHAM_DP_GRAD_DTYPE=auto picks bf16
    # for a bf16 runtime; 2 bytes per gradient through RCCL).  One rank: the collective is the identity on the ROUNDED gradients, so the run
    # must be bit-identical to the plain step with its gradients rounded to bf16 before Adam (early bucket + remainder, async + sync paths)
    os.environ["CHAM_DP_GRAD_DTYPE"] = "auto"
    p = _params(C=256, gemm_dtype="bf16")
    batches = synthetic.make_batches(2 + STEPS, 48, 8, 1000, p['session_features_config'], length_dist='g1', seed=6)
    ref = _train(p, batches, None, round_grads_bf16=True)
    losses, flat, m, v, info = _train(p, batches, "allreduce")
    assert info[0] and info[1] and info[2] == 2 * info[3], info
    out["bf16/C256/allreduce+bf16 exchange"] = (bool(np.array_equal(losses, ref[0])), bool(np.array_equal(flat, ref[1])), bool(np.array_equal(m, ref[2])),
                                                bool(np.array_equal(v, ref[3])), float(np.abs(losses - ref[0]).max()))
    # the 33.6 MB-class two-bucket exchange itself, timed on this rank's stream (bench.py reports the same as dp_self_exchange_ms)
    g = torch.zeros(8 << 20, device="cuda")
    for _ in range(3):
        dist.all_reduce(g[:1 << 20]); dist.all_reduce(g[1 << 20:])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "result.npy"), np.array([out], dtype=object), allow_pickle=True)
    dist.destroy_process_group()


def test_every_exchange_mode_on_rccl_world_of_one_is_bit_identical(gpu, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    out = np.load(str(tmp_path / "result.npy"), allow_pickle=True)[0]
    assert len(out) == 13          # three arithmetics x four exchange modes (fp32 exchange) + bf16 compute with the bf16 exchange (BASELINE configs[2] as written)
    for mode, (l_ok, w_ok, m_ok, v_ok, dl) in out.items():
        assert l_ok and w_ok and m_ok and v_ok, "mode %s on RCCL: losses %s (max diff %g) weights %s m %s v %s" % (mode, l_ok, dl, w_ok, m_ok, v_ok)
