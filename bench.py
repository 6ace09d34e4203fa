#!/usr/bin/env python
"""bench.py - NAR training sessions/sec on MI355X (BASELINE.json metric), G1-shape synthetic workload.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one optimizer step of the NAR hot path (negative sampling -> features -> CAR -> UGRNN -> scorer ->
sampled-softmax loss -> backward -> L2 + TF-Adam -> recent-clicks state update) on one batch of 256 sessions PER
GPU (weak scaling) whose tensors are already resident in HBM.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

G1 = dict(n_items=46000, ace_dim=250, seq_len=20, batch=256, neg=50, neg_from_buffer=3000, buffer=20000, for_norm=2000,
          C=1024, H=255)
TINY = dict(n_items=1000, ace_dim=64, seq_len=8, batch=64, neg=10, neg_from_buffer=100, buffer=2000, for_norm=200,
            C=1024, H=255)
# BASELINE.json configs[3]: Adressa-shape (13k articles, seq_len 30, 2-layer GRU hidden 256, 100 negatives); hyper-parameters of
# the reference's Adressa script
ADRESSA = dict(n_items=13000, ace_dim=250, seq_len=30, batch=256, neg=100, neg_from_buffer=5000, buffer=20000, for_norm=5000,
               C=1024, H=256, dataset='adressa', rnn_cell='gru', rnn_num_layers=2, softmax_temperature=0.2, lr=3e-4,
               reg_weight_decay=1e-4)
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz


def dense_step_flops(B, T, N, F, C, H, L=1):
    """Reference-dense forward FLOPs per step (SURVEY.md 8d / BASELINE.md section 4); training ~ 3x."""
    rows_neg, rows_io = B * T * N, B * T
    fwd = 2 * (rows_neg + 2 * rows_io) * (F * C + C * C)
    I = C
    for _ in range(L):
        fwd += 2 * B * T * (I + H) * 2 * H
        I = H
    fwd += 2 * rows_io * (H * 512 + 512 * C)
    fwd += 2 * (rows_neg + rows_io) * (C * 128 + 128 * 64 + 64 * 32 + 32)
    return fwd


def cpu_baseline(params, cfg, length_dist, seed):
    """The restated CPU oracle ("port": TF 1.12 cannot run here) on a bounded sample of the same workload."""
    import torch
    from chameleon_recsys_amd.nar import synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, batch_clicks_for_state
    from oracle.nar_oracle import NAROracle
    cores = min(len(os.sched_getaffinity(0)), 32)     # beyond ~32 threads the oracle's many small ops only get slower
    torch.set_num_threads(cores)
    Bs = 64                                    # sample: 64-session batches of the same shape
    p = dict(params); p['batch_size'] = Bs
    batches = synthetic.make_batches(8, Bs, cfg['seq_len'], cfg['n_items'], p['session_features_config'], seed=seed,
                                     length_dist=length_dist, sessions_per_hour=Bs * 4)
    orc = NAROracle(p, seed=seed)
    st = ClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'])
    times = []
    for i, (f, l) in enumerate(batches):
        t0 = time.perf_counter()
        orc.train_step(f, l, st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm())
        ids, ts = batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp'])
        st.update_items_state(ids, ts)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:])                        # first step = warm-up
    return dict(value=round(Bs * (len(times) - 1) / dt, 3), unit="sessions/s", cores=cores, kind="port",
                sample="%d timed optimizer steps of 64-session batches (same shape, 1 warm-up step), restated CPU oracle "
                       "(PyTorch-CPU fp32, TF 1.12 unavailable), %d threads" % (len(times) - 1, cores))


def hitrate_parity(seed):
    """HitRate@5 / MRR@5 of the HIP path vs the CPU oracle on identical inputs (the second half of BASELINE.json's metric):
    G1-tiny shape (configs[0]), 30 optimizer steps from the same initial weights on both, then 4 held-out batches ranked
    against 20 sampled negatives (bit-identical negatives on both sides)."""
    import torch
    from chameleon_recsys_amd.nar import metrics, synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, batch_clicks_for_state
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime
    from oracle.nar_oracle import NAROracle
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 32))
    p = synthetic.default_params(1000, 64, seq_len=8, batch_size=64, neg=10, neg_from_buffer=100, buffer_size=2000, for_norm=200,
                                 C=128, H=255, seed=seed, lr=1e-3, eval_total_negative_samples=20, eval_negative_samples_from_buffer=200)
    batches = synthetic.make_batches(34, 64, 8, 1000, p['session_features_config'], seed=seed, length_dist='g1')
    rt = NARRuntime(p, seed=seed)
    orc = NAROracle(p, weights=rt.logical_weights())
    mk = lambda mode, n, nb: NARModuleModel(mode, None, None, p['session_features_config'], p['articles_features_config'], 64,
                                            p['lr'], 1.0, n, nb, p['content_article_embeddings_matrix'],
                                            softmax_temperature=p['softmax_temperature'], reg_weight_decay=p['reg_weight_decay'],
                                            recent_clicks_buffer_max_size=2000, recent_clicks_for_normalization=200,
                                            articles_metadata=p['articles_metadata'], CAR_embedding_size=128, rnn_units=255, runtime=rt)
    train, ev = mk(ModeKeys.TRAIN, 10, 100), mk(ModeKeys.EVAL, 20, 200)
    st = ClickedItemsState(1.0, 2000, 200, 1000)
    for f, l in batches[:30]:
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        train.feed_state(pop, buf); train.train_step(train.upload_batch(f, l))
        orc.train_step(f, l, buf, pop)
        st.update_items_state(*batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp']))
    # third column: the oracle EVALUATING THE HIP-TRAINED WEIGHTS - separates the eval path (must agree to the last hit) from the
    # divergence of two fp32 training runs (30 Adam steps at lr 1e-3 amplify last-bit differences; the per-step bounds are in
    # tests/test_step_gpu.py::test_training_curve_matches_oracle)
    orc_hw = NAROracle(p, weights=rt.logical_weights())
    m = {k: (metrics.HitRate(5), metrics.MRR(5)) for k in ("hip", "cpu_oracle", "cpu_oracle_on_hip_weights")}
    for i, (f, l) in enumerate(batches[30:]):
        buf, pop = st.get_recent_clicks_buffer().copy(), st.get_articles_recent_pop_norm().copy()
        ev.feed_state(pop, buf); ev.evaluate_step(ev.upload_batch(f, l))
        ids = ev.predicted_item_ids.eval()
        key = NARModuleModel.eval_step_key(orc.global_step, i)
        ref = orc.forward(f, l, buf, pop, mode='eval', step=key)
        ref_hw = orc_hw.forward(f, l, buf, pop, mode='eval', step=key)
        for k, pred in (("hip", ids), ("cpu_oracle", ref['predicted_item_ids'].numpy()),
                        ("cpu_oracle_on_hip_weights", ref_hw['predicted_item_ids'].numpy())):
            m[k][0].add(pred, l['label_next_item']); m[k][1].add(pred, l['label_next_item'])
        st.update_items_state(*batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp']))
    return {"hitrate_at_5": {k: round(v[0].result(), 5) for k, v in m.items()},
            "mrr_at_5": {k: round(float(v[1].result()), 5) for k, v in m.items()},
            "protocol": "G1-tiny synthetic, 30 training steps (lr 1e-3) + 4 eval batches x 64 sessions, 20 eval negatives; same initial "
                        "weights / inputs / negatives on both sides; cpu_oracle_on_hip_weights = the oracle evaluating the HIP-trained weights"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="g1", choices=["g1", "tiny", "adressa"],
                    help="g1 = BASELINE.json configs[1] (the headline); adressa = configs[3] (2-layer GRU, 100 negatives); tiny = configs[0]")
    ap.add_argument("--length-dist", default="full", choices=["full", "g1"],
                    help="full: every session has seq_len clicks (no padded rows); g1: G1-like ragged lengths")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32: exact fp32 MFMA (BASELINE configs[1], the headline); bf16: bf16-rounded GEMM operands, fp32 accumulate/"
                         "storage/softmax/loss/Adam (BASELINE configs[2] arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ragged-leg", action="store_true",
                    help="skip the secondary G1-like-session-lengths leg (profiling runs: keeps per-symbol averages to the headline leg)")
    ap.add_argument("--state", default="device", choices=["device", "host"],
                    help="recent-clicks state: device-resident (csrc/state.hip) or the host numpy class fed every step")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from chameleon_recsys_amd.nar import synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, DeviceClickedItemsState, batch_clicks_for_state
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime
    from chameleon_recsys_amd.nar.parallel import DataParallelNAR

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    # CHAM_DIST_BACKEND=gloo: debugging aid - exercises this script's multi-rank path with several ranks on ONE GPU (RCCL refuses
    # two ranks per device); the driver's runs use the default, RCCL with one rank per GPU
    backend = os.environ.get("CHAM_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = {"g1": G1, "tiny": TINY, "adressa": ADRESSA}[args.config]
    Bl = cfg['batch']                 # per-GPU batch (weak scaling)
    Bg = Bl * world
    params = synthetic.default_params(cfg['n_items'], cfg['ace_dim'], seq_len=cfg['seq_len'], batch_size=Bg, neg=cfg['neg'],
                                      neg_from_buffer=cfg['neg_from_buffer'], buffer_size=cfg['buffer'],
                                      for_norm=cfg['for_norm'], C=cfg['C'], H=cfg['H'], seed=args.seed,
                                      **{k: cfg[k] for k in ('dataset', 'rnn_cell', 'rnn_num_layers', 'softmax_temperature', 'lr',
                                                             'reg_weight_decay') if k in cfg})
    params['gemm_dtype'] = args.dtype
    n_distinct = 8
    batches = synthetic.make_batches(n_distinct, Bg, cfg['seq_len'], cfg['n_items'], params['session_features_config'],
                                     seed=args.seed, length_dist=args.length_dist, sessions_per_hour=Bg * 2)
    rt = NARRuntime(params, device="cuda:%d" % local_rank, seed=args.seed)
    model = NARModuleModel(ModeKeys.TRAIN, None, None, params['session_features_config'], params['articles_features_config'],
                           Bg, params['lr'], 1.0, cfg['neg'], cfg['neg_from_buffer'], params['content_article_embeddings_matrix'],
                           softmax_temperature=params['softmax_temperature'], reg_weight_decay=params['reg_weight_decay'],
                           recent_clicks_buffer_max_size=cfg['buffer'], recent_clicks_for_normalization=cfg['for_norm'],
                           articles_metadata=params['articles_metadata'], CAR_embedding_size=cfg['C'], rnn_units=cfg['H'],
                           runtime=rt)
    dp = DataParallelNAR(model)
    if args.state == "device":
        state = DeviceClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'], device="cuda:%d" % local_rank)
    else:
        state = ClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'])
    dev_batches = [dp.upload(f, l) for f, l in batches]          # inputs resident in HBM before the timed region
    host_clicks = [batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp']) for f, l in batches]

    def one_step(i):
        k = i % n_distinct
        if args.state == "device":
            model.feed_state(state, state)                                                              # hook.before_run
            model.train_step(dev_batches[k])
            state.update_from_device_batch(dev_batches[k]['aci'], dev_batches[k]['g_event_ts'])        # hook.after_run
            model.presample(dev_batches[(k + 1) % n_distinct])        # next batch's negatives behind the state update
        else:
            model.feed_state(state.get_articles_recent_pop_norm(), state.get_recent_clicks_buffer())
            model.train_step(dev_batches[k])
            state.update_items_state(*host_clicks[k])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = dp.global_loss().cpu().numpy()

    # ---- roofline leg: HIP-event timing of every GEMM launch over a few extra steps -------------------------
    rt.profile = []
    nprof = 3
    for i in range(nprof):
        one_step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    prof, rt.profile = rt.profile, None
    # dominant kernel SYMBOL = gemm_f32_kernel<256,128,4,2,16,true,false,2,true,0,false>: NN layout, bias + tanh epilogue - the CAR layer-2
    # forward over the B*T*(1+N) candidate rows (the small NN bias+tanh GEMMs - CAR layer 2 on the clicked rows, session FC2 -
    # run on the 128x128 instance, a different symbol).  Aggregated over all of its launches exactly like `rocprofv3 --stats`
    # aggregates per kernel symbol, so avg_launch_ms is comparable with the committed kernel_stats.
    def agg(sel):
        rows = [r for r in prof if sel(r)]
        ms = sum(r['ev'][0].elapsed_time(r['ev'][1]) for r in rows)
        fl = sum(2.0 * r['M'] * r['N'] * r['K'] for r in rows)
        return len(rows), ms, fl
    # launches that gemm.hip's launch_by_shape sends to the 256x128 NN bias+tanh instance (grid of at least 256 workgroups)
    dom = lambda r: (r['N'] > 64 and r['M'] * r['N'] >= (1 << 20) and -(-r['M'] // 256) * -(-r['N'] // 128) >= 256
                     and not r['transA'] and not r['transB'] and r['act'] == 2)
    n_nn, ms_nn, fl_nn = agg(dom)
    n_all, ms_all, fl_all = agg(lambda r: True)
    achieved = fl_nn / (ms_nn * 1e-3) / 1e12 if ms_nn > 0 else 0.0
    DOM_SYMBOL = "void gemm_f32_kernel<256, 128, 4, 2, 16, true, false, 2, true, 0, false>(GemmParams)"
    traffic, traffic_src = None, None
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if fn.endswith("_pmc_traffic.json"):     # PMC passes cannot be collected inside the timed bench: committed separately
            rec = json.load(open(os.path.join(ROOT, "profiles", fn))).get(DOM_SYMBOL)
            if rec and args.config == "g1" and world == 1:
                traffic, traffic_src = rec["traffic_bytes_per_launch"], "profiles/" + fn
            break

    # ---- secondary leg: the same step on G1-LIKE session lengths (2 + min(Geometric(0.45), seq_len - 2), mean
    # ~4 clicks, zero-padded to the batch maximum).  The headline above is the "all sessions full length" stress/roofline
    # variant (no padded positions: the most work a session can carry); with ragged sessions the step runs its row-wise
    # stages on the valid positions only (nar_model.upload_batch), so sessions/s rises with the padding fraction.
    ragged = None
    if args.length_dist == "full" and args.state == "device" and not args.no_ragged_leg:
        rb = synthetic.make_batches(n_distinct, Bg, cfg['seq_len'], cfg['n_items'], params['session_features_config'],
                                    seed=args.seed + 1, length_dist="g1", sessions_per_hour=Bg * 2)
        rdev = [dp.upload(f, l) for f, l in rb]

        def ragged_step(i):
            k = i % n_distinct
            model.feed_state(state, state)
            model.train_step(rdev[k])
            state.update_from_device_batch(rdev[k]['aci'], rdev[k]['g_event_ts'])
            model.presample(rdev[(k + 1) % n_distinct])
        for i in range(max(n_distinct, args.warmup)):        # every distinct padded length T allocates its StepPlan once
            ragged_step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ragged_step(args.warmup + i)
        barrier()
        rdt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([rdt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            rdt = float(tt.item())
        ragged = {"session_lengths": "g1: 2 + min(Geometric(0.45), seq_len - 2), zero-padded to the batch maximum",
                  "value": round(Bg * args.steps / rdt, 2), "unit": "sessions/s", "ms_per_step": round(rdt / args.steps * 1e3, 3),
                  "mean_clicks_per_session": round(float(np.mean([np.asarray(f['session_size']).mean() for f, _ in rb])), 2),
                  "valid_positions_of_padded": round(float(sum(d['P'] for d in rdev)) / float(sum(d['B'] * d['T'] for d in rdev)), 4),
                  "padded_T": [int(d['T']) for d in rdev]}

    if rank == 0:
        L = rt.layout
        T = cfg['seq_len'] - 1
        dense_fwd = dense_step_flops(Bl, T, cfg['neg'], L.F, cfg['C'], cfg['H'], cfg.get('rnn_num_layers', 1))
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "NAR training sessions/sec", "value": round(Bg * args.steps / dt, 2), "unit": "sessions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": {"g1": "G1-shape synthetic (BASELINE.json configs[1])", "tiny": "G1-tiny synthetic (configs[0])",
                                    "adressa": "Adressa-shape synthetic (BASELINE.json configs[3])"}[args.config],
                       "n_items": cfg['n_items'], "ace_dim": cfg['ace_dim'], "seq_len": cfg['seq_len'],
                       "sessions_per_gpu_per_step": Bl, "global_batch": Bg, "negatives": cfg['neg'],
                       "CAR_embedding_size": cfg['C'], "rnn_units": cfg['H'], "rnn_cell": cfg.get('rnn_cell', 'ugrnn'), "rnn_layers": cfg.get('rnn_num_layers', 1),
                       "session_lengths": args.length_dist, "parallelism": "dp%d" % world, "clicked_items_state": args.state,
                       "final_loss": [round(float(x), 5) for x in loss]},
            "roofline": {"bound": "mfma", "kernel": DOM_SYMBOL + " = fp32 MFMA GEMM, NN, bias+tanh (CAR layer 2 forward), all launches of a step",
                         "achieved": round(achieved, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "launches_per_step": n_nn // nprof, "avg_launch_ms": round(ms_nn / max(1, n_nn), 4),
                         "algorithmic_gflop_per_launch": round(fl_nn / max(1, n_nn) / 1e9, 3),
                         # operands + output of the launch(es), each touched once: A [M,K] + W [K,N] + bias + out [M,N], fp32
                         "algorithmic_bytes_per_launch": round(sum((r['M'] * r['K'] + r['K'] * r['N'] + r['N'] + r['M'] * r['N']) * 4.0
                                                                   for r in prof if dom(r)) / max(1, n_nn)),
                         "all_gemm_ms_per_step": round(ms_all / nprof, 3),
                         "all_gemm_tflops": round(fl_all / (ms_all * 1e-3) / 1e12, 2) if ms_all > 0 else 0.0,
                         "step_reference_dense_tflops": round(3 * dense_fwd / (ms_step * 1e-3) / 1e12, 2)},
        }
        if args.dtype == "bf16":     # operands move as fp32, the matrix cores run 16x faster: the same launches are HBM/L2-bound
            algo_bytes = out["roofline"]["algorithmic_bytes_per_launch"] or 0
            gbs = algo_bytes / (ms_nn / max(1, n_nn) * 1e-3) / 1e9 if ms_nn > 0 else 0.0
            out["roofline"].update(bound="hbm", kernel="gemm_bf16_kernel<256,128,4,2,32,true,false,2,false> = bf16-MFMA GEMM, NN, bias+tanh "
                                   "(CAR layer 2 forward), all launches of a step", achieved=round(gbs, 1), peak=8000.0, unit="GB/s",
                                   frac=round(gbs / 8000.0, 4), traffic=None, traffic_source=None)
        if ragged is not None:
            out["g1_like_session_lengths"] = ragged
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, cfg, args.length_dist, args.seed)
            out["accuracy_vs_cpu_ref"] = hitrate_parity(args.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
