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
X3_MATRIX_PEAK_TFLOPS = 2500.0 / 6    # fp32 GEMM as six bf16-plane products per fp32 product (csrc/gemm_x3.hip): the bf16 dense peak / 6
H2_MATRIX_PEAK_TFLOPS = 2500.0 / 3    # fp32-grade GEMM as three fp16-plane products per fp32 product (csrc/gemm_h2.hip): the fp16 dense peak / 3
BF16_MATRIX_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (AMD's 5 PFLOP/s headline includes 2:1 sparsity)
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E


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


def boundary_setup(cfg, d, n_steps, length_dist, seed, state="device", gemm_dtype="f32"):
    """What both legs that go through the drop-in boundary start from: GZIP TFRecord session files for n_steps batches (written by the
    synthetic generator under directory d: the same seed gives the same files), the ACR-module resources loaded the way the trainer loads
    them, and the Estimator the trainer's build_estimator returns for them.  -> (trainer module, estimator, files, session features config)"""
    from chameleon_recsys_amd.nar import nar_trainer_gcom as T, synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, DeviceClickedItemsState
    B = cfg['batch']
    per_file = 50 * B
    n_files = -(-n_steps * B // per_file)
    files, csv, pkl = synthetic.write_dataset(d, n_files, per_file, cfg['n_items'], cfg['ace_dim'], seq_len=cfg['seq_len'],
                                              seed=seed, length_dist=length_dist)
    argv = ['--batch_size', str(B), '--truncate_session_length', str(cfg['seq_len']), '--learning_rate', '1e-4', '--reg_l2', '1e-5',
            '--softmax_temperature', '0.1', '--recent_clicks_buffer_max_size', str(cfg['buffer']),
            '--recent_clicks_for_normalization', str(cfg['for_norm']), '--eval_metrics_top_n', '5',
            '--CAR_embedding_size', str(cfg['C']), '--rnn_units', str(cfg['H']),
            '--train_total_negative_samples', str(cfg['neg']), '--train_negative_samples_from_buffer', str(cfg['neg_from_buffer']),
            '--eval_total_negative_samples', str(cfg['neg']), '--eval_negative_samples_from_buffer', str(cfg['neg_from_buffer']),
            '--content_embedding_scale_factor', '6.0', '--disable_eval_benchmarks', '--model_dir', os.path.join(d, 'model'),
            '--clicked_items_state', state, '--gemm_dtype', gemm_dtype]
    T.FLAGS = T.define_flags().parse_args(argv)
    meta_df, ace = T.load_acr_module_resources(csv, pkl)
    ace = T.l2_normalize_rows(ace) * np.float32(6.0)
    acfg = T.get_articles_features_config(n_items=ace.shape[0])
    meta = T.process_articles_metadata(meta_df, acfg)
    scfg = T.get_session_features_config()
    T.eval_sessions_metrics_log = []
    T.clicked_items_state = (DeviceClickedItemsState if state == "device" else ClickedItemsState)(1.0, cfg['buffer'], cfg['for_norm'], ace.shape[0])
    est = T.build_estimator(os.path.join(d, 'model'), ace, meta, acfg, scfg)
    est.config.log_step_count_steps = 0
    est.config.save_checkpoints_secs = 0
    return T, est, files, scfg


def cpu_baseline(cfg, length_dist, seed, n_steps=20):
    """The restated CPU oracle ("port": TF 1.12 cannot run here) on a bounded sample of THE SAME JOB the through-boundary leg runs
    (SURVEY 8d): the first batches of the same GZIP TFRecord session files (same generator, same seed), decoded by the same input_fn
    (datasets.SessionDataset), at the workload's own batch size, with the Estimator params the trainer builds - 1 warm-up + n_steps timed
    optimizer steps (forward, backward, TF-Adam, recent-clicks state update), with the per-stage breakdown."""
    import shutil
    import tempfile
    import torch
    from chameleon_recsys_amd.nar import datasets
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, batch_clicks_for_state
    from oracle.nar_oracle import NAROracle
    host_cores = len(os.sched_getaffinity(0))         # what the node gives this process ("core count printed")
    cores = min(host_cores, 32)                       # threads actually used: beyond ~32 the oracle's many small ops only get slower
    torch.set_num_threads(cores)
    B = cfg['batch']
    d = tempfile.mkdtemp(prefix="cham_cpu_")
    try:
        T, est, files, scfg = boundary_setup(cfg, d, n_steps + 1, length_dist, seed, state="host")
        orc = NAROracle(est.params, seed=seed)
        st = ClickedItemsState(1.0, cfg['buffer'], cfg['for_norm'], cfg['n_items'])
        times, t_in = [], 0.0
        it = iter(datasets.SessionDataset(files, scfg, batch_size=B, truncate_sequence_length=cfg['seq_len']))
        for i in range(n_steps + 1):
            t0 = time.perf_counter()
            f, l = next(it)                        # (decode + batch: the C++ codec's share of the step, reported separately)
            t1 = time.perf_counter()
            if i == 1:
                orc.timers = {}                    # first step = warm-up
                t_in = 0.0
            orc.train_step(f, l, st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm())
            with orc._stage('state_update'):
                ids, ts = batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp'])
                st.update_items_state(ids, ts)
            times.append(time.perf_counter() - t0)
            t_in += t1 - t0
    finally:
        shutil.rmtree(d, ignore_errors=True)
    dt, n = sum(times[1:]), len(times) - 1
    stages = {k: round(v / n * 1e3, 1) for k, v in orc.timers.items()}
    stages['input_fn'] = round(t_in / n * 1e3, 1)
    return dict(value=round(B * n / dt, 3), unit="sessions/s", cores=cores, host_cores_visible=host_cores, host_cpu_count=os.cpu_count(),
                torch_threads=torch.get_num_threads(), kind="port",
                sample="%d timed optimizer steps (1 warm-up) of %d-session batches decoded from the through-boundary leg's own GZIP TFRecord "
                       "files (same generator + seed, %s session lengths) by the same input_fn, restated CPU oracle (PyTorch-CPU fp32, TF 1.12 "
                       "unavailable), %d threads on a host that shows this process %d cores" % (n, B, length_dist, cores, host_cores),
                ms_per_step=round(dt / n * 1e3, 1), stage_ms_per_step=stages)


def gemm_symbol(r):
    """The C++ kernel symbol a profiled GEMM launch ran on, rebuilt from the library's own record of the launch (tile instance,
    epilogue variant) - the names `rocprofv3 --stats` lists."""
    tf = lambda b: "true" if b else "false"
    ak, bkc, epi = not r['transA'], bool(r['transB']), r['epi']
    if r.get('b16'):      # bf16-resident kernels (csrc/gemm_b16.hip)
        bm, bn, wm, wn = {0: (128, 128, 2, 2), 1: (256, 128, 4, 2), 2: (256, 128, 2, 2), 3: (256, 64, 4, 1), 4: (256, 32, 4, 1)}[r['tile']]
        return "void gemm_b16_kernel<%d, %d, %d, %d, %s, %s, %d, %s>(B16Params)" % (bm, bn, wm, wn, tf(ak), tf(bkc), epi, tf(r['out_f32'] or epi == 6))
    if r.get('b1w'):      # ... its NT forms on the 64-byte-source-piece kernel (round 6)
        return "void gemm_b1w_kernel<%d>(P3Params)" % epi
    if r.get('b1'):       # bf16-resident operands on the LDS-DMA core (csrc/gemm_p3.hip, gemm_b1_kernel)
        return "void gemm_b1_kernel<%s, %d>(P3Params)" % (tf(bool(r['transA'])), epi)
    if r.get('dmf'):      # scorer layer-1 dgrad fused with the cand (.) pred backward (csrc/dm_fused.hip)
        return "void k_dm_mulpred_fused<%d>(DmfParams)" % (2 if r.get('bf16') else (3 if r.get('f16p') else (1 if r.get('h2out') else 0)))
    if r.get('h2w'):      # ... NT form on the 64-byte-source-piece kernel (round 5)
        return "void gemm_h2w_kernel<%d, %s>(H2Params)" % (epi, tf(bool(r.get('h2blk'))))
    if r.get('h2'):       # plane products over operands stored as two fp16 planes + a power-of-two scale (csrc/gemm_h2.hip)
        return "void gemm_h2_kernel<%s, %d>(H2Params)" % (tf(bool(r['transA'])), epi)
    if r.get('p3'):       # plane products over operands that already are three bf16 planes in HBM (csrc/gemm_p3.hip)
        return "void gemm_p3_kernel<%s, %d>(P3Params)" % (tf(bool(r['transA'])), epi)
    if r.get('x3'):       # fp32 through three bf16 planes (csrc/gemm_x3.hip)
        bm, bn, wm, wn = {0: (128, 128, 2, 2), 1: (256, 128, 4, 2)}[r['tile']]
        rs = r['rowscale'] and ((epi == 1 and ak and not bkc) or (epi in (0, 6) and not ak and not bkc))
        return "void gemm_x3_kernel<%d, %d, %d, %d, %s, %s, %d, %s, %d>(GemmParams)" % (bm, bn, wm, wn, tf(ak), tf(bkc), epi, tf(rs), 2 if r.get('x2h') else 3)
    if r['tile'] == 5:    # small-output TN weight gradients: v_mfma_f32_32x32x2_f32 fed straight from global memory (csrc/gemm.hip, round 6)
        return "void gemm_tn_small_kernel<%d, %d>(GemmParams)" % (1 if r['M'] <= 32 else 2, 1 if r['N'] <= 32 else 2)
    bm, bn, wm, wn = {0: (128, 128, 2, 2), 1: (256, 128, 4, 2), 2: (256, 256, 4, 2), 3: (256, 64, 4, 1), 4: (256, 32, 4, 1)}[r['tile'] % 8]
    rs = r['rowscale'] and ((epi == 1 and ak and not bkc) or (epi in (0, 6) and not ak and not bkc))
    if r['bf16']:
        return "void gemm_bf16_kernel<%d, %d, %d, %d, 32, %s, %s, %d, %s>(GemmParams)" % (bm, bn, wm, wn, tf(ak), tf(bkc), epi, tf(rs))
    return "void gemm_f32_kernel<%d, %d, %d, %d, 16, %s, %s, %d, %s>(GemmParams)" % (bm, bn, wm, wn, tf(ak), tf(bkc), epi, tf(rs))


def child_arm(args, extra):
    """This script as a child process for another configuration; returns its headline, its G1-like-lengths leg and its roofline entry."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(10, min(args.steps, 20))), "--warmup", str(args.warmup),
           "--seed", str(args.seed), "--no-cpu-baseline", "--no-boundary-leg", "--no-native-arm", "--no-arms", "--graph", "off"] + extra
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr.strip()[-400:])}
        d = json.loads(lines[-1])
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}
    rf = d["roofline"]
    arm = {"command": " ".join(["python", "bench.py"] + cmd[2:]), "workload": d["config"]["workload"], "dtype": d["dtype"], "gemm": d["config"]["gemm"],
           "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
           "sessions_per_gpu_per_step": d["config"]["sessions_per_gpu_per_step"],
           "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_ms",
                                               "algorithmic_gflop_per_launch", "algorithmic_bytes_per_launch", "all_gemm_ms_per_step")},
           "top_gemms": [{k: g.get(k) for k in ("kernel", "avg_launch_ms", "ms_per_step", "tflops", "frac_of_mfma_peak", "mfma_peak_tflops",
                                                 "algorithmic_GBps", "frac_of_hbm_peak")} for g in rf.get("top_gemms", [])]}
    if "g1_like_session_lengths" in d:
        arm["g1_like_session_lengths"] = {k: d["g1_like_session_lengths"][k] for k in ("value", "unit", "ms_per_step")}
    return arm


def dp_self_exchange(rt):
    """The gradient exchange of the data-parallel step on RCCL with the ONE GPU of a --gpus 1 run: a process group of one rank on the
    "nccl" backend, the flat gradient buffer all-reduced in the step's two buckets ([Wf1 .. Ws4] first, the rest after it), HIP-event
    timed on the stream the collectives run on.  Not a scaling number (one rank exchanges with itself): it shows RCCL loaded,
    stream-ordered and what the fixed cost of the two calls is.  tests/test_dp_rccl_gpu.py drives every exchange mode this way."""
    import torch
    import torch.distributed as dist
    own = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1)
            own = True
        L = rt.layout
        a, b = L.entries['Wf1'].offset, L.entries['Ws4'].offset + L.entries['Ws4'].size
        g = rt.grads.clone()
        ref = g.clone()
        def exchange():
            dist.all_reduce(g[a:b]); dist.all_reduce(g[:a]); dist.all_reduce(g[b:])
        for _ in range(3):
            exchange()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            exchange()
        e1.record(); torch.cuda.synchronize()
        out = {"backend": dist.get_backend(), "world": 1, "bytes": int(g.numel() * 4), "buckets": [int((b - a) * 4), int((g.numel() - (b - a)) * 4)],
               "ms": round(e0.elapsed_time(e1) / n, 4), "unchanged": bool(torch.equal(g, ref))}
    except Exception as ex:          # (never fail the bench line over this leg)
        out = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if own and dist.is_initialized():
        dist.destroy_process_group()
    return out


EPI_NAMES = {0: "plain", 1: "bias+leaky", 2: "bias+tanh", 3: "x leaky'", 4: "x tanh'", 5: "bias", 6: "split-K partials (+ reduce kernel)",
             10: "plain -> bf16", 12: "bias+tanh -> bf16", 13: "x leaky' -> bf16"}


def through_boundary(cfg, length_dist, warm_steps=50, timed_steps=200, seed=42, state="device", gemm_dtype="f32"):
    """SURVEY 8d's metric: sessions/sec through the drop-in boundary - GZIP TFRecord session files -> input_fn
    (datasets.prepare_dataset_iterator: C++ codec + prefetch) -> Estimator.train -> nar_module_model_fn -> hooks, host buffers
    handed over every step (H2D copies included), >= 50 warm-up steps excluded, steady state over >= 200 steps (the checkpoint
    Estimator.train writes when it returns is outside the clock)."""
    import shutil
    import tempfile
    import torch
    from chameleon_recsys_amd.nar import datasets, nar_trainer_gcom as T, synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, DeviceClickedItemsState
    from chameleon_recsys_amd.nar.estimator import SessionRunHook
    B = cfg['batch']
    d = tempfile.mkdtemp(prefix="cham_bench_")
    try:
        t0 = time.time()
        T, est, files, scfg = boundary_setup(cfg, d, warm_steps + timed_steps, length_dist, seed, state, gemm_dtype)
        gen_s = time.time() - t0

        class Clock(SessionRunHook):
            n, t0, t1, n1, p0, p1 = 0, None, None, 0, 0, 0

            @staticmethod
            def plans():
                return getattr(est._store.get('runtime'), 'plans_created', 0)

            def after_run(self, ctx, values):
                self.n += 1
                if self.n == 1:       # warm-up: the buffer sets of every padded length the files can produce (ragged sessions: T = batch maximum - 1)
                    rt_ = est._store.get('runtime')
                    if rt_ is not None and length_dist != "full":
                        rt_.warm_plans(B, range(1, cfg['seq_len']), cfg['neg'], cfg['neg_from_buffer'])
                if self.n == warm_steps:
                    torch.cuda.synchronize(); self.t0 = time.perf_counter(); self.p0 = self.plans()

            def end(self, session=None):
                torch.cuda.synchronize(); self.t1, self.n1, self.p1 = time.perf_counter(), self.n, self.plans()
        clock = Clock()
        est.train(lambda: datasets.prepare_dataset_iterator(files, scfg, batch_size=B, truncate_session_length=cfg['seq_len']),
                  hooks=[clock])
        steps = clock.n1 - warm_steps
        dt = clock.t1 - clock.t0
        # the input pipeline alone (decode + batch, no model)
        t2 = time.perf_counter()
        n_in = sum(len(f['session_id']) for f, _ in datasets.SessionDataset(files, scfg, batch_size=B, truncate_sequence_length=cfg['seq_len']))
        dt_in = time.perf_counter() - t2
        return dict(value=round(steps * B / dt, 1), unit="sessions/s", ms_per_step=round(dt / steps * 1e3, 3), steps=steps, warmup=warm_steps,
                    session_lengths=length_dist, clicked_items_state=state,
                    path="GZIP TFRecord files -> prepare_dataset_iterator (C++ decode) -> Estimator.train -> model_fn + hooks; host "
                         "batches handed over every step (H2D included)",
                    # buffer sets (StepPlan) built INSIDE the timed steps: a padded length T seen for the first time after the warm-up allocates
                    # and zero-fills several GB - tens of ms in a 0.5 s window (a training run amortises it over an epoch; this window does not)
                    step_plans_created_in_timed_steps=clock.p1 - clock.p0,
                    input_pipeline_alone_sessions_per_s=round(n_in / dt_in, 1), dataset_generation_s=round(gen_s, 1))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def stress_arm(n_items=5_000_000, batch=4096, neg=200, micro=2048, steps=2):
    """BASELINE.json configs[4] AS WRITTEN (large-catalog stress: 5 M articles, 128-d ACE, batch 4096, 200 negatives, item-embedding width
    floor(8 n^0.25) = 378 -> a 7.56 GB table, 1.89 G parameters under TF-dense Adam) as a child process (scripts/stress_large_catalog.py:
    micro-batched optimizer step - two micro-batches of 2048 sessions, 176 GB of HBM: 0.468 s/step against 0.523 with eight of 512,
    profiles/r05_notes.md - the HBM rooflines of the gather, TF-dense Adam and embedding-gradient kernels): ~8 s of set-up + 0.5 s per
    step.  `stress_arm_1m` in the line is the same script at 1 M articles x 1024 sessions (rounds 3-4 reported only that one)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "stress_large_catalog.py"), "--n-items", str(n_items), "--batch", str(batch),
           "--neg", str(neg), "--micro", str(micro), "--steps", str(steps)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr.strip()[-400:])}
        d = json.loads(lines[-1])
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}
    keep = ("workload", "n_items", "ace_dim", "batch", "negatives", "micro_batch_sessions", "params", "item_embedding_dim", "setup_s", "step_s",
            "sessions_per_s", "hbm_gb_allocated", "item_rows_per_micro_batch")
    arm = {k: d.get(k) for k in keep}
    arm["command"] = " ".join(["python"] + [os.path.relpath(c, ROOT) if c.endswith(".py") else c for c in cmd[1:]])
    for k in ("adam", "gather_lds_tiles", "embedding_gradient"):
        arm[k] = {q: d[k].get(q) for q in ("bound", "algorithmic_bytes", "ms", "achieved_gbs", "peak_gbs", "frac")}
    return arm


def pmc_traffic_live(symbol, seed):
    """HBM bytes per launch of kernel `symbol`, MEASURED IN THIS RUN: two rocprofv3 counter passes (`--kernel-trace --pmc FETCH_SIZE`, then
    `--pmc WRITE_SIZE`: the two do not fit one pass, MI355X_MICROARCH.md "HBM") over a short child run of this script on the same
    workload, traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB; the x 2 is the guide's gfx950 correction - FETCH_SIZE tallies 128-byte read
    requests at 64 bytes), averaged over the symbol's dispatches.  -> (bytes or None, how it was obtained / why it failed)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="cham_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "2", "--warmup", "1", "--seed", str(seed), "--no-cpu-baseline",
             "--no-ragged-leg", "--no-boundary-leg", "--no-arms", "--no-native-arm", "--no-pmc", "--graph", "off"]
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_d = os.path.join(d, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out_d, "-o", "pmc", "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
            files = glob.glob(os.path.join(out_d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s pass failed (rc %d): %s" % (ctr, r.returncode, (r.stderr or "").strip()[-200:])
            tot, n = 0.0, 0
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    if row.get('Counter_Name') == ctr and row.get('Kernel_Name') == symbol:
                        tot += float(row['Counter_Value']); n += 1
            if n == 0:
                return None, "kernel symbol %r is not in the %s counter output" % (symbol, ctr)
            vals[ctr] = (tot / n, n)
    except Exception as ex:          # (time-out, unreadable output: the caller falls back to the committed profile and says so)
        return None, "%s: %s" % (type(ex).__name__, ex)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    traffic = int(round((2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024))
    return traffic, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py %s`, "
                     "2 x FETCH_SIZE + WRITE_SIZE averaged over %d dispatches of the symbol" % (" ".join(child[2:]), vals["FETCH_SIZE"][1]))


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


def hip_graph_arm(model, state, dev_batches, one_step, first, Bg, gstep, n=20):
    """The same step replayed from ONE captured hipGraph (nar_model.GraphedTrainStep): ms/step over `n` replays and the host's submission
    cost per step, beside the eager numbers of the same process.  Bit-identical training (tests/test_graph_step_gpu.py), so the eager
    legs that follow continue the same trajectory."""
    import torch
    from chameleon_recsys_amd.nar.nar_model import GraphedTrainStep
    nd = len(dev_batches)
    gs = GraphedTrainStep(model, state)
    why = gs.supports(dev_batches[first % nd])
    if why is not None:
        return {"used": False, "reason": why}
    try:
        torch.cuda.synchronize()
        gstep[0] = gs
        for i in range(4):                      # capture + first replays
            one_step(first + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            one_step(first + 4 + i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = {"used": True, "steps": n, "ms_per_step": round(dt / n * 1e3, 3), "value": round(Bg * n / dt, 2), "unit": "sessions/s",
               "host_submit_ms_per_step": round(t_host / n * 1e3, 3),
               "what": "forward + backward + TF-Adam + state update + next batch's negatives as ONE hipGraph; per step the host submits "
                       "[batch -> input slot (one D2D copy), cham_step_scalars_set, graph launch]"}
    except Exception as ex:                     # (a failed capture has executed nothing: the eager step carries on)
        out = {"used": False, "reason": "capture failed: %s: %s" % (type(ex).__name__, str(ex)[:300])}
    gstep[0] = None
    return out


XGMI_LINKS, XGMI_LINK_GBPS = 7, 153.0          # MI355X: 7 point-to-point links x ~153 GB/s per GPU (MI355X_MICROARCH.md)


def dp_exchange_report(dp, one_step, first_step, backend, n=5):
    """The "dp" object of an N > 1 line: proof of what the process group is - backend, world, the PCI bus id and name of
    every rank's device (all-gathered) - and what one step's gradient exchange moved and cost: payload bytes handed to the collectives, the
    ring-equivalent bus bytes (all-reduce: 2 (N - 1) / N x payload), HIP-event time around the synchronous part of the exchange over `n`
    extra steps (max over ranks of the per-rank mean; the early bucket's overlapped flight is not in it), and that as a fraction of the
    7 x 153 GB/s of xGMI a GPU has.  Collective: every rank calls it."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    prop = torch.cuda.get_device_properties(torch.cuda.current_device())
    ident = "%04x:%02x:%02x %s" % (getattr(prop, 'pci_domain_id', 0), getattr(prop, 'pci_bus_id', 0), getattr(prop, 'pci_device_id', 0), prop.name)
    devices = [None] * world
    dist.all_gather_object(devices, ident)
    dp.time_exchange = True
    for i in range(n):
        one_step(first_step + i)
    ms = dp.exchange_ms()
    dp.time_exchange = False
    mean_ms = sum(ms) / max(1, len(ms))
    t = torch.tensor([mean_ms], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mean_ms = float(t.item())
    payload = int(dp.last_exchange_bytes)
    bus = payload * 2.0 * (world - 1) / world
    return {"backend": "rccl" if backend == "nccl" else backend, "world": world, "devices": devices,
            "distinct_devices": len(set(devices)), "mode": dp.mode, "grad_dtype": "bf16" if dp.comm_bf16 else "f32",
            "early_bucket": dp._early is not None, "collective_calls_total": dict(dp.collective_calls),
            "exchange_bytes_per_step": payload, "ring_bus_bytes_per_step": int(bus), "exchange_ms": round(mean_ms, 4), "exchange_steps_timed": len(ms),
            "exchange_GBps": round(bus / (mean_ms * 1e-3) / 1e9, 2) if mean_ms > 0 else None,
            "frac_of_xgmi": round(bus / (mean_ms * 1e-3) / 1e9 / (XGMI_LINKS * XGMI_LINK_GBPS), 4) if mean_ms > 0 else None,
            "xgmi_peak_GBps": XGMI_LINKS * XGMI_LINK_GBPS,
            "note": "exchange_ms = HIP events on the step's stream around the synchronous collectives (max over ranks of the mean); "
                    "the early bucket [Wf1 .. Ws4] flies under the backward and only its wait is inside"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="g1", choices=["g1", "tiny", "adressa"],
                    help="g1 = BASELINE.json configs[1] (the headline); adressa = configs[3] (2-layer GRU, 100 negatives); tiny = configs[0]")
    ap.add_argument("--length-dist", default="full", choices=["full", "g1"],
                    help="full: every session has seq_len clicks (no padded rows); g1: G1-like ragged lengths")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32_native", "bf16"],
                    help="f32: exact fp32 MFMA (BASELINE configs[1], the headline); bf16: bf16-rounded GEMM operands, fp32 accumulate/"
                         "storage/softmax/loss/Adam (BASELINE configs[2] arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary-leg", action="store_true",
                    help="skip the through-the-Estimator-boundary leg (TFRecord files -> input_fn -> Estimator.train, 50 + 200 steps)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: 256 sessions per GPU per step (global batch 256 x N); strong: global batch 256 sharded over the N GPUs "
                         "(SURVEY 8e / BASELINE configs[2]: 32 rows per GPU at N = 8)")
    ap.add_argument("--no-native-arm", action="store_true",
                    help="skip the comparison leg of --dtype f32: the same steps with every GEMM on the native fp32 MFMA")
    ap.add_argument("--no-ragged-leg", action="store_true",
                    help="skip the secondary G1-like-session-lengths leg (profiling runs: keeps per-symbol averages to the headline leg)")
    ap.add_argument("--no-arms", action="store_true",
                    help="skip the bf16_arm (BASELINE configs[2] arithmetic) and adressa_arm (configs[3]) legs of the default g1 / f32 run: "
                         "each is this script run as a child process with its own roofline entry")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 counter passes (child runs of this script) that measure roofline.traffic in this run")
    ap.add_argument("--state", default="device", choices=["device", "host"],
                    help="recent-clicks state: device-resident (csrc/state.hip) or the host numpy class fed every step")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--graph", default="arm", choices=["arm", "on", "off"],
                    help="the step replayed from ONE captured hipGraph (nar_model.GraphedTrainStep; bit-identical to the eager step).  arm (default): "
                         "the timed steps are eager - measured faster on ROCm 7.2: the graph's nodes overlap less than the hand-scheduled lanes, "
                         "profiles/r06_notes.md section 3 - and the replay is timed as an extra leg (hip_graph_arm); on: the timed steps ARE replays")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: until it is printed, file descriptor 1 points at stderr, so that nothing a library
    # writes at C level (librccl prints a version banner through stdio when a communicator is created) can end up next to it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
    if args.scaling == "strong":
        if cfg['batch'] % world:
            raise SystemExit("--scaling strong: global batch %d is not divisible by %d ranks" % (cfg['batch'], world))
        Bg, Bl = cfg['batch'], cfg['batch'] // world
    else:
        Bl = cfg['batch']             # per-GPU batch (weak scaling)
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

    graph = {"requested": args.graph, "used": False, "reason": None}
    gstep = [None]

    def one_step(i):
        k = i % n_distinct
        if gstep[0] is not None:
            gstep[0].step(dev_batches[k], dev_batches[(k + 1) % n_distinct])
            return
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
    # ---- the timed steps as replays of ONE captured hipGraph (the reference: one session.run per step) when the step is capturable
    if args.graph == "on":
        from chameleon_recsys_amd.nar.nar_model import GraphedTrainStep
        gs = GraphedTrainStep(model, state)
        k0 = args.warmup % n_distinct
        why = ("%d ranks (collectives inside the step)" % world) if world > 1 else (gs.supports(dev_batches[k0]) if args.warmup > 0 else "no eager warm-up step")
        if why is None:
            try:
                torch.cuda.synchronize()
                gstep[0] = gs
                for i in range(3):               # capture + first replays, untimed (n_distinct more warm-up steps keep the timed loop on the same batches)
                    one_step(args.warmup + i)
                for i in range(3, n_distinct):
                    one_step(args.warmup + i)
                torch.cuda.synchronize()
                graph.update(used=True, replays_before_the_clock=gs.replays)
            except Exception as ex:          # (a failed capture has executed nothing: the eager step carries on)
                gstep[0], why = None, "capture failed: %s: %s" % (type(ex).__name__, str(ex)[:300])
        if why is not None:
            graph["reason"] = why
            if args.graph == "on":
                raise SystemExit("--graph on: " + why)
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

    # ---- host-side cost of submitting a step (Python + ctypes + HIP launches, no synchronisation): what bounds the step once the GPU
    # is faster than the host (bf16 / short ragged batches)
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for i in range(3):
        one_step(args.warmup + args.steps + 100 + i)
    host_enqueue_ms = (time.perf_counter() - t_h) / 3 * 1e3
    torch.cuda.synchronize()
    host_enqueue_eager_ms = host_enqueue_ms
    if args.graph == "arm" and world == 1 and args.state == "device":
        graph["arm"] = hip_graph_arm(model, state, dev_batches, one_step, args.warmup + args.steps + 3, Bg, gstep)
    if gstep[0] is not None:         # the legs below (per-launch HIP events, arms) drive the EAGER step; its host cost beside the replay's
        graph["graph_replays_timed"] = gstep[0].replays
        gstep[0] = None
        t_h = time.perf_counter()
        for i in range(3):
            one_step(args.warmup + args.steps + 103 + i)
        host_enqueue_eager_ms = (time.perf_counter() - t_h) / 3 * 1e3
        torch.cuda.synchronize()

    # ---- roofline leg: HIP-event timing of every GEMM launch over a few extra steps -------------------------
    rt.profile = []
    nprof = 3
    for i in range(nprof):
        one_step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    prof, rt.profile = rt.profile, None
    # Per kernel SYMBOL (rebuilt from the library's record of each launch: tile instance + epilogue variant), aggregated over all of
    # its launches exactly like `rocprofv3 --stats` aggregates, so avg_launch_ms is comparable with the committed kernel_stats.  The
    # roofline object describes the symbol with the LARGEST total time; the top three are listed beside it.
    by_sym = {}
    for r in prof:
        if r['tile'] < 0:
            continue
        e = by_sym.setdefault(gemm_symbol(r), dict(n=0, ms=0.0, flop=0.0, bytes=0.0, r=r, shapes={}))
        if r['M'] * r['N'] * r['K'] > e['r']['M'] * e['r']['N'] * e['r']['K']:
            e['r'] = r
        t_ms = r['ev'][0].elapsed_time(r['ev'][1])
        e['n'] += 1
        e['ms'] += t_ms
        e['flop'] += 2.0 * r['M'] * r['N'] * r['K']
        sh = e['shapes'].setdefault((r['M'], r['N'], r['K']), [0, 0.0])
        sh[0] += 1; sh[1] += t_ms
        # operands + output, each touched once: A + B + bias + C (+ the saved activation a dgrad epilogue reads); fp32 storage, or
        # bf16 operands / saved activations and a bf16 or fp32 output for the bf16-resident kernels
        if r.get('dmf') and r.get('bf16'):      # the bf16 twin: every matrix one bf16 array
            e['bytes'] += 2.0 * r['M'] * r['K'] + 2.0 * r['K'] * r['N'] + 2.0 * r['M'] * r['N'] + 2.0 * r['M'] * r['N']
        elif r.get('dmf'):      # dS1 in, Z2c in, the planes out (6 B per element as three bf16 planes, 4 B as two fp16 planes), pred / dpred / b2 partials per position
            e['bytes'] += 4.0 * r['M'] * r['K'] + 6.0 * r['K'] * r['N'] + 4.0 * r['M'] * r['N'] + (4.0 if r.get('h2out') else 6.0) * r['M'] * r['N']
        elif r.get('h2'):       # operands as two fp16 planes (4 B per element), fp32 output, the dgrad reads the activation's h plane
            e['bytes'] += 4.0 * (r['M'] * r['K'] + r['K'] * r['N']) + 4.0 * r['M'] * r['N'] + (2.0 * r['M'] * r['N'] if r['dref'] else 0) + \
                (4.0 * r['N'] if r['bias'] else 0)
        elif r.get('p3'):       # operands as three bf16 planes (6 B per element), fp32 output, the dgrad reads the activation's h plane
            e['bytes'] += 6.0 * (r['M'] * r['K'] + r['K'] * r['N']) + 4.0 * r['M'] * r['N'] + (2.0 * r['M'] * r['N'] if r['dref'] else 0) + \
                (4.0 * r['N'] if r['bias'] else 0)
        elif r.get('b16') or r.get('b1'):
            e['bytes'] += 2.0 * (r['M'] * r['K'] + r['K'] * r['N'] + (r['M'] * r['N'] if r['dref'] else 0)) + \
                (4.0 if (r['out_f32'] or r['epi'] == 6) else 2.0) * r['M'] * r['N'] + (4.0 * r['N'] if r['bias'] else 0)
        else:
            e['bytes'] += 4.0 * (r['M'] * r['K'] + r['K'] * r['N'] + r['M'] * r['N'] * (2 if r['dref'] else 1) + (r['N'] if r['bias'] else 0))
    ranked = sorted(by_sym.items(), key=lambda kv: -kv[1]['ms'])
    ms_all = sum(e['ms'] for e in by_sym.values())
    fl_all = sum(e['flop'] for e in by_sym.values())

    def describe(sym, e):
        r = e['r']
        mode = "NN" if not r['transA'] and not r['transB'] else ("TN (wgrad)" if r['transA'] else
                                                                 ("NT (dgrad)" if r['dref'] or not r['bias'] else "NT (forward, transposed weight shadow)"))
        return "%s = %s MFMA GEMM, %s, %s%s; M,N,K of its largest launch %d,%d,%d" % (
            sym, ("bf16-resident, LDS-DMA staging" if r.get('b1') else "bf16-resident" if r.get('b16') else "bf16 (fp32 storage, rounded while staged)") if r['bf16'] else
            ("fp32-grade (operands resident in HBM as two fp16 planes x a power-of-two scale written by their producers, three fp16 MFMA "
             "products per fp32 product, fp32 accumulate, LDS-DMA staging)" if r.get('h2') else
             "fp32-grade (operands resident in HBM as three bf16 planes written by their producers, six bf16 MFMA products per fp32 product, "
             "fp32 accumulate, LDS-DMA staging)" if r.get('p3') else
             "fp32-grade (two fp16 planes x a power-of-two scale per operand, split while staged, three fp16 MFMA products per fp32 product, "
             "fp32 accumulate)" if r.get('x2h') else
             "fp32 (three bf16 planes per operand, six bf16 MFMA products per fp32 product, fp32 accumulate)" if r.get('x3') else "fp32"), mode, EPI_NAMES.get(r['epi'], "?"), ", row-scale prologue" if r['rowscale'] else "",
            r['M'], r['N'], r['K'])

    def gemm_entry(sym, e):
        tf_s = e['flop'] / (e['ms'] * 1e-3) / 1e12
        gbs = e['bytes'] / (e['ms'] * 1e-3) / 1e9
        peak = BF16_MATRIX_PEAK_TFLOPS if e['r']['bf16'] else (H2_MATRIX_PEAK_TFLOPS if (e['r'].get('h2') or e['r'].get('x2h') or e['r'].get('f16p')) else
                                                               X3_MATRIX_PEAK_TFLOPS if (e['r'].get('x3') or e['r'].get('p3') or e['r'].get('dmf')) else FP32_MATRIX_PEAK_TFLOPS)
        return {"kernel": describe(sym, e), "launches_per_step": round(e['n'] / nprof, 2), "avg_launch_ms": round(e['ms'] / e['n'], 4),
                "ms_per_step": round(e['ms'] / nprof, 3), "tflops": round(tf_s, 2), "frac_of_mfma_peak": round(tf_s / peak, 4),
                "mfma_peak_tflops": round(peak, 1), "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBPS, 4),
                # one symbol can serve launches of very different shapes (e.g. the CAR dgrad and a K = 64 scorer dgrad): per shape
                "by_shape": [{"M": m, "N": n_, "K": k, "launches_per_step": round(c / nprof, 2), "avg_launch_ms": round(t / c, 4),
                              "tflops": round(2.0 * m * n_ * k / (t / c * 1e-3) / 1e12, 2),
                              "frac_of_mfma_peak": round(2.0 * m * n_ * k / (t / c * 1e-3) / 1e12 / peak, 4)}
                             for (m, n_, k), (c, t) in sorted(e['shapes'].items(), key=lambda kv: -kv[1][1])[:3]]}
    DOM_SYMBOL, dom = ranked[0] if ranked else ("", dict(n=1, ms=1.0, flop=0.0, bytes=0.0, r=None))
    n_nn, ms_nn, fl_nn = dom['n'], dom['ms'], dom['flop']
    dom_h2 = bool(dom['r'] and dom['r'].get('h2'))
    dom_x3 = bool(dom['r'] and (dom['r'].get('x3') or dom['r'].get('p3')))
    dom_peak = H2_MATRIX_PEAK_TFLOPS if dom_h2 else (X3_MATRIX_PEAK_TFLOPS if dom_x3 else FP32_MATRIX_PEAK_TFLOPS)
    achieved = fl_nn / (ms_nn * 1e-3) / 1e12 if ms_nn > 0 else 0.0
    # roofline.traffic: measured in THIS run by two rocprofv3 counter passes over a short child run (pmc_traffic_live) on the default line;
    # otherwise (child arms, --no-pmc, no rocprofv3) the newest committed profiles/*_pmc_traffic.json, LABELLED as such - and a symbol that
    # is missing from it is reported in `traffic_error`, never silently null
    traffic, traffic_src, traffic_err = None, None, None
    headline = args.config == "g1" and world == 1 and args.scaling == "weak" and args.dtype == "f32"
    if rank == 0 and headline and not args.no_pmc and not args.no_arms:
        traffic, traffic_src = pmc_traffic_live(DOM_SYMBOL, args.seed)
        if traffic is None:
            traffic_err, traffic_src = traffic_src, None
    if traffic is None and headline:
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            if fn.endswith("_pmc_traffic.json"):
                pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
                rec = pmc.get(DOM_SYMBOL)
                if rec:
                    traffic = rec["traffic_bytes_per_launch"]
                    traffic_src = "committed profile, NOT measured in this run: profiles/%s (%s)" % (fn, pmc.get("_generated_by", "scripts/pmc_traffic.py"))
                else:
                    traffic_err = ((traffic_err + "; ") if traffic_err else "") + "kernel symbol %r is not in profiles/%s" % (DOM_SYMBOL, fn)
                    print("bench.py: roofline.traffic unavailable - %s" % traffic_err, file=sys.stderr)
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

    # ---- comparison leg (default fp32 arithmetic only): the SAME runtime, weights and batches with the plane-product GEMMs switched off,
    # i.e. every GEMM on v_mfma_f32_32x32x2_f32 - so the line carries both arms measured in one run
    native_arm = None
    if args.dtype == "f32" and getattr(rt, 'x3', False) and not args.no_native_arm:
        had_p3, rt.x3, rt.p3 = rt.p3, False, False       # p3 off as well: the plane-resident CAR GEMMs are plane products too
        n_arm = max(5, min(args.steps, 10))
        for i in range(3):
            one_step(args.warmup + args.steps + 200 + i)
        barrier()
        t0 = time.perf_counter()
        for i in range(n_arm):
            one_step(args.warmup + args.steps + 203 + i)
        barrier()
        adt = time.perf_counter() - t0
        rt.x3, rt.p3 = True, had_p3
        if world > 1:
            tt = torch.tensor([adt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            adt = float(tt.item())
        native_arm = {"gemm": "every GEMM on v_mfma_f32_32x32x2_f32 (--dtype f32_native), same runtime / weights / batches", "steps": n_arm,
                      "value": round(Bg * n_arm / adt, 2), "unit": "sessions/s", "ms_per_step": round(adt / n_arm * 1e3, 3)}

    # ---- N > 1: what the exchange is and what it cost (every rank takes part: collectives - never under `if rank == 0`)
    dp_info = dp_exchange_report(dp, one_step, args.warmup + args.steps + 200, backend) if world > 1 else None

    if rank == 0:
        L = rt.layout
        T = cfg['seq_len'] - 1
        dense_fwd = dense_step_flops(Bl, T, cfg['neg'], L.F, cfg['C'], cfg['H'], cfg.get('rnn_num_layers', 1))
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "NAR training sessions/sec", "value": round(Bg * args.steps / dt, 2), "unit": "sessions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": {"g1": "G1-shape synthetic (BASELINE.json configs[1])", "tiny": "G1-tiny synthetic (configs[0])",
                                    "adressa": "Adressa-shape synthetic (BASELINE.json configs[3])"}[args.config],
                       "n_items": cfg['n_items'], "ace_dim": cfg['ace_dim'], "seq_len": cfg['seq_len'],
                       "sessions_per_gpu_per_step": Bl, "global_batch": Bg, "negatives": cfg['neg'],
                       "CAR_embedding_size": cfg['C'], "rnn_units": cfg['H'], "rnn_cell": cfg.get('rnn_cell', 'ugrnn'), "rnn_layers": cfg.get('rnn_num_layers', 1),
                       "session_lengths": args.length_dist, "parallelism": "dp%d" % world, "clicked_items_state": args.state,
                       "gemm": {"f32": ("fp32 accumulate / epilogues, fp32-grade error (float64-error bar next to the native fp32 MFMA: tests/test_gemm_h2_gpu.py, "
                                        "tests/test_gemm_x3_gpu.py): the three candidate-row CAR GEMMs as THREE fp16-plane products per fp32 product over "
                                        "(h, l) fp16 planes x a device-derived power-of-two scale that their producers wrote to HBM (csrc/gemm_h2.hip, "
                                        "v_mfma_f32_32x32x16_f16; the NT forms stage 64-byte source pieces: gemm_h2w_kernel); the two backward kernels of the scorer's first layer - its weight gradient and the products inside its fused dgrad - "
                                        "as three fp16-plane products too, split while staged (cham_gemm_f32x2h = gemm_x3_kernel<..., 2>, k_dm_mulpred_fused<3>; "
                                        "tests/test_gemm_x2h_gpu.py, tests/test_dm_fused_gpu.py; CHAM_S1_H2); the other GEMMs - that layer's forward among them - "
                                        "with N > 64 as six bf16-plane products split while staged (csrc/gemm_x3.hip); N <= 64 on v_mfma_f32_32x32x2_f32.  Accuracy contract of the "
                                        "two-plane operands: relative to the MATRIX bound, not to each row - fp32 grade for rows within 2^-18 of the largest "
                                        "entry, an absolute error of 2^-40 of the bound below that (INTEGRATION.md)") if getattr(rt, 'h2', False) else
                                       ("fp32 accumulate / epilogues; GEMMs with N > 64 as six bf16-plane products per fp32 product on "
                                        "v_mfma_f32_32x32x16_bf16 (fp32-grade error: tests/test_gemm_x3_gpu.py, tests/test_gemm_p3_gpu.py) - the three candidate-row "
                                        "CAR GEMMs over planes their producers wrote to HBM (csrc/gemm_p3.hip), the others split while staged "
                                        "(csrc/gemm_x3.hip); N <= 64 on v_mfma_f32_32x32x2_f32"),
                                "f32_native": "every GEMM on v_mfma_f32_32x32x2_f32",
                                "bf16": "bf16-resident candidate-row matrices, fp32 accumulate"}[args.dtype],
                       "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
                       "host_enqueue_ms_per_step_eager": round(host_enqueue_eager_ms, 3),
                       "hip_graph": graph,
                       "rnn_coop_spin_timeouts": int(rt.rnn_coop_timed_out()),       # bounded spins of the cooperative recurrent kernels (ragged leg): must be 0
                       "final_loss": [round(float(x), 5) for x in loss]},
            "roofline": {"bound": "mfma", "kernel": describe(DOM_SYMBOL, dom) + " - the GEMM symbol with the largest total time in the step",
                         "achieved": round(achieved, 2), "peak": round(dom_peak, 1), "unit": "TFLOP/s",
                         "peak_note": ("fp16 dense MFMA peak 2500 TFLOP/s / 3 plane products per fp32 product; algorithmic fp32 FLOPs in `achieved` "
                                       "(the native fp32 MFMA peak is %.1f)" % FP32_MATRIX_PEAK_TFLOPS) if dom_h2 else
                                      ("bf16 dense MFMA peak 2500 TFLOP/s / 6 plane products per fp32 product; algorithmic fp32 FLOPs in `achieved` "
                                       "(the native fp32 MFMA peak is %.1f)" % FP32_MATRIX_PEAK_TFLOPS) if dom_x3 else "fp32 dense MFMA peak",
                         "frac": round(achieved / dom_peak, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_error": traffic_err,
                         "launches_per_step": round(n_nn / nprof, 2), "avg_launch_ms": round(ms_nn / max(1, n_nn), 4),
                         "algorithmic_gflop_per_launch": round(fl_nn / max(1, n_nn) / 1e9, 3),
                         "algorithmic_bytes_per_launch": round(dom['bytes'] / max(1, n_nn)),
                         "top_gemms": [gemm_entry(sym, e) for sym, e in ranked[:3]],
                         "all_gemm_ms_per_step": round(ms_all / nprof, 3),
                         "all_gemm_tflops": round(fl_all / (ms_all * 1e-3) / 1e12, 2) if ms_all > 0 else 0.0,
                         "step_reference_dense_tflops": round(3 * dense_fwd / (ms_step * 1e-3) / 1e12, 2)},
        }
        if args.dtype == "bf16":     # bf16 MFMA is 16x the fp32 rate: name whichever roof binds the dominant launch at its shape
            t_mfma, t_hbm = fl_nn / (BF16_MATRIX_PEAK_TFLOPS * 1e12), dom['bytes'] / (HBM_PEAK_GBPS * 1e9)
            if t_hbm > t_mfma:
                gbs = dom['bytes'] / (ms_nn * 1e-3) / 1e9 if ms_nn > 0 else 0.0
                out["roofline"].update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBPS, 4))
            else:
                out["roofline"].update(bound="mfma", peak=BF16_MATRIX_PEAK_TFLOPS, frac=round(achieved / BF16_MATRIX_PEAK_TFLOPS, 4))
            out["roofline"].update(traffic=None, traffic_source=None)
        if dp_info is not None:
            out["dp"] = dp_info
        if ragged is not None:
            out["g1_like_session_lengths"] = ragged
        if native_arm is not None:
            out["native_fp32_mfma_arm"] = native_arm
        if world == 1 and not args.no_boundary_leg and args.config == "g1" and args.dtype in ("f32", "f32_native"):
            # SURVEY 8d's metric proper (input pipeline + H2D + hooks inside the clock), same workload as the headline and its
            # G1-like-session-lengths variant
            out["through_boundary"] = through_boundary(cfg, args.length_dist, seed=args.seed, state=args.state, gemm_dtype=args.dtype)
            if ragged is not None:
                out["through_boundary_g1_like_session_lengths"] = through_boundary(cfg, "g1", seed=args.seed, state=args.state, gemm_dtype=args.dtype)
        if world == 1 and not args.no_boundary_leg:
            out["dp_self_exchange_ms"] = dp_self_exchange(rt)
        if world == 1 and not args.no_arms and args.config == "g1" and args.dtype == "f32" and args.length_dist == "full":
            # the other configurations BASELINE.json names, timed in the same invocation (children of this process, one after the
            # other, this process idle meanwhile): configs[2]'s arithmetic on one GPU, configs[3]'s shape
            out["bf16_arm"] = child_arm(args, ["--dtype", "bf16"])
            out["adressa_arm"] = child_arm(args, ["--config", "adressa", "--no-ragged-leg"])
            out["stress_arm"] = stress_arm()          # configs[4] at its full size: 5 M articles x 4096 sessions x 200 negatives
            out["stress_arm_1m"] = stress_arm(n_items=1_000_000, batch=1024, micro=512)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.length_dist, args.seed)
            out["accuracy_vs_cpu_ref"] = hitrate_parity(args.seed)
    # every rank drains its (redirected) stdout - the RCCL banner is C stdio - and meets at a barrier BEFORE rank 0 prints the one JSON line.
    # (Round 3 had this barrier inside the rank-0 block: the other ranks went straight to destroy_process_group and rank 0's barrier
    # failed with "connection closed by peer" - caught in round 4 by running the launch contract with 4 and 8 gloo ranks on one GPU,
    # profiles/r04_dp_proof_gloo_one_gpu.jsonl; tests/test_bench_bookkeeping.py now checks that no collective sits under `if rank == 0`.)
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if world > 1:
        dist.barrier()
    if rank == 0:
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
