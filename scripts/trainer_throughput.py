#!/usr/bin/env python
"""End-to-end rate of the drop-in boundary (host buffers handed over every step): GZIP TFRecord session files ->
input_fn (C++ codec, prefetch thread) -> Estimator.train -> nar_module_model_fn -> HIP step, G1 shape, 1 GPU.
This is the PCIe-inclusive number DESIGN.md quotes next to bench.py's HBM-resident one.

  python scripts/trainer_throughput.py [--hours 4] [--sessions-per-hour 5120] [--state device|host]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=int, default=4)
    ap.add_argument("--sessions-per-hour", type=int, default=5120)
    ap.add_argument("--state", default="device", choices=["device", "host"])
    ap.add_argument("--length-dist", default="full", choices=["full", "g1"], help="session lengths of the generated files")
    a = ap.parse_args()
    import torch
    from chameleon_recsys_amd.nar import datasets, nar_trainer_gcom as T, synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, DeviceClickedItemsState
    d = tempfile.mkdtemp(prefix="cham_g1_")
    t0 = time.time()
    files, csv, pkl = synthetic.write_dataset(d, a.hours + 1, a.sessions_per_hour, 46000, 250, seq_len=20, seed=42, length_dist=a.length_dist)
    gen_s = time.time() - t0
    argv = ['--batch_size', '256', '--truncate_session_length', '20', '--learning_rate', '1e-4', '--reg_l2', '1e-5',
            '--softmax_temperature', '0.1', '--recent_clicks_buffer_max_size', '20000', '--recent_clicks_for_normalization', '2000',
            '--eval_metrics_top_n', '5', '--CAR_embedding_size', '1024', '--rnn_units', '255', '--train_total_negative_samples', '50',
            '--train_negative_samples_from_buffer', '3000', '--eval_total_negative_samples', '50',
            '--eval_negative_samples_from_buffer', '3000', '--content_embedding_scale_factor', '6.0', '--disable_eval_benchmarks',
            '--model_dir', os.path.join(d, 'model'), '--clicked_items_state', a.state]
    T.FLAGS = T.define_flags().parse_args(argv)
    meta_df, ace = T.load_acr_module_resources(csv, pkl)
    ace = T.l2_normalize_rows(ace) * np.float32(6.0)
    acfg = T.get_articles_features_config(n_items=ace.shape[0])
    meta = T.process_articles_metadata(meta_df, acfg)
    scfg = T.get_session_features_config()
    T.eval_sessions_metrics_log = []
    T.clicked_items_state = (DeviceClickedItemsState if a.state == "device" else ClickedItemsState)(1.0, 20000, 2000, ace.shape[0])
    est = T.build_estimator(os.path.join(d, 'model'), ace, meta, acfg, scfg)
    est.config.log_step_count_steps = 0
    input_fn = lambda fs: (lambda: datasets.prepare_dataset_iterator(fs, scfg, batch_size=256, truncate_session_length=20))
    est.train(input_fn(files[:1]))                      # warm-up hour (allocations, first-batch statistics)
    torch.cuda.synchronize()
    s0 = est.global_step
    t1 = time.time()
    est.train(input_fn(files[1:a.hours + 1]))
    torch.cuda.synchronize()
    dt = time.time() - t1
    steps = est.global_step - s0
    # input pipeline alone
    t2 = time.time()
    n = sum(len(f['session_id']) for f, _ in datasets.SessionDataset(files[1:a.hours + 1], scfg, batch_size=256, truncate_sequence_length=20))
    dt_in = time.time() - t2
    print(json.dumps(dict(workload="G1-shape synthetic via the Estimator boundary (TFRecord files -> input_fn -> model_fn)",
                          clicked_items_state=a.state, session_lengths=a.length_dist, steps=steps, sessions=steps * 256, seconds=round(dt, 3),
                          sessions_per_s=round(steps * 256 / dt, 1), ms_per_step=round(dt / steps * 1e3, 3),
                          includes="GZIP+TFRecord+protobuf decode, H2D copies of every batch, hooks, checkpoint at the end",
                          input_pipeline_alone_sessions_per_s=round(n / dt_in, 1), dataset_generation_s=round(gen_s, 1))), flush=True)


if __name__ == "__main__":
    main()
