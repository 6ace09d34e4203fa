#!/usr/bin/env python
"""Input-pipeline throughput on the GPU box's host (SURVEY 8 f2 / VERDICT r01 #5): GZIP TFRecord session files -> SessionDataset
batches (inflate threads -> batcher -> decode workers), G1 shape, for several thread counts.  No GPU involved.

  python scripts/decode_throughput.py [--hours 8] [--sessions-per-hour 12800]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=int, default=8)
    ap.add_argument("--sessions-per-hour", type=int, default=12800)
    a = ap.parse_args()
    from chameleon_recsys_amd.nar import config, datasets, synthetic
    scfg = config.get_session_features_config_gcom(46000)
    out = dict(host_cores=len(os.sched_getaffinity(0)), files=a.hours, sessions_per_file=a.sessions_per_hour, results=[])
    for dist in ("full", "g1"):
        d = tempfile.mkdtemp(prefix="cham_decode_")
        files, _, _ = synthetic.write_dataset(d, a.hours, a.sessions_per_hour, 46000, 250, seq_len=20, seed=42, length_dist=dist)
        for workers, inflate in ((1, 1), (4, 2), (8, 4), (16, 4), (16, 8)):
            os.environ["CHAM_TFRECORD_THREADS"], os.environ["CHAM_TFRECORD_INFLATE_THREADS"] = str(workers), str(inflate)
            best = 0.0
            for _ in range(2):
                t0 = time.perf_counter()
                n = sum(len(f['session_id']) for f, _ in datasets.SessionDataset(files, scfg, batch_size=256, truncate_sequence_length=20))
                best = max(best, n / (time.perf_counter() - t0))
            out["results"].append(dict(session_lengths=dist, decode_threads=workers, inflate_threads=inflate, sessions_per_s=round(best)))
            print(out["results"][-1], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
