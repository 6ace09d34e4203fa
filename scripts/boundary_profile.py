#!/usr/bin/env python
"""Where the HOST time of the through-the-boundary loop goes (bench.through_boundary with wall-clock accumulators around the pieces
of Estimator.train: input hand-over, hooks, upload, enqueue of forward / backward / Adam, look-ahead staging).  No cProfile (its
per-call cost doubles the numbers): perf_counter around a dozen methods.  python scripts/boundary_profile.py [g1|full]"""
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acc, cnt = collections.Counter(), collections.Counter()


def timed(cls, name, label=None):
    fn = getattr(cls, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[label] += time.perf_counter() - t
        cnt[label] += 1
        return r
    setattr(cls, name, w)


def main():
    import bench
    from chameleon_recsys_amd.nar import datasets, estimator, nar_model
    M, H = nar_model.NARModuleModel, nar_model.ItemsStateUpdaterHook
    for n in ("upload_batch", "_h2d", "presample", "forward", "backward", "apply_gradients", "train_step", "stage_next", "feed"):
        timed(M, n)
    timed(M, "_neg_sample")
    import torch
    timed(torch.cuda.Stream, "wait_event", "Stream.wait_event"); timed(torch.cuda.Event, "record", "Event.record")
    timed(torch.Tensor, "record_stream", "Tensor.record_stream"); timed(nar_model.NARRuntime, "plan", "rt.plan")
    timed(nar_model._PinnedRing, "begin", "ring.begin")
    timed(H, "before_run", "hook.before_run"); timed(H, "after_run", "hook.after_run")
    timed(datasets.SessionDataset, "advance", "dataset.advance"); timed(datasets.SessionDataset, "peek", "dataset.peek")
    dist = sys.argv[1] if len(sys.argv) > 1 else "g1"
    out = bench.through_boundary(bench.G1, dist, warm_steps=50, timed_steps=300)
    steps = cnt["train_step"]
    print(json.dumps(out))
    print("per step over %d steps (warm-up included), ms:" % steps)
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-24s %7.3f  (%d calls)" % (k, v / steps * 1e3, cnt[k]))


if __name__ == "__main__":
    main()
