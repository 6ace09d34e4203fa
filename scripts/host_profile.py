#!/usr/bin/env python
"""Where the HOST time of a step goes (cProfile over the G1-like-session-lengths leg of bench.py: short GPU steps, the host's enqueue
work is what bounds the through-the-boundary throughput).  python scripts/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from chameleon_recsys_amd.nar import synthetic
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    from chameleon_recsys_amd.nar.nar_model import ModeKeys, NARModuleModel, NARRuntime
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    p = synthetic.default_params(46000, 250, seq_len=20, batch_size=256, neg=50, neg_from_buffer=3000, buffer_size=20000, for_norm=2000, C=1024, H=255)
    rt = NARRuntime(p)
    model = NARModuleModel(ModeKeys.TRAIN, None, None, p['session_features_config'], p['articles_features_config'], 256, p['lr'], 1.0, 50, 3000,
                           p['content_article_embeddings_matrix'], softmax_temperature=0.1, reg_weight_decay=1e-5, recent_clicks_buffer_max_size=20000,
                           recent_clicks_for_normalization=2000, articles_metadata=p['articles_metadata'], CAR_embedding_size=1024, rnn_units=255, runtime=rt)
    batches = synthetic.make_batches(8, 256, 20, 46000, p['session_features_config'], length_dist='g1', sessions_per_hour=512)
    state = DeviceClickedItemsState(1.0, 20000, 2000, 46000)
    dev = [model.upload_batch(f, l) for f, l in batches]

    def step(i):
        k = i % 8
        model.feed_state(state, state)
        model.train_step(dev[k])
        state.update_from_device_batch(dev[k]['aci'], dev[k]['g_event_ts'])
        model.presample(dev[(k + 1) % 8])
    for i in range(16):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("resident G1-like leg: %.3f ms/step wall, host enqueue %.3f ms/step" % (t_all / steps * 1e3, t_enq / steps * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(steps):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    print("---- by own time (per %d steps)" % steps)
    st.print_stats(28)
    st.sort_stats("cumulative")
    print("---- by cumulative time")
    st.print_stats(22)


if __name__ == "__main__":
    main()
