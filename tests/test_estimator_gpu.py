"""The drop-in boundary end to end on the GPU: TFRecord files -> input_fn -> Estimator.train / .evaluate through
``nar_module_model_fn`` and ``ItemsStateUpdaterHook`` - against the CPU oracle driven over the same files.
Tolerances: loss within 1e-3 per step, negative samples bit exact, HitRate@n / MRR@n from the same ranked lists."""
import numpy as np
import pytest
import torch

from chameleon_recsys_amd.nar import datasets, metrics, nar_trainer_gcom as T, synthetic
from chameleon_recsys_amd.nar.clicked_items_state import ClickedItemsState, batch_clicks_for_state
from chameleon_recsys_amd.nar.estimator import SessionRunArgs, SessionRunHook
from chameleon_recsys_amd.nar.nar_model import NARModuleModel

pytestmark = pytest.mark.gpu

ARGS = ['--batch_size', '24', '--truncate_session_length', '10', '--learning_rate', '1e-3', '--reg_l2', '1e-5',
        '--softmax_temperature', '0.2', '--recent_clicks_buffer_max_size', '600', '--recent_clicks_for_normalization', '100',
        '--eval_metrics_top_n', '3', '--CAR_embedding_size', '64', '--rnn_units', '40', '--train_total_negative_samples', '7',
        '--train_negative_samples_from_buffer', '50', '--eval_total_negative_samples', '12',
        '--eval_negative_samples_from_buffer', '60', '--content_embedding_scale_factor', '6.0',
        '--training_hours_for_each_eval', '2', '--disable_eval_benchmarks', '--save_eval_sessions_negative_samples']


class Capture(SessionRunHook):
    def __init__(self):
        self.losses, self.negs = [], []

    def before_run(self, ctx):
        return SessionRunArgs(fetches={'loss': ctx.model.loss_t, 'neg': ctx.model.batch_negative_items})

    def after_run(self, ctx, vals):
        self.losses.append(vals.results['loss'].copy()); self.negs.append(vals.results['neg'].copy())


def test_estimator_train_evaluate_matches_oracle(gpu, tmp_path):
    from oracle.nar_oracle import NAROracle
    files, csv, pkl = synthetic.write_dataset(str(tmp_path / "data"), 5, 60, 400, 24, seq_len=12, seed=21)
    argv = ARGS + ['--train_set_path_regex', str(tmp_path / "data" / "sessions_hour_*.tfrecord.gz"),
                   '--acr_module_articles_metadata_csv_path', csv, '--acr_module_articles_content_embeddings_pickle_path', pkl,
                   '--model_dir', str(tmp_path / "model")]
    T.FLAGS = T.define_flags().parse_args(argv)
    meta_df, ace = T.load_acr_module_resources(csv, pkl)
    ace = T.l2_normalize_rows(ace) * np.float32(6.0)
    assert np.allclose(np.linalg.norm(ace, axis=1), 6.0, atol=1e-4)
    acfg = T.get_articles_features_config(n_items=ace.shape[0])
    meta = T.process_articles_metadata(meta_df, acfg)
    scfg = T.get_session_features_config()
    T.eval_sessions_metrics_log = []
    T.sessions_negative_items_log = []
    T.clicked_items_state = ClickedItemsState(1.0, 600, 100, ace.shape[0])
    est = T.build_estimator(str(tmp_path / "model"), ace, meta, acfg, scfg)
    input_fn = lambda fs: (lambda: datasets.prepare_dataset_iterator(fs, scfg, batch_size=24, truncate_session_length=10))

    # ---- HIP path through the Estimator
    cap = Capture()
    est.train(input_fn(files[:2]), hooks=[cap])
    w_after_train = est._store['runtime'].logical_weights()
    res = est.evaluate(input_fn(files[2]))
    cap2 = Capture()
    est.train(input_fn(files[2:4]), hooks=[cap2])

    # ---- oracle over the same files
    p = dict(est.params); p['tf_random_seed'] = 42; p['rnn_num_layers'] = 1
    rt = est._store['runtime']
    orc = NAROracle(p, weights=rt.layout.unpack(rt.layout.pack(rt.layout.init_logical(42))))
    st = ClickedItemsState(1.0, 600, 100, ace.shape[0])

    def oracle_train(fs, cap):
        k = 0
        for f, l in datasets.SessionDataset(fs, scfg, batch_size=24, truncate_sequence_length=10):
            ref = orc.train_step(f, l, st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm())
            assert np.array_equal(cap.negs[k], ref['neg_items'].numpy()), k
            assert abs(cap.losses[k][0] - float(ref['total_loss'])) < 1e-3, (k, cap.losses[k], float(ref['total_loss']))
            st.update_items_state(*batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp']))
            k += 1
        assert k == len(cap.losses)
    oracle_train(files[:2], cap)
    for k, v in orc.weights_numpy().items():
        assert np.abs(v - w_after_train[k]).max() < 2.5e-3 * len(cap.losses), k
    # evaluation: state snapshot, eval negatives, ranked lists, metrics
    st.save_state_checkpoint()
    hr, mrr = metrics.HitRate(3), metrics.MRR(3)
    it = 0
    losses = []
    for f, l in datasets.SessionDataset(files[2], scfg, batch_size=24, truncate_sequence_length=10):
        step = NARModuleModel.eval_step_key(orc.global_step, it)
        out = orc.forward(f, l, st.get_recent_clicks_buffer(), st.get_articles_recent_pop_norm(), mode='eval', step=step)
        hr.add(out['predicted_item_ids'].numpy(), l['label_next_item']); mrr.add(out['predicted_item_ids'].numpy(), l['label_next_item'])
        losses.append(float(out['total_loss']))
        st.update_items_state(*batch_clicks_for_state(f['item_clicked'], l['label_last_item'], f['event_timestamp']))
        it += 1
    st.restore_state_checkpoint()
    log = T.eval_sessions_metrics_log[-1]
    n_clicks = hr.hitrate_total
    assert log['clicks_count'] >= n_clicks and log['sessions_count'] == 60
    # near-tied probabilities may swap ranks between two correct fp32 evaluations: allow 1% of the clicks to differ
    assert abs(res['hitrate_at_n'] - hr.result()) <= 0.01 + 1e-9, (res, hr.result())
    assert abs(res['mrr_at_n'] - mrr.result()) <= 0.01, (res, mrr.result())
    assert abs(log['hitrate_at_n_chameleon'] - res['hitrate_at_n']) < 1e-9 and abs(log['mrr_at_n_chameleon'] - res['mrr_at_n']) < 1e-6
    assert abs(res['loss'] - np.mean(losses)) < 1e-3
    assert len(T.sessions_negative_items_log) == 60
    oracle_train(files[2:4], cap2)
    # training resumed from the restored state (evaluation clicks did not leak into the buffer)
    assert np.array_equal(T.clicked_items_state.pop_recent_clicks_buffer, st.pop_recent_clicks_buffer)
    # checkpoint round trip: a new Estimator on the same model_dir restores weights, Adam slots and the global step
    est2 = T.build_estimator(str(tmp_path / "model"), ace, meta, acfg, scfg)
    res2 = est2.evaluate(input_fn(files[4]))
    assert est2.global_step == est.global_step == len(cap.losses) + len(cap2.losses)
    import copy
    st_again = copy.deepcopy(st)       # (evaluation leaves articles_recent_pop_norm un-restored, like the reference: clicked_items_state.py:61-79)
    T.clicked_items_state = st
    res1 = est.evaluate(input_fn(files[4]))
    assert abs(res1['loss'] - res2['loss']) < 1e-6 and res1['hitrate_at_n'] == res2['hitrate_at_n']
    torch.cuda.synchronize()
    # warm start (nar_trainer_gcom.py:450-459): a fresh model_dir starts from another job's checkpoint
    from chameleon_recsys_amd.nar.estimator import Estimator, RunConfig
    est3 = Estimator(est._model_fn, model_dir=str(tmp_path / "model_warm"), config=RunConfig(tf_random_seed=42), params=est.params,
                     warm_start_from=str(tmp_path / "model"))
    T.clicked_items_state = st_again
    res3 = est3.evaluate(input_fn(files[4]))
    assert est3.global_step == est.global_step and abs(res3['loss'] - res1['loss']) < 1e-6
    # tf.estimator.Estimator.get_variable_names / get_variable_value under the reference graph's TF variable names
    names = est.get_variable_names()
    assert 'main/CAR/PreCAR_representation/kernel' in names and 'main/RNN/rnn/multi_rnn_cell/cell_0/ugrnn_cell/kernel' in names
    k = est.get_variable_value('main/RNN/rnn/multi_rnn_cell/cell_0/ugrnn_cell/kernel:0')
    rtl = est._store['runtime'].layout
    assert k.shape == (rtl.C + rtl.H, 2 * rtl.H) and np.array_equal(k, est.get_variable_value('rnn/0/kernel'))
    assert np.array_equal(est.get_variable_value('main/user_personalized_contextual_article_embedding/input/CAR_representation/kernel'),
                          est.get_variable_value('main/CAR/CAR_representation/kernel'))
    # a checkpoint written for another parameter layout is refused (same flat size would otherwise load silently)
    rt = est._store['runtime']
    sd = rt.state_dict()
    assert set(sd) >= {'flat', 'm', 'v', 'global_step', 'layout', 'dp_mode'}
    sd['layout'] = 'deadbeef'
    with pytest.raises(ValueError):
        rt.load_state_dict(sd)


def test_trainer_main_cli_end_to_end(gpu, tmp_path):
    """`python -m chameleon_recsys_amd.nar.nar_trainer_gcom --flags` (the reference's CLI, nar_trainer_gcom.py:418-586): hourly
    train -> evaluate loop over TFRecord files, metrics CSV, negative-samples / recommendations logs, checkpoint."""
    import json
    import os
    files, csv, pkl = synthetic.write_dataset(str(tmp_path / "data"), 5, 40, 300, 16, seq_len=10, seed=5)
    argv = ARGS + ['--train_set_path_regex', str(tmp_path / "data" / "sessions_hour_*.tfrecord.gz"),
                   '--acr_module_articles_metadata_csv_path', csv, '--acr_module_articles_content_embeddings_pickle_path', pkl,
                   '--model_dir', str(tmp_path / "model"), '--save_eval_sessions_recommendations', '--save_results_each_n_evals', '1']
    est = T.main(argv)
    md = str(tmp_path / "model")
    assert os.path.exists(os.path.join(md, "model.ckpt.pt"))
    lines = open(os.path.join(md, "eval_stats_benchmarks.csv")).read().strip().splitlines()
    assert len(lines) == 1 + 2 and 'hitrate_at_n_chameleon' in lines[0] and 'mrr_at_n' in lines[0]       # 5 files, chunks of 2 -> 2 evals
    recs = [json.loads(x) for x in open(os.path.join(md, "eval_chameleon_recommendations_log.json"))]
    assert len(recs) == 80 and len(recs[0]['predicted_item_ids'][0]) == 13 and recs[-1]['eval_hour_id'] == 1
    negs = [json.loads(x) for x in open(os.path.join(md, "eval_sessions_negative_samples.json"))]
    assert len(negs) == 80
    assert est.global_step == 8                                  # 4 training files x 40 sessions / batch 24 -> 2 steps each
    assert len(T.eval_sessions_metrics_log) == 2 and 0.0 <= T.eval_sessions_metrics_log[-1]['hitrate_at_n'] <= 1.0


def test_eval_cold_start_and_metrics_by_session_position_flags(gpu, tmp_path):
    """--eval_cold_start / --eval_metrics_by_session_position (nar_model.py:1444-1470, 1480-1494, 1621-1625, 1662-1666, 1718): the
    ranked candidates are produced in TRAIN steps too, the hook keeps the first-click / first-recommendation bookkeeping in both
    modes, evaluation snapshots it, and the eval metrics carry the per-position and cold-start statistics."""
    files, csv, pkl = synthetic.write_dataset(str(tmp_path / "data"), 5, 40, 300, 16, seq_len=10, seed=5)
    argv = ARGS + ['--train_set_path_regex', str(tmp_path / "data" / "sessions_hour_*.tfrecord.gz"),
                   '--acr_module_articles_metadata_csv_path', csv, '--acr_module_articles_content_embeddings_pickle_path', pkl,
                   '--model_dir', str(tmp_path / "model"), '--eval_cold_start', '--eval_metrics_by_session_position']
    est = T.main(argv)
    log = T.eval_sessions_metrics_log
    assert len(log) == 2
    last = log[-1]
    cs = last['coldstart_chameleon']
    assert cs['uniqueClickedItemsCount'] > 0 and cs['uniqueRecommendedItemsCount'] > 0 and cs['min'] >= 0
    pos_keys = [k for k in last if k.startswith('hitrate_at_n_by_pos_chameleon_')]
    assert pos_keys and all(0.0 <= last[k] <= 1.0 for k in pos_keys)
    assert 'clicks_at_pos_chameleon_01' in last and 'avg_norm_pop_by_pos_chameleon_01' in last
    n_pos = sum(last[k] for k in last if k.startswith('clicks_at_pos_chameleon_'))
    assert n_pos == last['clicks_count']
    st = T.clicked_items_state
    # TRAIN steps counted too (8 of them), the 4 evaluation steps rolled back by the snapshot restore
    assert st.get_current_step() == est.global_step == 8
    assert len(st.items_first_click_step) > 0


def test_adressa_trainer_main_end_to_end(gpu, tmp_path):
    """`python -m chameleon_recsys_amd.nar.nar_trainer_adressa` (nar_trainer_adressa.py:409-577): ACR resources as ONE pickle
    (label encoders, metadata DataFrame without the <PAD> row, ACE matrix with it), cardinalities from the encoder pickles, bytes
    ``user_id``, the Adressa click context; hourly train -> evaluate loop."""
    import os
    import pickle
    import pandas as pd
    from chameleon_recsys_amd.nar import nar_trainer_adressa as TA, tf_records_management as tfm
    rng = np.random.default_rng(3)
    n_items, D = 200, 16
    enc = lambda n: {'v%d' % i: i for i in range(n)}
    acr_enc = {'article_id': enc(n_items), 'category0': enc(12), 'category1': enc(30), 'author': enc(25)}
    df = pd.DataFrame({'article_id': np.arange(1, n_items), 'created_at_ts': synthetic.BASE_TS_MS - rng.integers(0, 5 * 86400000, n_items - 1),
                       'category0': rng.integers(0, 12, n_items - 1), 'category1': rng.integers(0, 30, n_items - 1),
                       'author': rng.integers(0, 25, n_items - 1)})
    ace = rng.standard_normal((n_items, D)).astype(np.float32); ace[0] = 0
    d = tmp_path / "adressa"; os.makedirs(d)
    with open(d / "acr.pickle", 'wb') as fh:
        pickle.dump((acr_enc, df, ace), fh)
    nar_enc = {'city': enc(40), 'region': enc(15), 'country': enc(11), 'device': enc(4), 'os': enc(8), 'referrer_class': enc(7)}
    with open(d / "nar.pickle", 'wb') as fh:
        pickle.dump({'nar_label_encoders': nar_enc}, fh)
    argv = [a for a in ARGS] + ['--train_set_path_regex', str(d / "sessions_hour_*.tfrecord.gz"), '--acr_module_resources_path', str(d / "acr.pickle"),
                                '--nar_module_preprocessing_resources_path', str(d / "nar.pickle"), '--model_dir', str(tmp_path / "model")]
    TA.base.FLAGS = TA.define_flags().parse_args(argv)
    scfg = TA.get_session_features_config(nar_enc)
    assert scfg['single_features']['user_id']['dtype'] == 'bytes' and scfg['sequence_features']['city']['cardinality'] == 40
    sid = 0
    for hour in range(5):
        ss = synthetic.make_sessions(40, 10, n_items, scfg, 7, hour, 'g1', sid)
        sid += len(ss)
        for s in ss:
            s['user_id'] = ('user-%d' % s['user_id']).encode()
        tfm.save_rows_to_tf_record_file(ss, scfg, str(d / ("sessions_hour_%03d.tfrecord.gz" % hour)))
    est = TA.main(argv)
    assert est.global_step == 8
    assert os.path.exists(os.path.join(str(tmp_path / "model"), "model.ckpt.pt"))
    L = est._store['runtime'].layout
    assert 'meta_emb/category1' in L.entries and 'ctx_emb/city' in L.entries and L.entries['ctx_emb/city'].shape[0] == 40
    assert len(TA.base.eval_sessions_metrics_log) == 2 and 0.0 <= TA.base.eval_sessions_metrics_log[-1]['hitrate_at_n'] <= 1.0


def _train_free_running(tmp_path, tag, files, csv, pkl, presample):
    """Estimator.train over the files with the device-resident state and NO per-step fetches (nothing synchronises the host with the
    device inside the loop: uploads, negative pre-sampling and state updates of the next batch run ahead on their own streams)."""
    import os
    from chameleon_recsys_amd.nar.clicked_items_state import DeviceClickedItemsState
    os.environ["CHAM_PRESAMPLE"] = "1" if presample else "0"
    try:
        argv = ['--batch_size', '32', '--truncate_session_length', '20', '--learning_rate', '1e-3', '--reg_l2', '1e-5', '--softmax_temperature', '0.2',
                '--recent_clicks_buffer_max_size', '2000', '--recent_clicks_for_normalization', '300', '--eval_metrics_top_n', '3',
                '--CAR_embedding_size', '256', '--rnn_units', '255', '--train_total_negative_samples', '20', '--train_negative_samples_from_buffer', '300',
                '--eval_total_negative_samples', '20', '--eval_negative_samples_from_buffer', '300', '--content_embedding_scale_factor', '6.0',
                '--disable_eval_benchmarks', '--clicked_items_state', 'device', '--model_dir', str(tmp_path / ("model_" + tag)),
                '--train_set_path_regex', str(tmp_path / "data" / "sessions_hour_*.tfrecord.gz"),
                '--acr_module_articles_metadata_csv_path', csv, '--acr_module_articles_content_embeddings_pickle_path', pkl]
        T.FLAGS = T.define_flags().parse_args(argv)
        meta_df, ace = T.load_acr_module_resources(csv, pkl)
        ace = T.l2_normalize_rows(ace) * np.float32(6.0)
        acfg = T.get_articles_features_config(n_items=ace.shape[0])
        meta = T.process_articles_metadata(meta_df, acfg)
        scfg = T.get_session_features_config()
        T.eval_sessions_metrics_log = []
        T.clicked_items_state = DeviceClickedItemsState(1.0, 2000, 300, ace.shape[0])
        est = T.build_estimator(str(tmp_path / ("model_" + tag)), ace, meta, acfg, scfg)
        est.config.log_step_count_steps = 0
        est.config.save_checkpoints_secs = 0
        est.train(lambda: datasets.prepare_dataset_iterator(files, scfg, batch_size=32, truncate_session_length=20))
        rt = est._store['runtime']
        assert rt.p3 and rt.presample == presample
        sd = rt.state_dict()
        return sd['flat'].numpy().copy(), sd['m'].numpy().copy(), int(sd['global_step'])
    finally:
        os.environ.pop("CHAM_PRESAMPLE", None)


def test_free_running_estimator_is_bit_reproducible_and_independent_of_the_look_ahead(gpu, tmp_path):
    """Ragged sessions (several padded lengths: every StepPlan is created inside the run, some of them by the look-ahead staging of the NEXT
    batch), plane-resident GEMMs, device-resident state, no host synchronisation inside Estimator.train: two runs end with bit-identical
    weights and Adam slots, and so does a run that draws its negatives at the head of each step instead of ahead of it."""
    files, csv, pkl = synthetic.write_dataset(str(tmp_path / "data"), 4, 160, 3000, 64, seq_len=20, seed=33, length_dist='g1')
    a = _train_free_running(tmp_path, "a", files, csv, pkl, True)
    b = _train_free_running(tmp_path, "b", files, csv, pkl, True)
    c = _train_free_running(tmp_path, "c", files, csv, pkl, False)
    assert a[2] == b[2] == c[2] >= 15
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "two free-running runs differ"
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), "look-ahead sampling changes the result"
