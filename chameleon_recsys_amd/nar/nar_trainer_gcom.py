"""G1 (Globo.com) NAR trainer on MI355X - mirror of nar_module/nar/nar_trainer_gcom.py for the training path:

  python -m chameleon_recsys_amd.nar.nar_trainer_gcom --train_set_path_regex '/data/sessions_hour_*.tfrecord.gz' \\
      --acr_module_articles_metadata_csv_path articles_metadata.csv \\
      --acr_module_articles_content_embeddings_pickle_path articles_embeddings.pickle --model_dir /tmp/m ...

Same flag names / defaults (:37-91; the two ACR path flags are the ones ``main`` actually reads, :463-464), same
feature configs (:99-218), ``nar_module_model_fn`` (:234-332), ``build_estimator`` (:335-386) and hourly
train -> evaluate-next-hour loop (:511-525), metrics CSV (nar_utils.py:31-40).  Out of scope: GCS / ML-Engine
plumbing (--use_local_cache_model_dir, --warmup_model_dir from GCS, TF_CONFIG trials) and the baseline recommenders
(``--disable_eval_benchmarks`` is effectively always on).
"""
import argparse
import json
import os
import pickle
import sys
from collections import OrderedDict
from time import time

import numpy as np

from .clicked_items_state import ClickedItemsState
from .datasets import prepare_dataset_iterator
from .estimator import Estimator, EstimatorSpec, RunConfig, StreamingMean
from .nar_model import ItemsStateUpdaterHook, ModeKeys, NARModuleModel
from .utils import chunks, resolve_files

RANDOM_SEED = 42
ALL_FEATURES = 'ALL'


def _bool(v):
    return str(v).lower() in ('1', 'true', 't', 'yes', 'y')


def _list(v):
    return [x for x in str(v).split(',') if x != '']


def define_flags():
    """tf.flags of nar_trainer_gcom.py:37-91, same names, defaults and help strings (argparse)."""
    ap = argparse.ArgumentParser(description="CHAMELEON NAR module trainer (G1), MI355X-native")
    a = ap.add_argument
    a('--batch_size', type=int, default=64, help='Batch size')
    a('--truncate_session_length', type=int, default=20, help='Truncate long sessions to this max. size')
    a('--learning_rate', type=float, default=1e-3, help='Lerning Rate')
    a('--dropout_keep_prob', type=float, default=1.0, help='Dropout (keep prob.)')
    a('--reg_l2', type=float, default=0.0002, help='L2 regularization')
    a('--softmax_temperature', type=float, default=1.0, help='Initial value for temperature for softmax')
    a('--recent_clicks_buffer_hours', type=float, default=1.0, help='Number of hours that will be kept in the recent clicks buffer')
    a('--recent_clicks_buffer_max_size', type=int, default=500, help='Maximum size of recent clicks buffer')
    a('--recent_clicks_for_normalization', type=int, default=500, help='Number of recent clicks to normalize recency / novelty')
    a('--eval_metrics_top_n', type=int, default=3, help='Eval. metrics Top N')
    a('--CAR_embedding_size', type=int, default=512, help='CAR submodule embedding size')
    a('--rnn_units', type=int, default=1024, help='Number of units of RNN cell')
    a('--rnn_num_layers', type=int, default=1, help='Number of of RNN layers')
    a('--train_total_negative_samples', type=int, default=5, help='Total negative samples for training')
    a('--train_negative_samples_from_buffer', type=int, default=10, help='Training Negative samples from recent clicks buffer')
    a('--eval_total_negative_samples', type=int, default=20, help='Total negative samples for evaluation')
    a('--eval_negative_samples_from_buffer', type=int, default=50, help='Eval. Negative samples from recent clicks buffer')
    a('--save_histograms', type=_bool, default=False, nargs='?', const=True)
    a('--disable_eval_benchmarks', type=_bool, default=False, nargs='?', const=True, help='Disable eval benchmarks')
    a('--eval_metrics_by_session_position', type=_bool, default=False, nargs='?', const=True)
    a('--novelty_reg_factor', type=float, default=0.0)
    a('--diversity_reg_factor', type=float, default=0.0)
    a('--eval_negative_sample_relevance', type=float, default=0.1)
    a('--content_embedding_scale_factor', type=float, default=1.0)
    a('--enabled_clicks_input_features_groups', type=_list, default=[ALL_FEATURES])
    a('--enabled_articles_input_features_groups', type=_list, default=[ALL_FEATURES])
    a('--enabled_internal_features', type=_list, default=[ALL_FEATURES])
    a('--train_set_path_regex', default='/train*.tfrecord', help='Train set regex')
    a('--acr_module_resources_path', default='/pickles', help='ACR module resources path')
    a('--acr_module_articles_metadata_csv_path', default=None)
    a('--acr_module_articles_content_embeddings_pickle_path', default=None)
    a('--model_dir', default='./tmp', help='Directory where save model checkpoints')
    a('--warmup_model_dir', default=None)
    a('--train_files_from', type=int, default=0)
    a('--train_files_up_to', type=int, default=100)
    a('--training_hours_for_each_eval', type=int, default=5)
    a('--save_results_each_n_evals', type=int, default=5)
    a('--save_eval_sessions_negative_samples', type=_bool, default=False, nargs='?', const=True)
    a('--save_eval_sessions_recommendations', type=_bool, default=False, nargs='?', const=True)
    a('--eval_cold_start', type=_bool, default=False, nargs='?', const=True)
    a('--use_local_cache_model_dir', type=_bool, default=False, nargs='?', const=True)
    a('--job-dir', default='./tmp')
    # ---- extensions of this implementation (not in the reference's flag set)
    a('--rnn_cell', default='ugrnn', choices=['ugrnn', 'gru'], help="recurrent cell: 'ugrnn' = the reference's tf.contrib.rnn.UGRNNCell "
      "(nar_model.py:1317), 'gru' = its commented-out GRUCell alternative (:1315)")
    a('--gemm_dtype', default='f32', choices=['f32', 'f32_native', 'bf16'],
      help="f32: fp32 operands and fp32-grade error, wide GEMMs as six bf16-plane products on the bf16 matrix cores (caveats: an INFINITE "
           "operand yields NaN where fp32 yields +-inf - behind a tanh layer the native path saturates to a finite +-1 instead - and "
           "operands below 2^-100 lose their lowest plane; tests/test_gemm_p3_gpu.py pins both); f32_native: every GEMM on the native "
           "fp32 MFMA; bf16: bf16-resident candidate-row matrices, fp32 accumulate")
    a('--clicked_items_state', default='host', choices=['host', 'device'], help="keep the recent-clicks state in host numpy (reference "
      "class) or in HBM (bit-identical, no host round trip per step)")
    return ap


FLAGS = define_flags().parse_args([])          # module-level defaults, replaced by main()


def get_articles_features_config(n_items=None):
    """nar_trainer_gcom.py:99-128 (+ the article_id cardinality the model reads at nar_model.py:183, which the G1
    trainer forgets - the Adressa trainer sets it, nar_trainer_adressa.py:130-132)."""
    cfg = OrderedDict([
        ('article_id', {'type': 'categorical', 'dtype': 'int'}),
        ('created_at_ts', {'type': 'numerical', 'dtype': 'int'}),
        ('category_id', {'type': 'categorical', 'dtype': 'int', 'cardinality': 461}),
    ])
    if n_items is not None:
        cfg['article_id']['cardinality'] = int(n_items)
    groups = {'category': ['category_id']}
    if FLAGS.enabled_articles_input_features_groups != [ALL_FEATURES]:
        for g, feats in groups.items():
            if g not in FLAGS.enabled_articles_input_features_groups:
                for f in feats:
                    del cfg[f]
    return cfg


def get_session_features_config():
    """nar_trainer_gcom.py:150-218."""
    from .config import get_session_features_config_gcom
    cfg = get_session_features_config_gcom(364047)
    groups = {'time': ['local_hour_sin', 'local_hour_cos', 'local_weekday'], 'device': ['environment', 'deviceGroup', 'os'],
              'location': ['country', 'region'], 'referrer': ['referrer_type']}
    if FLAGS.enabled_clicks_input_features_groups != [ALL_FEATURES]:
        for g, feats in groups.items():
            if g not in FLAGS.enabled_clicks_input_features_groups:
                for f in feats:
                    del cfg['sequence_features'][f]
    return cfg


def get_internal_enabled_features_config():
    """nar_trainer_gcom.py:220-231."""
    valid = ['recency', 'novelty', 'article_content_embeddings', 'item_clicked_embeddings']
    enabled = set(valid) if FLAGS.enabled_internal_features == [ALL_FEATURES] else set(FLAGS.enabled_internal_features) & set(valid)
    return {f: (f in enabled) for f in valid}


def load_acr_module_resources(articles_metadata_csv_path, articles_content_embeddings_pickle_path):
    """nar_trainer_gcom.py:131-139: (articles metadata DataFrame, ACE matrix [n_items, D])."""
    import pandas as pd
    with open(articles_content_embeddings_pickle_path, 'rb') as fh:
        ace = pickle.load(fh)
    return pd.read_csv(articles_metadata_csv_path), np.asarray(ace, dtype=np.float32)


def process_articles_metadata(articles_metadata_df, articles_features_config):
    return {name: articles_metadata_df[name].values for name in articles_features_config}


def l2_normalize_rows(x):
    """sklearn.preprocessing.Normalizer(norm='l2').fit_transform (nar_trainer_gcom.py:470-471): rows of zero norm stay 0."""
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt((x.astype(np.float32) ** 2).sum(axis=1, keepdims=True))
    n[n == 0.0] = 1.0
    return x / n


# Global vars updated by the Estimator hook (nar_trainer_gcom.py:411-415)
clicked_items_state = None
eval_sessions_metrics_log = []
sessions_negative_items_log = None
sessions_chameleon_recommendations_log = None
global_eval_hour_id = 0


def nar_module_model_fn(features, labels, mode, params):
    """nar_trainer_gcom.py:234-332."""
    if mode == ModeKeys.TRAIN:
        negative_samples = params['train_total_negative_samples']
        negative_sample_from_buffer = params['train_negative_samples_from_buffer']
    elif mode == ModeKeys.EVAL:
        negative_samples = params['eval_total_negative_samples']
        negative_sample_from_buffer = params['eval_negative_samples_from_buffer']
    else:
        raise ValueError("mode %r: the reference defines TRAIN and EVAL only" % (mode,))
    dropout_keep_prob = params['dropout_keep_prob'] if mode == ModeKeys.TRAIN else 1.0
    internal_features_config = params.get('internal_features_config') or get_internal_enabled_features_config()
    eval_metrics_top_n = params['eval_metrics_top_n']
    model = NARModuleModel(mode, features, labels,
                           session_features_config=params['session_features_config'],
                           articles_features_config=params['articles_features_config'],
                           batch_size=params['batch_size'], lr=params['lr'], keep_prob=dropout_keep_prob,
                           negative_samples=negative_samples, negative_sample_from_buffer=negative_sample_from_buffer,
                           reg_weight_decay=params['reg_weight_decay'], softmax_temperature=params['softmax_temperature'],
                           articles_metadata=params['articles_metadata'],
                           content_article_embeddings_matrix=params['content_article_embeddings_matrix'],
                           recent_clicks_buffer_hours=params['recent_clicks_buffer_hours'],
                           recent_clicks_buffer_max_size=params['recent_clicks_buffer_max_size'],
                           recent_clicks_for_normalization=params['recent_clicks_for_normalization'],
                           CAR_embedding_size=params['CAR_embedding_size'], rnn_units=params['rnn_units'],
                           rnn_num_layers=params.get('rnn_num_layers', 1),
                           metrics_top_n=eval_metrics_top_n, plot_histograms=params['save_histograms'],
                           novelty_reg_factor=params['novelty_reg_factor'], diversity_reg_factor=params['diversity_reg_factor'],
                           internal_features_config=internal_features_config, eval_cold_start=params['eval_cold_start'],
                           rnn_cell=params.get('rnn_cell', 'ugrnn'), gemm_dtype=params.get('gemm_dtype', 'f32'))
    state = params.get('clicked_items_state') or clicked_items_state
    metrics_log = params.get('eval_sessions_metrics_log', eval_sessions_metrics_log)
    eval_metrics = {'hitrate_at_n': StreamingMean(), 'mrr_at_n': StreamingMean()} if mode == ModeKeys.EVAL else {}
    hooks = [ItemsStateUpdaterHook(mode, model, eval_metrics_top_n=eval_metrics_top_n, clicked_items_state=state,
                                   eval_sessions_metrics_log=metrics_log,
                                   sessions_negative_items_log=sessions_negative_items_log,
                                   sessions_chameleon_recommendations_log=sessions_chameleon_recommendations_log,
                                   content_article_embeddings_matrix=params['content_article_embeddings_matrix'],
                                   articles_metadata=params['articles_metadata'],
                                   eval_negative_sample_relevance=params['eval_negative_sample_relevance'],
                                   eval_benchmark_classifiers=[],
                                   eval_metrics_by_session_position=params['eval_metrics_by_session_position'],
                                   eval_cold_start=params['eval_cold_start'], eval_metric_ops=eval_metrics)]
    if mode == ModeKeys.TRAIN:
        return EstimatorSpec(mode, loss=model.loss_t, train_op=model.train, training_chief_hooks=hooks)
    return EstimatorSpec(mode, loss=model.loss_t, eval_metric_ops=eval_metrics, evaluation_hooks=hooks)


def build_estimator(model_dir, content_article_embeddings_matrix, articles_metadata, articles_features_config,
                    session_features_config):
    """nar_trainer_gcom.py:335-386."""
    run_config = RunConfig(tf_random_seed=RANDOM_SEED, keep_checkpoint_max=1, save_checkpoints_secs=1200,
                           save_summary_steps=100, log_step_count_steps=100)
    return Estimator(config=run_config, model_dir=model_dir, model_fn=nar_module_model_fn, params={
        'batch_size': FLAGS.batch_size, 'lr': FLAGS.learning_rate, 'dropout_keep_prob': FLAGS.dropout_keep_prob,
        'reg_weight_decay': FLAGS.reg_l2, 'recent_clicks_buffer_hours': FLAGS.recent_clicks_buffer_hours,
        'recent_clicks_buffer_max_size': FLAGS.recent_clicks_buffer_max_size,
        'recent_clicks_for_normalization': FLAGS.recent_clicks_for_normalization,
        'eval_metrics_top_n': FLAGS.eval_metrics_top_n, 'CAR_embedding_size': FLAGS.CAR_embedding_size,
        'rnn_units': FLAGS.rnn_units, 'rnn_num_layers': 1,   # the reference never forwards --rnn_num_layers (:252-275)
        'rnn_cell': FLAGS.rnn_cell, 'gemm_dtype': FLAGS.gemm_dtype,
        'train_total_negative_samples': FLAGS.train_total_negative_samples,
        'train_negative_samples_from_buffer': FLAGS.train_negative_samples_from_buffer,
        'eval_total_negative_samples': FLAGS.eval_total_negative_samples,
        'eval_negative_samples_from_buffer': FLAGS.eval_negative_samples_from_buffer,
        'softmax_temperature': FLAGS.softmax_temperature, 'save_histograms': FLAGS.save_histograms,
        'eval_metrics_by_session_position': FLAGS.eval_metrics_by_session_position,
        'novelty_reg_factor': FLAGS.novelty_reg_factor, 'diversity_reg_factor': FLAGS.diversity_reg_factor,
        'eval_negative_sample_relevance': FLAGS.eval_negative_sample_relevance, 'eval_cold_start': FLAGS.eval_cold_start,
        'session_features_config': session_features_config, 'articles_features_config': articles_features_config,
        'articles_metadata': articles_metadata, 'content_article_embeddings_matrix': content_article_embeddings_matrix})


def save_eval_benchmark_metrics_csv(metrics_log, output_dir, training_hours_for_each_eval, output_csv='eval_stats_benchmarks.csv'):
    """nar_utils.py:31-40."""
    import pandas as pd
    df = pd.DataFrame(metrics_log).reset_index()
    if len(df):
        df['hour'] = df['index'].apply(lambda x: ((x + 1) * training_hours_for_each_eval) % 24)
        df['day'] = df['index'].apply(lambda x: int(((x + 1) * training_hours_for_each_eval) / 24))
    df.to_csv(os.path.join(output_dir, output_csv), index=False)


def _append_json_lines(path, rows):
    with open(path, 'a') as fh:
        for r in rows:
            fh.write(json.dumps(r) + '\n')


def train_and_evaluate_loop(flags, ace, articles_metadata, articles_features_config, session_features_config,
                            state_cls=ClickedItemsState):
    """The hourly loop shared by both trainers: nar_trainer_gcom.py:476-582 / nar_trainer_adressa.py:472-570.
    ``ace`` is the raw ACE matrix; it is L2-row-normalised and scaled here (:470-474)."""
    global FLAGS, clicked_items_state, eval_sessions_metrics_log, sessions_negative_items_log
    global sessions_chameleon_recommendations_log, global_eval_hour_id
    FLAGS = flags
    if FLAGS.use_local_cache_model_dir or FLAGS.warmup_model_dir:
        raise NotImplementedError("GCS model-dir caching / warm start download is out of scope (copy model.ckpt.pt into --model_dir)")
    np.random.seed(RANDOM_SEED)
    os.makedirs(FLAGS.model_dir, exist_ok=True)
    ace = l2_normalize_rows(ace) * np.float32(FLAGS.content_embedding_scale_factor)
    eval_sessions_metrics_log = []
    sessions_negative_items_log = [] if FLAGS.save_eval_sessions_negative_samples else None
    sessions_chameleon_recommendations_log = [] if FLAGS.save_eval_sessions_recommendations else None
    if getattr(FLAGS, 'clicked_items_state', 'host') == 'device':
        from .clicked_items_state import DeviceClickedItemsState
        state_cls = DeviceClickedItemsState
    clicked_items_state = state_cls(FLAGS.recent_clicks_buffer_hours, FLAGS.recent_clicks_buffer_max_size,
                                    FLAGS.recent_clicks_for_normalization, ace.shape[0])
    model = build_estimator(FLAGS.model_dir, ace, articles_metadata, articles_features_config, session_features_config)
    train_files = resolve_files(FLAGS.train_set_path_regex)
    if FLAGS.train_files_from > FLAGS.train_files_up_to:
        raise Exception('Final training file cannot be lower than Starting training file')
    train_files = train_files[FLAGS.train_files_from:FLAGS.train_files_up_to + 1]
    print('INFO:{} files where the network will be trained and evaluated on, from {} to {}'.format(
        len(train_files), train_files[0], train_files[-1]), flush=True)
    start_train = time()
    training_files_chunks = list(chunks(train_files, FLAGS.training_hours_for_each_eval))
    input_fn = lambda files: (lambda: prepare_dataset_iterator(files, session_features_config, batch_size=FLAGS.batch_size,
                                                               truncate_session_length=FLAGS.truncate_session_length))
    for chunk_id in range(0, len(training_files_chunks) - 1):
        chunk = training_files_chunks[chunk_id]
        print('INFO:Training files from {} to {}'.format(chunk[0], chunk[-1]), flush=True)
        model.train(input_fn=input_fn(chunk))
        eval_file = training_files_chunks[chunk_id + 1][0]
        print('INFO:Evaluating file {}'.format(eval_file), flush=True)
        res = model.evaluate(input_fn=input_fn(eval_file))
        print('INFO:eval {}'.format({k: (round(float(v), 5) if isinstance(v, (float, np.floating)) else v) for k, v in res.items()}),
              flush=True)
        if chunk_id % FLAGS.save_results_each_n_evals == 0:
            save_eval_benchmark_metrics_csv(eval_sessions_metrics_log, FLAGS.model_dir, FLAGS.training_hours_for_each_eval)
            if FLAGS.save_eval_sessions_negative_samples:
                _append_json_lines(os.path.join(FLAGS.model_dir, 'eval_sessions_negative_samples.json'), sessions_negative_items_log)
                sessions_negative_items_log.clear()
            if FLAGS.save_eval_sessions_recommendations:
                _append_json_lines(os.path.join(FLAGS.model_dir, 'eval_chameleon_recommendations_log.json'),
                                   [dict(eval_hour_id=global_eval_hour_id, **r) for r in sessions_chameleon_recommendations_log])
                sessions_chameleon_recommendations_log.clear()
                global_eval_hour_id += 1
    save_eval_benchmark_metrics_csv(eval_sessions_metrics_log, FLAGS.model_dir, FLAGS.training_hours_for_each_eval)
    if FLAGS.save_eval_sessions_negative_samples:
        _append_json_lines(os.path.join(FLAGS.model_dir, 'eval_sessions_negative_samples.json'), sessions_negative_items_log)
    if FLAGS.save_eval_sessions_recommendations:
        _append_json_lines(os.path.join(FLAGS.model_dir, 'eval_chameleon_recommendations_log.json'),
                           [dict(eval_hour_id=global_eval_hour_id, **r) for r in sessions_chameleon_recommendations_log])
    print('INFO:==== Finalized TRAINING Loop elapsed {:.1f} minutes'.format((time() - start_train) / 60.0), flush=True)
    return model


def main(argv=None):
    """nar_trainer_gcom.py:418-586."""
    global FLAGS
    FLAGS = define_flags().parse_args(argv)
    articles_metadata_df, ace = load_acr_module_resources(FLAGS.acr_module_articles_metadata_csv_path,
                                                          FLAGS.acr_module_articles_content_embeddings_pickle_path)
    articles_features_config = get_articles_features_config(n_items=ace.shape[0])
    articles_metadata = process_articles_metadata(articles_metadata_df, articles_features_config)
    session_features_config = get_session_features_config()
    return train_and_evaluate_loop(FLAGS, ace, articles_metadata, articles_features_config, session_features_config)


if __name__ == '__main__':
    main(sys.argv[1:])
