"""Adressa NAR trainer on MI355X - mirror of nar_module/nar/nar_trainer_adressa.py for the training path.

Differences from the G1 trainer (everything else is shared with nar_trainer_gcom): the ACR resources are ONE pickle
``(acr_label_encoders, articles_metadata_df, content_article_embeddings)`` (nar_utils.py:9-17), cardinalities come from the
label-encoder pickles (:130-132, :201-207), the articles metadata get a <PAD> row prepended (:142-144), ``user_id`` is a bytes
feature, the click context is city / region / country / device / os / referrer_class (:147-183).

  python -m chameleon_recsys_amd.nar.nar_trainer_adressa --train_set_path_regex '.../adressa_sessions_*.tfrecord.gz' \\
      --acr_module_resources_path acr_resources.pickle --nar_module_preprocessing_resources_path nar_resources.pickle ...
"""
import pickle
import sys
from collections import OrderedDict

import numpy as np

from . import nar_trainer_gcom as base
from .nar_trainer_gcom import ALL_FEATURES, nar_module_model_fn, build_estimator, train_and_evaluate_loop  # noqa: F401


def define_flags():
    ap = base.define_flags()
    ap.description = "CHAMELEON NAR module trainer (Adressa), MI355X-native"
    ap.add_argument('--nar_module_preprocessing_resources_path', default='/pickles', help='NAR module preprocessing resources path')
    return ap


def load_acr_module_resources(acr_module_resources_path):
    """nar_utils.py:9-17."""
    with open(acr_module_resources_path, 'rb') as fh:
        acr_label_encoders, articles_metadata_df, content_article_embeddings = pickle.load(fh)
    return acr_label_encoders, articles_metadata_df, np.asarray(content_article_embeddings, dtype=np.float32)


def load_nar_module_preprocessing_resources(path):
    """nar_utils.py:21-28."""
    with open(path, 'rb') as fh:
        return pickle.load(fh)['nar_label_encoders']


def get_articles_features_config(acr_label_encoders):
    """nar_trainer_adressa.py:106-134."""
    cfg = OrderedDict([
        ('article_id', {'type': 'categorical', 'dtype': 'int'}),
        ('created_at_ts', {'type': 'numerical', 'dtype': 'int'}),
        ('category0', {'type': 'categorical', 'dtype': 'int'}),
        ('category1', {'type': 'categorical', 'dtype': 'int'}),
        ('author', {'type': 'categorical', 'dtype': 'int'}),
    ])
    groups = {'category': ['category0', 'category1'], 'author': ['author']}
    if base.FLAGS.enabled_articles_input_features_groups != [ALL_FEATURES]:
        for g, feats in groups.items():
            if g not in base.FLAGS.enabled_articles_input_features_groups:
                for f in feats:
                    del cfg[f]
    for name in cfg:
        if name in acr_label_encoders and cfg[name]['type'] == 'categorical':
            cfg[name]['cardinality'] = len(acr_label_encoders[name])
    return cfg


def process_articles_metadata(articles_metadata_df, articles_features_config):
    """nar_trainer_adressa.py:137-145: a <PAD> row first, so that rows line up with the ACE matrix."""
    return {name: np.hstack([[0], articles_metadata_df[name].values]) for name in articles_features_config}


def get_session_features_config(nar_label_encoders_dict):
    """nar_trainer_adressa.py:147-210."""
    cfg = {
        'single_features': OrderedDict([
            ('user_id', {'type': 'categorical', 'dtype': 'bytes'}),
            ('session_id', {'type': 'numerical', 'dtype': 'int'}),
            ('session_size', {'type': 'numerical', 'dtype': 'int'}),
            ('session_start', {'type': 'numerical', 'dtype': 'int'}),
        ]),
        'sequence_features': OrderedDict([
            ('event_timestamp', {'type': 'numerical', 'dtype': 'int'}),
            ('item_clicked', {'type': 'categorical', 'dtype': 'int'}),
            ('city', {'type': 'categorical', 'dtype': 'int'}),
            ('region', {'type': 'categorical', 'dtype': 'int'}),
            ('country', {'type': 'categorical', 'dtype': 'int'}),
            ('device', {'type': 'categorical', 'dtype': 'int'}),
            ('os', {'type': 'categorical', 'dtype': 'int'}),
            ('local_hour_sin', {'type': 'numerical', 'dtype': 'float'}),
            ('local_hour_cos', {'type': 'numerical', 'dtype': 'float'}),
            ('weekday', {'type': 'numerical', 'dtype': 'float'}),
            ('referrer_class', {'type': 'categorical', 'dtype': 'int'}),
        ]),
    }
    groups = {'time': ['local_hour_sin', 'local_hour_cos', 'weekday'], 'device': ['device', 'os'],
              'location': ['country', 'region', 'city'], 'referrer': ['referrer_class']}
    if base.FLAGS.enabled_clicks_input_features_groups != [ALL_FEATURES]:
        for g, feats in groups.items():
            if g not in base.FLAGS.enabled_clicks_input_features_groups:
                for f in feats:
                    del cfg['sequence_features'][f]
    for group in cfg.values():
        for name, c in group.items():
            if name in nar_label_encoders_dict and c['type'] == 'categorical':
                c['cardinality'] = len(nar_label_encoders_dict[name])
    return cfg


def main(argv=None):
    """nar_trainer_adressa.py:409-577."""
    base.FLAGS = define_flags().parse_args(argv)
    acr_label_encoders, articles_metadata_df, ace = load_acr_module_resources(base.FLAGS.acr_module_resources_path)
    articles_features_config = get_articles_features_config(acr_label_encoders)
    articles_metadata = process_articles_metadata(articles_metadata_df, articles_features_config)
    nar_label_encoders = load_nar_module_preprocessing_resources(base.FLAGS.nar_module_preprocessing_resources_path)
    session_features_config = get_session_features_config(nar_label_encoders)
    return train_and_evaluate_loop(base.FLAGS, ace, articles_metadata, articles_features_config, session_features_config)


if __name__ == '__main__':
    main(sys.argv[1:])
