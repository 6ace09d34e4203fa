"""Feature configurations of the two shipped trainers (dict insertion order = feature concat order).

G1: nar_module/nar/nar_trainer_gcom.py:99-128 (articles) and :150-218 (sessions);
Adressa: nar_module/nar/nar_trainer_adressa.py:104-183 (cardinalities from its encoder pickles / comments).
"""
from collections import OrderedDict


def get_articles_features_config_gcom(n_items, category_cardinality=461):
    return OrderedDict([
        ('article_id', {'type': 'categorical', 'dtype': 'int', 'cardinality': n_items}),
        ('created_at_ts', {'type': 'numerical', 'dtype': 'int'}),
        ('category_id', {'type': 'categorical', 'dtype': 'int', 'cardinality': category_cardinality}),
    ])


def get_session_features_config_gcom(n_items=364047):
    return {
        'single_features': OrderedDict([
            ('user_id', {'type': 'categorical', 'dtype': 'int', 'cardinality': 341193}),
            ('session_id', {'type': 'categorical', 'dtype': 'int'}),
            ('session_start', {'type': 'categorical', 'dtype': 'int'}),
            ('session_size', {'type': 'categorical', 'dtype': 'int'}),
        ]),
        'sequence_features': OrderedDict([
            ('event_timestamp', {'type': 'numerical', 'dtype': 'int'}),
            ('item_clicked', {'type': 'categorical', 'dtype': 'int', 'cardinality': n_items}),
            ('environment', {'type': 'categorical', 'dtype': 'int', 'cardinality': 5}),
            ('deviceGroup', {'type': 'categorical', 'dtype': 'int', 'cardinality': 6}),
            ('os', {'type': 'categorical', 'dtype': 'int', 'cardinality': 23}),
            ('country', {'type': 'categorical', 'dtype': 'int', 'cardinality': 12}),
            ('region', {'type': 'categorical', 'dtype': 'int', 'cardinality': 29}),
            ('local_hour_sin', {'type': 'numerical', 'dtype': 'float'}),
            ('local_hour_cos', {'type': 'numerical', 'dtype': 'float'}),
            ('local_weekday', {'type': 'numerical', 'dtype': 'float'}),
            ('referrer_type', {'type': 'categorical', 'dtype': 'int', 'cardinality': 8}),
        ]),
    }


def get_articles_features_config_adressa(n_items, cards=(41, 128, 112)):
    return OrderedDict([
        ('article_id', {'type': 'categorical', 'dtype': 'int', 'cardinality': n_items}),
        ('created_at_ts', {'type': 'numerical', 'dtype': 'int'}),
        ('category0', {'type': 'categorical', 'dtype': 'int', 'cardinality': cards[0]}),
        ('category1', {'type': 'categorical', 'dtype': 'int', 'cardinality': cards[1]}),
        ('author', {'type': 'categorical', 'dtype': 'int', 'cardinality': cards[2]}),
    ])


def get_session_features_config_adressa(n_items=13000):
    return {
        'single_features': OrderedDict([
            ('user_id', {'type': 'categorical', 'dtype': 'int'}),
            ('session_id', {'type': 'categorical', 'dtype': 'int'}),
            ('session_start', {'type': 'categorical', 'dtype': 'int'}),
            ('session_size', {'type': 'categorical', 'dtype': 'int'}),
        ]),
        'sequence_features': OrderedDict([
            ('event_timestamp', {'type': 'numerical', 'dtype': 'int'}),
            ('item_clicked', {'type': 'categorical', 'dtype': 'int', 'cardinality': n_items}),
            ('city', {'type': 'categorical', 'dtype': 'int', 'cardinality': 1022}),
            ('region', {'type': 'categorical', 'dtype': 'int', 'cardinality': 237}),
            ('country', {'type': 'categorical', 'dtype': 'int', 'cardinality': 70}),
            ('device', {'type': 'categorical', 'dtype': 'int', 'cardinality': 5}),
            ('os', {'type': 'categorical', 'dtype': 'int', 'cardinality': 10}),
            ('referrer_class', {'type': 'categorical', 'dtype': 'int', 'cardinality': 7}),
            ('weekday', {'type': 'numerical', 'dtype': 'float'}),
            ('local_hour_sin', {'type': 'numerical', 'dtype': 'float'}),
            ('local_hour_cos', {'type': 'numerical', 'dtype': 'float'}),
        ]),
    }
