"""Synthetic G1-/Adressa-shaped catalog and sessions - there is no network for the real
datasets.  Seeded numpy; produces the same ``(features, labels)`` batches as ``datasets.prepare_dataset_iterator``
(datasets.py:35-143 of the reference) or session dicts for the TFRecord writer."""
import math
from collections import OrderedDict

import numpy as np

BASE_TS_MS = 1506826800000          # 2017-10-01 03:00 UTC (G1 era)
HOUR_MS = 3600 * 1000


def make_catalog(n_items, ace_dim, seed=42, scale_factor=6.0, meta_cardinalities=None):
    """ACE matrix (N(0,1) -> row L2-normalise -> x scale, nar_trainer_gcom.py:470-474) + articles metadata."""
    rng = np.random.default_rng(seed)
    ace = rng.standard_normal((n_items, ace_dim)).astype(np.float32)
    ace /= np.linalg.norm(ace, axis=1, keepdims=True)
    ace *= np.float32(scale_factor)
    meta = OrderedDict()
    meta['article_id'] = np.arange(n_items, dtype=np.int64)
    # published within ~a week before the first click hour, newer ids later (so the "recent window" moves)
    order = np.sort(rng.exponential(1.5 * 24 * HOUR_MS, size=n_items))[::-1]
    meta['created_at_ts'] = (BASE_TS_MS - order).astype(np.int64)
    for name, card in (meta_cardinalities or {'category_id': 461}).items():
        meta[name] = rng.integers(0, card, size=n_items, dtype=np.int64)
    return ace, meta


def _time_features(ts_ms):
    """nar_preprocess_gcom.py:53-73: hour-of-day sin/cos in America/Sao_Paulo (UTC-3, no DST handling), weekday."""
    secs = ts_ms // 1000 - 3 * 3600
    hour = (secs % 86400) / 3600.0
    wd = ((secs // 86400) + 3) % 7                        # 1970-01-01 was a Thursday (weekday() == 3)
    return (np.sin(2 * math.pi * (hour + 1e-6) / 24).astype(np.float32),
            np.cos(2 * math.pi * (hour + 1e-6) / 24).astype(np.float32), ((wd + 1) / 7.0).astype(np.float32))


def make_sessions(n_sessions, seq_len, n_items, session_features_config, seed=42, hour_index=0, length_dist='g1',
                  first_session_id=0):
    """Returns a list of session dicts {context ints, per-click lists} sorted by session_start."""
    rng = np.random.default_rng(seed + 7919 * (hour_index + 1))
    if length_dist == 'full':
        lens = np.full(n_sessions, seq_len, dtype=np.int64)
    else:                                                  # G1-like: mean ~4 clicks, hard max seq_len
        lens = 2 + np.minimum(rng.geometric(0.45, size=n_sessions), seq_len - 2)
    hour_start = BASE_TS_MS + hour_index * HOUR_MS
    starts = np.sort(hour_start + rng.integers(0, HOUR_MS, size=n_sessions))
    win = max(64, int(0.05 * n_items))
    lo = 1 + (hour_index * max(1, win // 8)) % max(1, n_items - 1 - win)
    seqcfg = session_features_config['sequence_features']
    sessions = []
    for i in range(n_sessions):
        n = int(lens[i])
        # Zipf(1.1)-ranked ids inside the sliding "recent" window
        r = np.minimum(rng.zipf(1.1, size=n), win) - 1
        ids = lo + (r * 7919) % win                       # scatter ranks over the window
        ids = np.clip(ids, 1, n_items - 1).astype(np.int64)
        ts = starts[i] + np.concatenate([[0], np.cumsum(rng.exponential(60000.0, size=n - 1))]).astype(np.int64)
        hs, hc, wd = _time_features(ts)
        s = OrderedDict(user_id=int(rng.integers(1, 341193)), session_id=int(first_session_id + i),
                        session_start=int(starts[i]), session_size=n,
                        event_timestamp=ts.astype(np.int64), item_clicked=ids)
        for name, cfg in seqcfg.items():
            if name in ('event_timestamp', 'item_clicked'):
                continue
            if cfg['type'] == 'categorical':
                # device / location context is (mostly) constant within a session
                s[name] = np.full(n, rng.integers(0, cfg['cardinality']), dtype=np.int64)
            elif name == 'local_hour_sin':
                s[name] = hs
            elif name == 'local_hour_cos':
                s[name] = hc
            elif name in ('local_weekday', 'weekday'):
                s[name] = wd
            else:
                s[name] = rng.standard_normal(n).astype(np.float32)
        sessions.append(s)
    return sessions


def batch_from_sessions(sessions, session_features_config, truncate_session_length=20):
    """The transform of datasets.py:35-143: truncate, labels = clicks shifted by one, drop last input, zero-pad."""
    single = session_features_config['single_features']
    seqcfg = session_features_config['sequence_features']
    B = len(sessions)
    sizes = np.array([min(s['session_size'], truncate_session_length) for s in sessions], dtype=np.int64)
    T = int(sizes.max()) - 1
    feats = OrderedDict()
    for name in single:
        if name == 'session_size':
            feats[name] = sizes
        else:
            feats[name] = np.array([s[name] for s in sessions], dtype=np.int64)
    for name, cfg in seqcfg.items():
        dt = np.int64 if cfg['dtype'] == 'int' else np.float32
        a = np.zeros((B, T), dtype=dt)
        for i, s in enumerate(sessions):
            v = np.asarray(s[name])[:truncate_session_length][:-1]
            a[i, :len(v)] = v
        feats[name] = a
    nxt = np.zeros((B, T), dtype=np.int64)
    last = np.zeros((B, 1), dtype=np.int64)
    for i, s in enumerate(sessions):
        it = np.asarray(s['item_clicked'])[:truncate_session_length]
        nxt[i, :len(it) - 1] = it[1:]
        last[i, 0] = it[-1]
    return feats, OrderedDict(label_next_item=nxt, label_last_item=last)


def make_batches(n_batches, batch_size, seq_len, n_items, session_features_config, seed=42, length_dist='g1',
                 sessions_per_hour=None):
    """Convenience: a list of (features, labels) batches in stream order (one "hour" = sessions_per_hour sessions)."""
    sessions_per_hour = sessions_per_hour or batch_size * 4
    out, sid, hour, pend = [], 0, 0, []
    while len(out) < n_batches:
        ss = make_sessions(sessions_per_hour, seq_len, n_items, session_features_config, seed, hour, length_dist, sid)
        sid += len(ss); hour += 1
        pend += ss
        while len(pend) >= batch_size and len(out) < n_batches:
            out.append(batch_from_sessions(pend[:batch_size], session_features_config, seq_len))
            pend = pend[batch_size:]
    return out


def default_params(n_items, ace_dim, seq_len=20, batch_size=256, neg=50, neg_from_buffer=3000, buffer_size=20000,
                   for_norm=2000, C=1024, H=255, dataset='gcom', seed=42, **over):
    """The Estimator ``params`` dict (nar_trainer_gcom.py:355-384) with the shipped G1 hyper-parameters
    (scripts/run_nar_train_gcom_mlengine.sh:29-51)."""
    from . import config
    if dataset == 'gcom':
        scfg = config.get_session_features_config_gcom(n_items)
        acfg = config.get_articles_features_config_gcom(n_items)
        ace, meta = make_catalog(n_items, ace_dim, seed, 6.0, {'category_id': 461})
    else:
        scfg = config.get_session_features_config_adressa(n_items)
        acfg = config.get_articles_features_config_adressa(n_items)
        ace, meta = make_catalog(n_items, ace_dim, seed, 6.0, {'category0': 41, 'category1': 128, 'author': 112})
    p = dict(batch_size=batch_size, lr=1e-4, dropout_keep_prob=1.0, reg_weight_decay=1e-5,
             recent_clicks_buffer_hours=1.0, recent_clicks_buffer_max_size=buffer_size,
             recent_clicks_for_normalization=for_norm, eval_metrics_top_n=10, CAR_embedding_size=C, rnn_units=H,
             rnn_num_layers=1, train_total_negative_samples=neg, train_negative_samples_from_buffer=neg_from_buffer,
             eval_total_negative_samples=neg, eval_negative_samples_from_buffer=neg_from_buffer, softmax_temperature=0.1,
             save_histograms=False, eval_metrics_by_session_position=False, novelty_reg_factor=0.0,
             diversity_reg_factor=0.0, eval_negative_sample_relevance=0.02, eval_cold_start=False,
             session_features_config=scfg, articles_features_config=acfg, articles_metadata=meta,
             content_article_embeddings_matrix=ace, truncate_session_length=seq_len)
    p.update(over)
    return p


def write_dataset(out_dir, n_hours, sessions_per_hour, n_items, ace_dim, seq_len=20, seed=42, length_dist='g1',
                  dataset='gcom'):
    """Writes a synthetic G1-shaped dataset the trainer can consume unchanged: ``sessions_hour_NNN.tfrecord.gz`` (one GZIP
    TFRecord file per hour) + the ACR-module resources ``articles_metadata.csv`` and
    ``articles_embeddings.pickle`` (un-normalised ACE matrix; the trainer L2-normalises and scales it).
    Returns (files, metadata_csv, embeddings_pickle)."""
    import os
    import pickle
    from . import config, tf_records_management as tfm
    os.makedirs(out_dir, exist_ok=True)
    scfg = config.get_session_features_config_gcom(n_items) if dataset == 'gcom' else config.get_session_features_config_adressa(n_items)
    cards = {'category_id': 461} if dataset == 'gcom' else {'category0': 41, 'category1': 128, 'author': 112}
    ace, meta = make_catalog(n_items, ace_dim, seed, 1.0, cards)
    rng = np.random.default_rng(seed + 1)
    ace = ace * rng.uniform(0.5, 2.0, size=(n_items, 1)).astype(np.float32)         # rows are NOT unit length on disk
    with open(os.path.join(out_dir, 'articles_embeddings.pickle'), 'wb') as fh:
        pickle.dump(ace, fh)
    cols = list(meta.keys())
    with open(os.path.join(out_dir, 'articles_metadata.csv'), 'w') as fh:
        fh.write(','.join(cols) + '\n')
        for i in range(n_items):
            fh.write(','.join(str(int(meta[c][i])) for c in cols) + '\n')
    files, sid = [], 0
    for hour in range(n_hours):
        ss = make_sessions(sessions_per_hour, seq_len, n_items, scfg, seed, hour, length_dist, sid)
        sid += len(ss)
        path = os.path.join(out_dir, 'sessions_hour_%03d.tfrecord.gz' % hour)
        tfm.save_rows_to_tf_record_file(ss, scfg, path)
        files.append(path)
    return files, os.path.join(out_dir, 'articles_metadata.csv'), os.path.join(out_dir, 'articles_embeddings.pickle')


if __name__ == "__main__":      # python -m chameleon_recsys_amd.nar.synthetic --out /tmp/g1 --hours 8 --sessions-per-hour 2000
    import argparse
    ap = argparse.ArgumentParser(description="Write a synthetic G1-shaped dataset the NAR trainer can consume")
    ap.add_argument("--out", required=True)
    ap.add_argument("--hours", type=int, default=8)
    ap.add_argument("--sessions-per-hour", type=int, default=2000)
    ap.add_argument("--n-items", type=int, default=46000)
    ap.add_argument("--ace-dim", type=int, default=250)
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    files, csv_path, pkl_path = write_dataset(a.out, a.hours, a.sessions_per_hour, a.n_items, a.ace_dim, a.seq_len, a.seed)
    print("wrote %d session files, e.g. %s\n  articles metadata: %s\n  content embeddings: %s" % (len(files), files[0], csv_path, pkl_path))
    print("train with:\n  python -m chameleon_recsys_amd.nar.nar_trainer_gcom --train_set_path_regex '%s/sessions_hour_*.tfrecord.gz' "
          "--acr_module_articles_metadata_csv_path %s --acr_module_articles_content_embeddings_pickle_path %s --model_dir %s/model "
          "--batch_size 256 --truncate_session_length %d --learning_rate 1e-4 --reg_l2 1e-5 --softmax_temperature 0.1 "
          "--recent_clicks_buffer_max_size 20000 --recent_clicks_for_normalization 2000 --eval_metrics_top_n 5 --CAR_embedding_size 1024 "
          "--rnn_units 255 --train_total_negative_samples 50 --train_negative_samples_from_buffer 3000 --eval_total_negative_samples 50 "
          "--eval_negative_samples_from_buffer 3000 --content_embedding_scale_factor 6.0 --training_hours_for_each_eval 2 "
          "--disable_eval_benchmarks" % (a.out, csv_path, pkl_path, a.out, a.seq_len))
