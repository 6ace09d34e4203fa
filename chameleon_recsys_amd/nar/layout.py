"""Parameter layout of the NAR model in HBM: ONE flat fp32 buffer (regularised tensors first), padded shapes,
feature-column descriptors, and conversion to / from the reference's logical variable shapes.

Logical variable inventory: nar_module/nar/nar_model.py (scopes under ``main/``):
  :736-742 ``{name}_cat_embedding``, :911-919 ``items_embedding``, :890-898 ``gamma_scale/beta_center``,
  :375-388 PreCAR/CAR Dense, :1317 UGRNNCell kernel/bias, :411-426 FC1/FC2, :447-473 matching_dense_layer_1..4.
Padding rules (MI355X): feature widths -> multiples of 4 floats (16-byte rows for float4 / global_load_dwordx4),
rnn_units -> multiple of 128 (4 waves x 32-wide MFMA tiles); pad weights are 0 and provably stay 0.
"""
import math
from collections import OrderedDict

import numpy as np

ARTICLE_REQ_FEATURES = ['article_id', 'created_at_ts']          # nar_model.py:22
SESSION_REQ_SEQ_FEATURES = ['item_clicked', 'event_timestamp']  # nar_model.py:23

COL_ZERO, COL_OHE, COL_EMB, COL_NUM, COL_ACE, COL_ITEMEMB, COL_RECENCY, COL_NOVELTY = range(8)

DEFAULT_INTERNAL_FEATURES = {'recency': True, 'novelty': True, 'article_content_embeddings': True,
                             'item_clicked_embeddings': True}


def get_embedding_size(unique_val_count, const_mult=8):
    """nar_model.py:25-26."""
    return int(math.floor(const_mult * unique_val_count ** 0.25))


def ceil_to(x, m):
    return ((x + m - 1) // m) * m


class Entry:
    __slots__ = ("name", "shape", "offset", "size", "reg", "init", "fan")

    def __init__(self, name, shape, reg, init, fan=None):
        self.name, self.shape, self.reg, self.init, self.fan = name, tuple(shape), reg, init, fan
        self.size = ceil_to(int(np.prod(shape)), 4)
        self.offset = -1


class ParamLayout:
    def __init__(self, session_features_config, articles_features_config, n_items, ace_dim, CAR_embedding_size,
                 rnn_units, rnn_num_layers=1, rnn_cell='ugrnn', internal_features_config=None,
                 max_cardinality_for_ohe=10):
        ifc = dict(DEFAULT_INTERNAL_FEATURES)
        if internal_features_config:
            ifc.update(internal_features_config)
        self.ifc = ifc
        self.C = C = CAR_embedding_size
        if C % 4:
            raise ValueError("CAR_embedding_size must be a multiple of 4")
        self.H = H = rnn_units
        self.Hp = Hp = ceil_to(H, 128)
        if Hp > 512 and rnn_cell != 'ugrnn':
            raise ValueError("rnn_units > 384 is not supported by the GRU recurrent kernel")
        self.rnn_stepwise = Hp > 512            # UGRNN beyond the fused kernel's LDS budget: GEMM + pointwise kernel per time step
        self.L = rnn_num_layers
        self.cell = rnn_cell
        if rnn_cell not in ('ugrnn', 'gru'):
            raise ValueError("rnn_cell=%r: 'ugrnn' (the reference's cell, nar_model.py:1317) or 'gru'" % rnn_cell)
        if rnn_cell == 'gru' and Hp > 384:
            raise ValueError("rnn_units > 384 is not supported by the GRU recurrent kernel (LDS budget)")
        self.NG = 2 if rnn_cell == 'ugrnn' else 3          # gate/candidate column blocks of the input projection
        self.n_items = n_items
        self.D = ace_dim
        scfg = session_features_config['sequence_features']
        acfg = articles_features_config
        self.max_ohe = max_cardinality_for_ohe

        emb, ctx_cols, item_cols = [], [], []
        self.ctx_cat_names, self.ctx_num_names, self.meta_names = [], [], []
        self.meta_is_float = set()
        # ---- user-context columns, nar_model.py:746-767 (dict insertion order)
        for name, cfg in scfg.items():
            if name in SESSION_REQ_SEQ_FEATURES:
                continue
            if cfg['type'] == 'categorical':
                fi = len(self.ctx_cat_names)
                self.ctx_cat_names.append(name)
                card = cfg['cardinality']
                if card <= max_cardinality_for_ohe:
                    ctx_cols += [(COL_OHE, fi, s, card, None) for s in range(card)]
                else:
                    dim = get_embedding_size(card)
                    e = Entry('ctx_emb/' + name, (card, dim), True, 'xavier')
                    emb.append(e)
                    ctx_cols += [(COL_EMB, fi, s, dim, e) for s in range(dim)]
            elif cfg['type'] == 'numerical':
                fi = len(self.ctx_num_names)
                self.ctx_num_names.append(name)
                ctx_cols.append((COL_NUM, fi, 0, 1, None))
            else:
                raise Exception('Invalid feature type: {}'.format(name))     # nar_model.py:760
        if not ctx_cols:
            ctx_cols.append((COL_ZERO, 0, 0, 1, None))                        # nar_model.py:323-325
        # ---- item columns, nar_model.py:926-990
        for name, cfg in acfg.items():
            if name in ARTICLE_REQ_FEATURES:
                continue
            fi = len(self.meta_names)
            self.meta_names.append(name)
            if cfg['type'] == 'categorical':
                card = cfg['cardinality']
                if card <= max_cardinality_for_ohe:
                    item_cols += [(COL_OHE, fi, s, card, None) for s in range(card)]
                else:
                    dim = get_embedding_size(card)
                    e = Entry('meta_emb/' + name, (card, dim), True, 'xavier')
                    emb.append(e)
                    item_cols += [(COL_EMB, fi, s, dim, e) for s in range(dim)]
            elif cfg['type'] == 'numerical':
                is_float = cfg.get('dtype') == 'float'          # stored as float32 bits in the int64 metadata table (NARRuntime)
                if is_float:
                    self.meta_is_float.add(name)
                item_cols.append((COL_NUM, fi, 1 if is_float else 0, 1, None))
            else:
                raise Exception('Invalid feature type: {}'.format(name))
        if ifc['article_content_embeddings']:
            item_cols += [(COL_ACE, 0, s, self.D, None) for s in range(self.D)]
        if ifc['item_clicked_embeddings']:
            # (items_embedding_size: test aid - the embedding width of a larger catalog on a small one, tests/test_config5_parity_gpu.py;
            # the reference always derives it from the vocabulary size, nar_model.py:911-919)
            E = int(ifc.get('items_embedding_size') or get_embedding_size(n_items))
            e = Entry('items_embedding', (n_items, E), True, 'xavier')
            emb.append(e)
            item_cols += [(COL_ITEMEMB, 0, s, E, e) for s in range(E)]
        if ifc['recency']:
            item_cols.append((COL_RECENCY, 0, 0, 1, None))
        if ifc['novelty']:
            item_cols.append((COL_NOVELTY, 0, 0, 1, None))
        self.f_ctx, self.f_item = len(ctx_cols), len(item_cols)
        self.F = self.f_ctx + self.f_item
        self.Fc, self.Fi = ceil_to(self.f_ctx, 4), ceil_to(self.f_item, 4)
        ctx_cols += [(COL_ZERO, 0, 0, 1, None)] * (self.Fc - self.f_ctx)
        item_cols += [(COL_ZERO, 0, 0, 1, None)] * (self.Fi - self.f_item)
        self._ctx_cols, self._item_cols = ctx_cols, item_cols

        Fc, Fi = self.Fc, self.Fi
        ents = list(emb)
        self.n_emb_entries = len(emb)
        ents += [Entry('gamma_ctx', (Fc,), True, 'gamma'), Entry('beta_ctx', (Fc,), True, 'zeros'),
                 Entry('gamma_item', (Fi,), True, 'gamma'), Entry('beta_item', (Fi,), True, 'zeros'),
                 Entry('W1c', (Fc, C), True, 'pad'), Entry('W1i', (Fi, C), True, 'pad'),
                 Entry('W2', (C, C), True, 'pad'), Entry('Wf1', (Hp, 512), True, 'pad'), Entry('Wf2', (512, C), True, 'pad'),
                 Entry('Ws1', (C, 128), True, 'pad'), Entry('Ws2', (128, 64), True, 'pad'), Entry('Ws3', (64, 32), True, 'pad'),
                 Entry('Ws4', (32,), True, 'pad')]
        ents += [Entry('b1', (C,), False, 'zeros'), Entry('b2', (C,), False, 'zeros')]
        for l in range(self.L):
            Ip = C if l == 0 else Hp
            ents += [Entry('rnn%d/Wx' % l, (Ip, self.NG * Hp), False, 'pad'), Entry('rnn%d/Wh' % l, (Hp, 2 * Hp), False, 'pad')]
            if rnn_cell == 'gru':    # W_ch directly behind W_gh: the recurrent kernel takes ONE pointer (chameleon_nar.h)
                ents.append(Entry('rnn%d/Wch' % l, (Hp, Hp), False, 'pad'))
            ents.append(Entry('rnn%d/b' % l, (self.NG * Hp,), False, 'zeros'))
        ents += [Entry('bf1', (512,), False, 'zeros'), Entry('bf2', (C,), False, 'zeros'), Entry('bs1', (128,), False, 'zeros'),
                 Entry('bs2', (64,), False, 'zeros'), Entry('bs3', (32,), False, 'zeros'), Entry('bs4', (1,), False, 'zeros')]
        off = 0
        self.entries = OrderedDict()
        for e in ents:
            e.offset = off
            off += e.size
            self.entries[e.name] = e
        self.total = ceil_to(off, 256)                          # pad: splits evenly over 1..64 data-parallel ranks, float4 slices
        self.n_reg = sum(e.size for e in ents if e.reg)
        self.emb_end = sum(e.size for e in emb)                 # embedding-gradient region [0, emb_end)
        # regularised entries must be a prefix
        seen_nonreg = False
        for e in ents:
            if not e.reg:
                seen_nonreg = True
            elif seen_nonreg:
                raise AssertionError("regularised tensors must come first")

    def fingerprint(self):
        """Hash of (name, shape, offset) of every tensor of the flat buffer + cell type: stored in checkpoints."""
        import hashlib
        h = hashlib.sha1(("%s|%d|" % (self.cell, self.total)).encode())
        for e in self.entries.values():
            h.update(("%s:%s@%d;" % (e.name, "x".join(map(str, e.shape)), e.offset)).encode())
        return h.hexdigest()

    # ------------------------------------------------------------------ descriptors
    def _desc(self, cols):
        d = np.zeros((len(cols), 5), dtype=np.int64)
        for i, (kind, feat, sub, dim, ent) in enumerate(cols):
            d[i] = (kind, feat, sub, dim, ent.offset if ent is not None else 0)
        return d

    def ctx_descriptors(self):
        return self._desc(self._ctx_cols)

    @staticmethod
    def _emb_groups(cols):
        """Embedding column groups of a feature matrix: (kind, feat, first column, dim, table rows, flat offset of the table)."""
        out, seen = [], set()
        for c, (kind, feat, sub, dim, ent) in enumerate(cols):
            if kind in (COL_EMB, COL_ITEMEMB) and ent.name not in seen:
                seen.add(ent.name)
                out.append((kind, feat, c - sub, dim, ent.shape[0], ent.offset))
        return out

    def item_segments(self):
        """(segs [n, 6] int64, singles int32) for cham_item_assemble_lds: runs of item columns that come from ONE contiguous source
        row - kind 0 ACE, 1 item embedding, 2 metadata embedding - as {kind, dst column, length, table offset, row pitch, feature}; the
        remaining columns (one-hot bits, numerics, recency, novelty, zero padding) are listed in `singles`."""
        segs, singles, c = [], [], 0
        cols = self._item_cols
        while c < len(cols):
            kind, feat, sub, dim, ent = cols[c]
            if kind == COL_ACE and sub == 0:
                segs.append((0, c, dim, 0, dim, 0)); c += dim
            elif kind == COL_ITEMEMB and sub == 0:
                segs.append((1, c, dim, ent.offset, dim, 0)); c += dim
            elif kind == COL_EMB and sub == 0:
                segs.append((2, c, dim, ent.offset, dim, feat)); c += dim
            else:
                singles.append(c); c += 1
        return (np.asarray(segs, dtype=np.int64).reshape(-1, 6), np.asarray(singles, dtype=np.int32))

    def ctx_emb_groups(self):
        return self._emb_groups(self._ctx_cols)

    def item_emb_groups(self):
        return self._emb_groups(self._item_cols)

    def item_descriptors(self):
        return self._desc(self._item_cols)

    # ------------------------------------------------------------------ logical <-> flat
    def logical_specs(self):
        """(name -> (shape, init, reg)) in the oracle's / reference's logical variable space."""
        C, H = self.C, self.H
        specs = OrderedDict()
        for name, e in self.entries.items():
            if name.startswith(('ctx_emb/', 'meta_emb/')) or name == 'items_embedding':
                specs[name] = (e.shape, 'xavier', True)
        specs['gamma'] = ((self.F,), 'ones', True)
        specs['beta'] = ((self.F,), 'zeros', True)
        specs['PreCAR/kernel'] = ((self.F, C), 'variance_scaling', True)
        specs['PreCAR/bias'] = ((C,), 'zeros', False)
        specs['CAR/kernel'] = ((C, C), 'xavier', True)
        specs['CAR/bias'] = ((C,), 'zeros', False)
        for l in range(self.L):
            I = C if l == 0 else H
            if self.cell == 'ugrnn':
                specs['rnn/%d/kernel' % l] = ((I + H, 2 * H), 'xavier', False)
                specs['rnn/%d/bias' % l] = ((2 * H,), 'zeros', False)
            else:     # tf.nn.rnn_cell.GRUCell variables: gates (bias init 1.0) and candidate
                specs['rnn/%d/gates/kernel' % l] = ((I + H, 2 * H), 'xavier', False)
                specs['rnn/%d/gates/bias' % l] = ((2 * H,), 'ones', False)
                specs['rnn/%d/candidate/kernel' % l] = ((I + H, H), 'xavier', False)
                specs['rnn/%d/candidate/bias' % l] = ((H,), 'zeros', False)
        specs['FC1/kernel'] = ((H, 512), 'variance_scaling', True)
        specs['FC1/bias'] = ((512,), 'zeros', False)
        specs['FC2/kernel'] = ((512, C), 'xavier', True)
        specs['FC2/bias'] = ((C,), 'zeros', False)
        dims = [C, 128, 64, 32, 1]
        inits = ['variance_scaling', 'variance_scaling', 'variance_scaling', 'lecun_uniform']
        for i in range(4):
            specs['match%d/kernel' % (i + 1)] = ((dims[i], dims[i + 1]), inits[i], True)
            specs['match%d/bias' % (i + 1)] = ((dims[i + 1],), 'zeros', False)
        return specs

    # ------------------------------------------------------------------ TF variable names
    def tf_variable_names(self):
        """OrderedDict: TensorFlow variable name of the reference graph -> logical tensor name (same shapes: the logical tensors ARE the
        reference's variables - un-padded, the UGRNN kernel re-assembled to [I + H, 2 H] from the Wx / Wh blocks of the flat buffer by
        unpack()).  So that weights can be cross-loaded from / exported to a real TF 1.12 checkpoint of nar_module (tf.train.load_variable
        / tf.train.list_variables give name -> array), should one ever be available; TF itself cannot run here.
        Scopes follow nar_model.py: "main" (:210) / "user_items_contextual_features" (:314) / "features" (:744) /
        "{name}_cat_embedding/{name}_embedding" (:737-739); "item_features" (:922) / "item_cat_embedding/items_embedding" (:913-916);
        "input_features_center_scale/{gamma_scale, beta_center}" (:890-895); "CAR/{PreCAR,CAR}_representation" (:374-387);
        "RNN/rnn/multi_rnn_cell/cell_l/ugrnn_cell" (:1309-1342; tf.nn.rnn_cell.GRUCell: "gru_cell/{gates, candidate}");
        "session_representation/{FC1, FC2}" (:410-425); "recommendations_ranking/matching_dense_layer_{1..4}" (:444-472).
        tf.layers.Dense OBJECTS bind their variable scope at the first __call__, not at construction (TF 1.x base Layer._set_scope):
        PreCAR_dense is built AND first called inside "CAR" (:372-381) -> "main/CAR/PreCAR_representation"; CAR_dense is built there
        but first called under "user_personalized_contextual_article_embedding/input" (:388-390); the four matching layers are built
        under "recommendations_ranking" (:444-471) and first called under its "cos_sim_positive" (:476-486).  The first-call spelling
        is canonical (what a TF 1.12 checkpoint reader lists); tf_variable_aliases() keeps the construction-scope spelling, which
        from_tf_variables() also accepts (TF cannot run here to confirm)."""
        m = OrderedDict()
        ucf = 'main/user_items_contextual_features/'
        for name in self.entries:
            if name.startswith('ctx_emb/'):
                f = name.split('/', 1)[1]
                m[ucf + 'features/%s_cat_embedding/%s_embedding' % (f, f)] = name
            elif name.startswith('meta_emb/'):
                f = name.split('/', 1)[1]
                m[ucf + 'item_features/features/%s_cat_embedding/%s_embedding' % (f, f)] = name
            elif name == 'items_embedding':
                m[ucf + 'item_features/item_cat_embedding/items_embedding'] = name
        m[ucf + 'input_features_center_scale/gamma_scale'] = 'gamma'
        m[ucf + 'input_features_center_scale/beta_center'] = 'beta'
        for v in ('kernel', 'bias'):
            m['main/CAR/PreCAR_representation/' + v] = 'PreCAR/' + v
            m['main/user_personalized_contextual_article_embedding/input/CAR_representation/' + v] = 'CAR/' + v
        for l in range(self.L):
            cell = 'main/RNN/rnn/multi_rnn_cell/cell_%d/' % l
            if self.cell == 'ugrnn':
                m[cell + 'ugrnn_cell/kernel'] = 'rnn/%d/kernel' % l
                m[cell + 'ugrnn_cell/bias'] = 'rnn/%d/bias' % l
            else:
                for part in ('gates', 'candidate'):
                    m[cell + 'gru_cell/%s/kernel' % part] = 'rnn/%d/%s/kernel' % (l, part)
                    m[cell + 'gru_cell/%s/bias' % part] = 'rnn/%d/%s/bias' % (l, part)
        for fc in ('FC1', 'FC2'):
            m['main/session_representation/%s/kernel' % fc] = fc + '/kernel'
            m['main/session_representation/%s/bias' % fc] = fc + '/bias'
        for i in range(1, 5):
            m['main/recommendations_ranking/cos_sim_positive/matching_dense_layer_%d/kernel' % i] = 'match%d/kernel' % i
            m['main/recommendations_ranking/cos_sim_positive/matching_dense_layer_%d/bias' % i] = 'match%d/bias' % i
        assert set(m.values()) == set(self.logical_specs())
        return m

    def tf_variable_aliases(self):
        """Alternative spelling -> canonical TF name: the CONSTRUCTION-scope spelling of the Dense layers whose first call happens in a
        nested scope (CAR_dense: built in "CAR", nar_model.py:383-387; the matching layers: built in "recommendations_ranking", :447-471)."""
        a = OrderedDict()
        for v in ('kernel', 'bias'):
            a['main/CAR/CAR_representation/' + v] = 'main/user_personalized_contextual_article_embedding/input/CAR_representation/' + v
        for i in range(1, 5):
            for v in ('kernel', 'bias'):
                a['main/recommendations_ranking/matching_dense_layer_%d/%s' % (i, v)] = \
                    'main/recommendations_ranking/cos_sim_positive/matching_dense_layer_%d/%s' % (i, v)
        return a

    def to_tf_variables(self, logical):
        """logical dict (unpack / logical_weights) -> {TF variable name: array of the reference's shape}."""
        return OrderedDict((tf_name, np.asarray(logical[lg])) for tf_name, lg in self.tf_variable_names().items())

    def from_tf_variables(self, tf_vars, strict=True):
        """{TF variable name (with or without ':0', canonical or alias spelling): array} -> logical dict for pack() /
        NARRuntime.load_logical_weights.  Optimizer slots (".../Adam", ".../Adam_1"), global_step and beta power accumulators of a real
        checkpoint are ignored; strict: every model variable must be present with the reference's shape."""
        names, alias, specs = self.tf_variable_names(), self.tf_variable_aliases(), self.logical_specs()
        out = OrderedDict()
        for k, v in tf_vars.items():
            k = k[:-2] if k.endswith(':0') else k
            k = alias.get(k, k)
            if k not in names:
                continue
            lg = names[k]
            v = np.asarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(specs[lg][0]):
                raise ValueError("TF variable %s has shape %r, the model expects %r" % (k, tuple(v.shape), tuple(specs[lg][0])))
            out[lg] = v
        missing = [t for t, lg in names.items() if lg not in out]
        if strict and missing:
            raise KeyError("TF checkpoint lacks %d model variables, e.g. %s" % (len(missing), missing[:3]))
        return OrderedDict((k, out[k]) for k in specs if k in out)

    def init_logical(self, seed=42, max_random_elems=None):
        """TF-1.12 initialiser distributions (xavier_initializer, variance_scaling_initializer(), lecun_uniform;
        nar_model.py:210, 377, 413, 449-470).  TF's own random streams are not reproducible."""
        rng = np.random.default_rng(seed)
        out = OrderedDict()
        for name, (shape, init, _) in self.logical_specs().items():
            if init == 'zeros' or (max_random_elems is not None and int(np.prod(shape)) > max_random_elems):
                w = np.zeros(shape, np.float32)      # (stress configurations: multi-GB tables start at zero instead of xavier)
            elif init == 'ones':
                w = np.ones(shape, np.float32)
            else:
                fan_in, fan_out = shape[0], shape[1]
                if init == 'xavier':
                    lim = math.sqrt(6.0 / (fan_in + fan_out))
                    w = rng.uniform(-lim, lim, size=shape)
                elif init == 'lecun_uniform':
                    lim = math.sqrt(3.0 / fan_in)
                    w = rng.uniform(-lim, lim, size=shape)
                else:   # variance_scaling_initializer(): factor 2, FAN_IN, truncated normal (2 sigma)
                    std = math.sqrt(1.3 * 2.0 / fan_in)
                    w = rng.normal(0, std, size=shape)
                    bad = np.abs(w) > 2 * std
                    while bad.any():
                        w[bad] = rng.normal(0, std, size=int(bad.sum()))
                        bad = np.abs(w) > 2 * std
                w = w.astype(np.float32)
            out[name] = w
        return out

    def _view(self, flat, name):
        e = self.entries[name]
        return flat[e.offset:e.offset + int(np.prod(e.shape))].reshape(e.shape)

    def pack(self, logical):
        """logical dict -> flat padded numpy buffer."""
        C, H, Hp, fc, fi = self.C, self.H, self.Hp, self.f_ctx, self.f_item
        flat = np.zeros(self.total, np.float32)
        v = lambda n: self._view(flat, n)
        for name in self.entries:
            if name.startswith(('ctx_emb/', 'meta_emb/')) or name == 'items_embedding':
                v(name)[...] = logical[name]
        g, b = np.asarray(logical['gamma']), np.asarray(logical['beta'])
        v('gamma_ctx')[:fc] = g[:fc]; v('gamma_item')[:fi] = g[fc:]
        v('beta_ctx')[:fc] = b[:fc]; v('beta_item')[:fi] = b[fc:]
        W1 = np.asarray(logical['PreCAR/kernel'])
        v('W1c')[:fc] = W1[:fc]; v('W1i')[:fi] = W1[fc:]
        v('b1')[...] = logical['PreCAR/bias']
        v('W2')[...] = logical['CAR/kernel']; v('b2')[...] = logical['CAR/bias']
        for l in range(self.L):
            I = C if l == 0 else H
            Wx, Wh, rb = v('rnn%d/Wx' % l), v('rnn%d/Wh' % l), v('rnn%d/b' % l)
            if self.cell == 'ugrnn':
                K = np.asarray(logical['rnn/%d/kernel' % l]); bb = np.asarray(logical['rnn/%d/bias' % l])
            else:
                K = np.asarray(logical['rnn/%d/gates/kernel' % l]); bb = np.asarray(logical['rnn/%d/gates/bias' % l])
                Kc = np.asarray(logical['rnn/%d/candidate/kernel' % l]); bc = np.asarray(logical['rnn/%d/candidate/bias' % l])
                Wx[:I, 2 * Hp:2 * Hp + H] = Kc[:I]
                v('rnn%d/Wch' % l)[:H, :H] = Kc[I:]
                rb[2 * Hp:2 * Hp + H] = bc
            Wx[:I, :H] = K[:I, :H]; Wx[:I, Hp:Hp + H] = K[:I, H:]
            Wh[:H, :H] = K[I:, :H]; Wh[:H, Hp:Hp + H] = K[I:, H:]
            rb[:H] = bb[:H]; rb[Hp:Hp + H] = bb[H:]
        v('Wf1')[:H] = logical['FC1/kernel']; v('bf1')[...] = logical['FC1/bias']
        v('Wf2')[...] = logical['FC2/kernel']; v('bf2')[...] = logical['FC2/bias']
        for i, (w, bn) in enumerate([('Ws1', 'bs1'), ('Ws2', 'bs2'), ('Ws3', 'bs3')]):
            v(w)[...] = logical['match%d/kernel' % (i + 1)]; v(bn)[...] = logical['match%d/bias' % (i + 1)]
        v('Ws4')[...] = np.asarray(logical['match4/kernel']).reshape(32)
        v('bs4')[...] = np.asarray(logical['match4/bias']).reshape(1)
        return flat

    def unpack(self, flat):
        """flat padded numpy buffer -> logical dict (works for params, grads, Adam slots alike)."""
        C, H, Hp, fc, fi = self.C, self.H, self.Hp, self.f_ctx, self.f_item
        flat = np.asarray(flat)
        v = lambda n: self._view(flat, n)
        out = OrderedDict()
        for name in self.entries:
            if name.startswith(('ctx_emb/', 'meta_emb/')) or name == 'items_embedding':
                out[name] = v(name).copy()
        out['gamma'] = np.concatenate([v('gamma_ctx')[:fc], v('gamma_item')[:fi]])
        out['beta'] = np.concatenate([v('beta_ctx')[:fc], v('beta_item')[:fi]])
        out['PreCAR/kernel'] = np.concatenate([v('W1c')[:fc], v('W1i')[:fi]], 0)
        out['PreCAR/bias'] = v('b1').copy()
        out['CAR/kernel'] = v('W2').copy(); out['CAR/bias'] = v('b2').copy()
        for l in range(self.L):
            I = C if l == 0 else H
            Wx, Wh, rb = v('rnn%d/Wx' % l), v('rnn%d/Wh' % l), v('rnn%d/b' % l)
            top = np.concatenate([Wx[:I, :H], Wx[:I, Hp:Hp + H]], 1)
            bot = np.concatenate([Wh[:H, :H], Wh[:H, Hp:Hp + H]], 1)
            if self.cell == 'ugrnn':
                out['rnn/%d/kernel' % l] = np.concatenate([top, bot], 0)
                out['rnn/%d/bias' % l] = np.concatenate([rb[:H], rb[Hp:Hp + H]])
            else:
                out['rnn/%d/gates/kernel' % l] = np.concatenate([top, bot], 0)
                out['rnn/%d/gates/bias' % l] = np.concatenate([rb[:H], rb[Hp:Hp + H]])
                out['rnn/%d/candidate/kernel' % l] = np.concatenate([Wx[:I, 2 * Hp:2 * Hp + H], v('rnn%d/Wch' % l)[:H, :H]], 0)
                out['rnn/%d/candidate/bias' % l] = rb[2 * Hp:2 * Hp + H].copy()
        out['FC1/kernel'] = v('Wf1')[:H].copy(); out['FC1/bias'] = v('bf1').copy()
        out['FC2/kernel'] = v('Wf2').copy(); out['FC2/bias'] = v('bf2').copy()
        for i, (w, bn) in enumerate([('Ws1', 'bs1'), ('Ws2', 'bs2'), ('Ws3', 'bs3')]):
            out['match%d/kernel' % (i + 1)] = v(w).copy(); out['match%d/bias' % (i + 1)] = v(bn).copy()
        out['match4/kernel'] = v('Ws4').reshape(32, 1).copy(); out['match4/bias'] = v('bs4').reshape(1).copy()
        # reorder to the logical spec order
        return OrderedDict((k, out[k]) for k in self.logical_specs())
