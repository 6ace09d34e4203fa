"""Dense float32 (optionally float64) CPU restatement of the NAR graph (forward, loss, gradients, TF-Adam).

TEST INFRASTRUCTURE (see oracle/__init__.py) - also the "port" CPU baseline timed by bench.py.
PARITY UNPINNED at the TF boundary: TensorFlow 1.12 cannot run here, the reference has no golden
vectors; this file follows nar_module/nar/nar_model.py line by line (citations inline) in the
reference's op order: NO de-duplication, NO PreCAR factorisation, padded rows computed.

Gradients come from PyTorch-CPU autograd over this restatement; the optimizer is a hand-written
TF-flavoured Adam (nar_model.py:708-722; tf.train.AdamOptimizer semantics).

Two adjudication modes (round 5, tests/golden/loss_curve_200.npz, oracle/make_loss_curve.py):
  * ``dtype=torch.float64``: weights, activations, gradients and Adam slots in float64 - the trajectory every fp32
    implementation (this oracle in fp32, TensorFlow's Eigen kernels, the HIP path) is a rounding of.  What the GRAPH
    itself quantises stays quantised: the int64 -> float32 time-stamp casts (nar_model.py:1058-1059) and the float32
    popularity placeholder (:1442) are rounded to float32 first, then widened.
  * ``sum_perm_seed=k``: the same fp32 arithmetic with every contraction summed in another order (a fixed permutation of
    the K index of each matmul) - a second, equally correct fp32 realisation.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import philox
from . import sampler as osampler

ARTICLE_REQ_FEATURES = ['article_id', 'created_at_ts']          # nar_model.py:22
SESSION_REQ_SEQ_FEATURES = ['item_clicked', 'event_timestamp']  # nar_model.py:23
MS_PER_DAY = 1000.0 * 60.0 * 60.0 * 24.0


def get_embedding_size(unique_val_count, const_mult=8):
    """nar_model.py:25-26."""
    return int(math.floor(const_mult * unique_val_count ** 0.25))


class _BF16MatMul(torch.autograd.Function):
    """a @ b with both operands rounded to bf16 (round-to-nearest-even) and fp32 accumulation, forward AND backward -
    the arithmetic of the HIP path's bf16 GEMM mode (BASELINE config 3; csrc/gemm.hip gemm_bf16_kernel)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return a.bfloat16().float() @ b.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gb = g.bfloat16().float()
        ga = gb @ b.bfloat16().float().transpose(-1, -2)
        a2 = a.bfloat16().float().reshape(-1, a.shape[-1])
        return ga, a2.t() @ gb.reshape(-1, g.shape[-1])


class _StoreBF16(torch.autograd.Function):
    """Value stored as bf16 in HBM (round to nearest even), gradient passed through: the bf16 configuration keeps the matrices with
    one row per candidate (PreCAR / CAR / scorer activations) in bf16 (csrc/gemm_b16.hip)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _leaky(x):
    return torch.nn.functional.leaky_relu(x, 0.2)   # tf.nn.leaky_relu default alpha=0.2


# ----------------------------------------------------------------------------- parameters
def param_specs(params):
    """Ordered (name -> (shape, init, regularised)) of every trainable variable."""
    scfg = params['session_features_config']['sequence_features']
    acfg = params['articles_features_config']
    ifc = params.get('internal_features_config',
                     dict(recency=True, novelty=True, article_content_embeddings=True,
                          item_clicked_embeddings=True))
    max_ohe = params.get('max_cardinality_for_ohe', 10)
    C = params['CAR_embedding_size']
    H = params['rnn_units']
    L = params.get('rnn_num_layers', 1)
    cell = params.get('rnn_cell', 'ugrnn')
    D = params['content_article_embeddings_matrix'].shape[1]
    n_items = acfg['article_id'].get('cardinality', params['content_article_embeddings_matrix'].shape[0])
    specs = OrderedDict()
    f_ctx = 0
    for name, cfg in scfg.items():                                  # nar_model.py:746-767
        if name in SESSION_REQ_SEQ_FEATURES:
            continue
        if cfg['type'] == 'categorical':
            if cfg['cardinality'] <= max_ohe:
                f_ctx += cfg['cardinality']
            else:
                dim = get_embedding_size(cfg['cardinality'])
                specs['ctx_emb/' + name] = ((cfg['cardinality'], dim), 'xavier', True)
                f_ctx += dim
        elif cfg['type'] == 'numerical':
            f_ctx += 1
        else:
            raise Exception('Invalid feature type: {}'.format(name))
    if f_ctx == 0:
        f_ctx = 1                                                  # nar_model.py:323-325 dummy zero column
    f_item = 0
    for name, cfg in acfg.items():                                  # nar_model.py:926-939
        if name in ARTICLE_REQ_FEATURES:
            continue
        if cfg['type'] == 'categorical':
            if cfg['cardinality'] <= max_ohe:
                f_item += cfg['cardinality']
            else:
                dim = get_embedding_size(cfg['cardinality'])
                specs['meta_emb/' + name] = ((cfg['cardinality'], dim), 'xavier', True)
                f_item += dim
        else:
            f_item += 1
    if ifc['article_content_embeddings']:
        f_item += D
    if ifc['item_clicked_embeddings']:
        E = int(ifc.get('items_embedding_size') or get_embedding_size(n_items))      # (override: test aid, see nar/layout.py)
        specs['items_embedding'] = ((n_items, E), 'xavier', True)  # nar_model.py:911-919
        f_item += E
    f_item += int(ifc['recency']) + int(ifc['novelty'])
    F = f_ctx + f_item
    specs['gamma'] = ((F,), 'ones', True)                          # nar_model.py:891-898
    specs['beta'] = ((F,), 'zeros', True)
    specs['PreCAR/kernel'] = ((F, C), 'variance_scaling', True)    # :375-380
    specs['PreCAR/bias'] = ((C,), 'zeros', False)
    specs['CAR/kernel'] = ((C, C), 'xavier', True)                 # :384-388
    specs['CAR/bias'] = ((C,), 'zeros', False)
    for l in range(L):                                             # :1314-1338
        I = C if l == 0 else H
        if cell == 'ugrnn':
            specs['rnn/%d/kernel' % l] = ((I + H, 2 * H), 'xavier', False)
            specs['rnn/%d/bias' % l] = ((2 * H,), 'zeros', False)
        elif cell == 'gru':
            specs['rnn/%d/gates/kernel' % l] = ((I + H, 2 * H), 'xavier', False)
            specs['rnn/%d/gates/bias' % l] = ((2 * H,), 'ones', False)
            specs['rnn/%d/candidate/kernel' % l] = ((I + H, H), 'xavier', False)
            specs['rnn/%d/candidate/bias' % l] = ((H,), 'zeros', False)
        else:
            raise ValueError(cell)
    specs['FC1/kernel'] = ((H, 512), 'variance_scaling', True)     # :411-416
    specs['FC1/bias'] = ((512,), 'zeros', False)
    specs['FC2/kernel'] = ((512, C), 'xavier', True)               # :423-426
    specs['FC2/bias'] = ((C,), 'zeros', False)
    dims = [C, 128, 64, 32, 1]                                     # :447-473
    inits = ['variance_scaling', 'variance_scaling', 'variance_scaling', 'lecun_uniform']
    for i in range(4):
        specs['match%d/kernel' % (i + 1)] = ((dims[i], dims[i + 1]), inits[i], True)
        specs['match%d/bias' % (i + 1)] = ((dims[i + 1],), 'zeros', False)
    return specs, dict(f_ctx=f_ctx, f_item=f_item, F=F)


def init_params(params, seed=42):
    """TF-1.12 initialiser *distributions* (SURVEY A.5); streams cannot match TF."""
    g = torch.Generator().manual_seed(seed)
    specs, _ = param_specs(params)
    out = OrderedDict()
    for name, (shape, init, _) in specs.items():
        if init == 'zeros':
            t = torch.zeros(shape)
        elif init == 'ones':
            t = torch.ones(shape)
        else:
            fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
            if init == 'xavier':
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                t = (torch.rand(shape, generator=g) * 2 - 1) * lim
            elif init == 'lecun_uniform':
                lim = math.sqrt(3.0 / fan_in)
                t = (torch.rand(shape, generator=g) * 2 - 1) * lim
            elif init == 'variance_scaling':
                std = math.sqrt(1.3 * 2.0 / fan_in)
                t = torch.empty(shape)
                torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)
            else:
                raise ValueError(init)
        out[name] = t.float()
    return out


# ----------------------------------------------------------------------------- model
class NAROracle:
    def __init__(self, params, weights=None, seed=42, dtype=torch.float32, sum_perm_seed=None):
        self.p = params
        self.dt = dtype
        self.sum_perm_seed, self._perms = sum_perm_seed, {}
        self.specs, self.dims = param_specs(params)
        w = weights if weights is not None else init_params(params, seed)
        self.w = OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=dtype).clone().requires_grad_(True))
                             for k, v in w.items())
        assert list(self.w.keys()) == list(self.specs.keys())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.w.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.w.items())
        self.global_step = 0
        self.ace = torch.as_tensor(np.asarray(params['content_article_embeddings_matrix'], dtype=np.float32)).to(dtype)
        self.meta = {k: torch.as_tensor(np.asarray(v)) for k, v in params['articles_metadata'].items()}
        self.ifc = params.get('internal_features_config',
                              dict(recency=True, novelty=True, article_content_embeddings=True,
                                   item_clicked_embeddings=True))
        self.max_ohe = params.get('max_cardinality_for_ohe', 10)
        self.cell = params.get('rnn_cell', 'ugrnn')
        self.seed = params.get('tf_random_seed', 42)
        self.gemm_dtype = params.get('gemm_dtype', 'f32')
        self._train, self._step = False, 0          # set by forward(): dropout is active in TRAIN mode only

    def _c(self, x):
        """Scalar constant in the working dtype."""
        return torch.tensor(x, dtype=self.dt)

    def _perm(self, K):
        pm = self._perms.get(K)
        if pm is None:
            g = torch.Generator().manual_seed(1000003 * int(self.sum_perm_seed) + K)
            pm = self._perms[K] = torch.randperm(K, generator=g)
        return pm

    def _mmf(self, a, b):
        """a @ b in the working dtype; with ``sum_perm_seed`` the contraction index is visited in a permuted order (same
        real-number product, another fp32 summation order)."""
        if self.sum_perm_seed is None:
            return a @ b
        pm = self._perm(a.shape[-1])
        return a[..., pm] @ b[pm]

    def _mm(self, a, b):
        """Dense / matmul of the graph; bf16-rounded operands when the bf16 compute mode is emulated."""
        return _BF16MatMul.apply(a, b) if self.gemm_dtype == 'bf16' else self._mmf(a, b)

    # -- nar_model.py:730-773 get_features
    def _get_features(self, values, config, ignore, prefix):
        feats = []
        for name, cfg in config.items():
            if name in ignore:
                continue
            x = values[name]
            if cfg['type'] == 'categorical':
                if cfg['cardinality'] <= self.max_ohe:
                    feats.append(torch.nn.functional.one_hot(x.long(), cfg['cardinality']).to(self.dt))
                else:
                    feats.append(self.w[prefix + name][x.long()])
            elif cfg['type'] == 'numerical':
                feats.append(x.float().to(self.dt).unsqueeze(-1))      # float32 feature column, widened in f64 mode
            else:
                raise Exception('Invalid feature type: {}'.format(name))
        return torch.cat(feats, dim=-1) if feats else None

    # -- nar_model.py:1011-1039, 996-1009
    @staticmethod
    def _normalize_values(x, stats):
        mean = stats.mean()
        var = ((stats - mean) ** 2).mean()                 # tf.nn.moments: population variance
        sd = torch.sqrt(var + torch.tensor(1e-24, dtype=x.dtype))
        z = (x - mean) / sd
        zs = (stats - mean) / sd
        eps = torch.tensor(1e-24, dtype=x.dtype)
        mn, mx = zs.min(), zs.max()
        scaled = (z - mn + eps) / torch.maximum(mx - mn, 2 * eps)
        return scaled * 2.0 - 1.0

    def _last_buffer_items(self, buffer_ids):
        nz = buffer_ids[buffer_ids != 0]                    # nar_model.py:1041-1044
        return nz[: self.p['recent_clicks_for_normalization']]

    @staticmethod
    def _log1p_base(x, base):
        return torch.log(x + 1.0) / torch.log(torch.tensor(base, dtype=x.dtype))   # :28-34

    def _elapsed_days(self, created, ref_ts):
        # nar_model.py:1055-1060: int64 -> float32 BEFORE the subtraction (the graph's own quantisation: kept in f64 mode, where the
        # two float32 values are widened and everything after the casts is float64)
        return torch.relu((ref_ts.to(torch.float32).to(self.dt) - created.to(torch.float32).to(self.dt)) / self._c(MS_PER_DAY))

    # -- nar_model.py:1092-1131 + 1062-1089
    def _recency(self, ids, ref_ts, buffer_ids, stats_ref_ts=None):
        created = self.meta['created_at_ts'][ids].unsqueeze(-1)
        eb = float(self.p.get('elapsed_days_smooth_log_base', 1.3))          # nar_model.py:122, 1071-1075
        x = self._log1p_base(self._elapsed_days(created, ref_ts), eb)
        last = self._last_buffer_items(buffer_ids)
        if last.numel() == 0:
            stats = x[(ids != 0)].reshape(-1)
        else:
            stats = self._log1p_base(self._elapsed_days(self.meta['created_at_ts'][last],
                                                        ref_ts.max() if stats_ref_ts is None else stats_ref_ts), eb)
        return self._normalize_values(x, stats)

    # -- nar_model.py:1134-1193
    def _novelty(self, ids, buffer_ids, pop_norm):
        pb = self._c(float(self.p.get('popularity_smooth_log_base', 2.0)))      # nar_model.py:123, 1148
        nov = -(torch.log(pop_norm[ids].unsqueeze(-1)) / torch.log(pb))
        last = self._last_buffer_items(buffer_ids)
        if last.numel() == 0:
            stats = nov[(ids != 0)]
        else:
            stats = -(torch.log(pop_norm[last].unsqueeze(-1)) / torch.log(pb))
        return self._normalize_values(nov, stats)

    # -- nar_model.py:921-994
    def _item_features(self, ids, ref_ts, buffer_ids, pop_norm, stats_ref_ts=None):
        acfg = self.p['articles_features_config']
        feats = []
        meta_vals = {n: self.meta[n][ids] for n in acfg if n not in ARTICLE_REQ_FEATURES}
        if meta_vals:
            feats.append(self._get_features(meta_vals, acfg, ARTICLE_REQ_FEATURES, 'meta_emb/'))
        if self.ifc['article_content_embeddings']:
            feats.append(self.ace[ids])
        if self.ifc['item_clicked_embeddings']:
            feats.append(self.w['items_embedding'][ids])
        if self.ifc['recency']:
            feats.append(self._recency(ids, ref_ts, buffer_ids, stats_ref_ts))
        if self.ifc['novelty']:
            feats.append(self._novelty(ids, buffer_ids, pop_norm))
        return torch.cat(feats, dim=-1)

    # -- tf.layers.dropout / DropoutWrapper (nar_model.py:338, 352, 368, 418, 1331): y = x / keep_prob * mask (TF 1.12 tf.nn.dropout).
    #    TF's random streams are not reproducible here; the mask is the counter-based one the HIP path uses (csrc/features.hip
    #    k_dropout): element (column c, step t, session row b, site, negative n) is kept iff
    #    Philox4x32-10(ctr = (c, t, b, site + 256 n), key = (seed, step))[0] < floor(keep_prob * 2^32)
    SITE_INPUT, SITE_POSITIVE, SITE_NEGATIVE, SITE_FC1, SITE_RNN = 16, 17, 18, 19, 20

    def _dropout(self, x, site, step):
        keep = float(self.p.get('dropout_keep_prob', 1.0))
        if keep >= 1.0 or not self._train:
            return x
        thr = int(np.float32(keep).astype(np.float64) * 4294967296.0)
        shp = x.shape                                     # [B, T, F] or [B, T, N, F]
        b = np.arange(shp[0], dtype=np.uint64).reshape(-1, 1, 1)
        t = np.arange(shp[1], dtype=np.uint64).reshape(1, -1, 1)
        c = np.arange(shp[-1], dtype=np.uint64).reshape(1, 1, -1)
        if x.dim() == 4:
            n = np.arange(shp[2], dtype=np.uint64).reshape(1, 1, -1, 1)
            r = philox.rand32(c[:, :, None, :], t[:, :, None, :], b[:, :, None, :], np.uint64(site) + np.uint64(256) * n, self.seed, step)
        else:
            r = philox.rand32(c, t, b, site, self.seed, step)
        mask = torch.from_numpy((np.broadcast_to(r, shp) < np.uint64(thr)).astype(np.float32)).to(self.dt)
        return x / self._c(float(np.float32(keep))) * mask

    def _store(self, x):
        """Candidate-row matrices are bf16-resident in the bf16 configuration; identity otherwise."""
        return _StoreBF16.apply(x) if self.gemm_dtype == 'bf16' else x

    def _car(self, x, candidate_rows=False):
        w = self.w
        pre = self._leaky_site('Z1', self._mm(x, w['PreCAR/kernel']) + w['PreCAR/bias'])        # nar_model.py:375-382
        self._tap('Z1', pre)
        out = torch.tanh(self._mm(pre, w['CAR/kernel']) + w['CAR/bias'])        # :384-403
        return self._store(out) if candidate_rows else out

    # -- nar_model.py:1308-1361 (UGRNNCell / GRUCell semantics: SURVEY A.6)
    def _rnn(self, x, lengths):
        B, T, _ = x.shape
        H = self.p['rnn_units']
        L = self.p.get('rnn_num_layers', 1)
        out = x
        for l in range(L):
            h = torch.zeros(B, H, dtype=self.dt)
            ys = []
            for t in range(T):
                xt = out[:, t]
                I = xt.shape[1]
                if self.cell == 'ugrnn':
                    K = self.w['rnn/%d/kernel' % l]
                    # [x, h] W = x W_x + h W_h; the input half is one hoisted GEMM on the HIP path (bf16 mode rounds it), the
                    # recurrent half runs in the fp32 time-step kernel
                    z = (self._mm(xt, K[:I]) + h @ K[I:] if self.gemm_dtype == 'bf16' else self._mmf(torch.cat([xt, h], 1), K)) \
                        + self.w['rnn/%d/bias' % l]
                    g_act, c_act = z[:, :H], z[:, H:]
                    c = torch.tanh(c_act)
                    g = torch.sigmoid(g_act + 1.0)
                    hn = g * h + (1 - g) * c
                else:
                    Kg, Kc = self.w['rnn/%d/gates/kernel' % l], self.w['rnn/%d/candidate/kernel' % l]
                    if self.gemm_dtype == 'bf16':
                        ru = torch.sigmoid(self._mm(xt, Kg[:I]) + h @ Kg[I:] + self.w['rnn/%d/gates/bias' % l])
                    else:
                        ru = torch.sigmoid(self._mmf(torch.cat([xt, h], 1), Kg) + self.w['rnn/%d/gates/bias' % l])
                    r, u = ru[:, :H], ru[:, H:]
                    if self.gemm_dtype == 'bf16':
                        c = torch.tanh(self._mm(xt, Kc[:I]) + (r * h) @ Kc[I:] + self.w['rnn/%d/candidate/bias' % l])
                    else:
                        c = torch.tanh(self._mmf(torch.cat([xt, r * h], 1), Kc) + self.w['rnn/%d/candidate/bias' % l])
                    hn = u * h + (1 - u) * c
                valid = (t < lengths).unsqueeze(1)
                ys.append(torch.where(valid, hn, torch.zeros_like(hn)))   # dynamic_rnn: zero output past length
                h = torch.where(valid, hn, h)                              # state carried unchanged
            out = self._dropout(torch.stack(ys, 1), self.SITE_RNN + l, self._step)        # DropoutWrapper(output_keep_prob), :1331
        return out

    def _scorer(self, m):
        w = self.w
        s1 = self._leaky_site('S1', self._mm(m, w['match1/kernel']) + w['match1/bias'])
        s2 = self._leaky_site('S2', self._mm(s1, w['match2/kernel']) + w['match2/bias'])
        s3 = self._store(self._leaky_site('S3', self._mm(s2, w['match3/kernel']) + w['match3/bias']))
        self._tap('S1', s1); self._tap('S2', s2); self._tap('S3', s3)
        return self._mmf(s3, w['match4/kernel']) + w['match4/bias']       # last layer: fused into the softmax kernel, fp32 in every mode

    def _stage(self, name):
        """Optional wall-clock accounting per stage (bench.py's cpu_baseline leg sets ``self.timers = {}``)."""
        import contextlib
        import time
        timers = getattr(self, 'timers', None)
        if timers is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def cm():
            t0 = time.perf_counter()
            yield
            timers[name] = timers.get(name, 0.0) + time.perf_counter() - t0
        return cm()

    def _leaky_site(self, name, x):
        """leaky_relu at a named site.  Tests may pin the BRANCH per element (``self.leaky_signs[name]`` = list of (sign, valid) boolean
        tensors, consumed in call order): a pre-activation within an fp32 ulp of zero takes either branch in two correct evaluations
        (|value| differs by < 1e-7, the gradient by a factor of 5), so gradient parity is checked with the oracle on the HIP path's
        branches.  Without an override this is tf.nn.leaky_relu."""
        ov = getattr(self, 'leaky_signs', None)
        if ov and ov.get(name):
            sign, valid = ov[name].pop(0)
            pos = torch.where(valid, sign, x.detach() > 0)
            return x * torch.where(pos, self._c(1.0), self._c(0.2))
        return _leaky(x)

    def _tap(self, name, t):
        """Optional capture of leaky-ReLU outputs (tests use them to detect kink sign flips)."""
        taps = getattr(self, 'debug_taps', None)
        if taps is not None:
            taps.setdefault(name, []).append(t.detach())

    def reg_loss(self):
        lam = self.p['reg_weight_decay']
        tot = torch.zeros((), dtype=self.dt)
        for name, (_, _, reg) in self.specs.items():
            if reg:
                tot = tot + lam * (self.w[name] ** 2).sum() / 2.0     # l2_regularizer = scale * l2_loss
        return tot

    # ------------------------------------------------------------------ forward (nar_model.py:210-704)
    def forward(self, features, labels, buffer_ids, pop_norm, mode='train', neg_items=None, step=None, global_max_ts=None):
        p = self.p
        train = (mode == 'train')
        N = p['train_total_negative_samples'] if train else p['eval_total_negative_samples']
        n_buf = p['train_negative_samples_from_buffer'] if train else p['eval_negative_samples_from_buffer']
        step = self.global_step if step is None else step
        item_clicked = torch.as_tensor(features['item_clicked']).long()
        B, T = item_clicked.shape
        seq_len = torch.as_tensor(features['session_size']).long().reshape(-1) - 1          # :227
        mask = torch.arange(T).unsqueeze(0) < seq_len.unsqueeze(1)                       # :231
        event_ts = torch.as_tensor(features['event_timestamp']).long().unsqueeze(-1)     # :233
        max_ts = event_ts.max()                                                           # :235
        if global_max_ts is not None:      # data-parallel row shard: the scalar of the GLOBAL batch
            max_ts = torch.as_tensor(int(global_max_ts))
        label_last = torch.as_tensor(labels['label_last_item']).long()
        label_next = torch.as_tensor(labels['label_next_item']).long()
        all_clicked = torch.cat([item_clicked, label_last], 1)                            # :241
        buffer_np = np.asarray(buffer_ids, dtype=np.int64)
        buffer_t = torch.as_tensor(buffer_np)
        pop_t = torch.as_tensor(np.asarray(pop_norm, dtype=np.float32)).to(self.dt)       # placeholder is tf.float32 (:1442): rounded first
        if neg_items is None:
            with self._stage('sampler'):
                neg_items = osampler.batch_negative_samples(all_clicked.numpy(), buffer_np, N, n_buf, self.seed, step)
        neg = torch.as_tensor(neg_items).long()                                           # [B,T,N]  (:275)

        scfg = p['session_features_config']['sequence_features']
        ctx_vals = {n: torch.as_tensor(features[n]) for n in scfg if n not in SESSION_REQ_SEQ_FEATURES}
        ctx = self._get_features(ctx_vals, scfg, SESSION_REQ_SEQ_FEATURES, 'ctx_emb/')    # :315-317
        if ctx is None:
            ctx = torch.zeros(B, T, 1, dtype=self.dt)                                      # :323-325
        gamma, beta = self.w['gamma'], self.w['beta']
        self._train, self._step = train, step
        with self._stage('gather'):
            x_in = torch.cat([ctx, self._item_features(item_clicked, event_ts, buffer_t, pop_t, max_ts if global_max_ts is not None else None)], 2) * gamma + beta   # :328-333
            x_pos = torch.cat([ctx, self._item_features(label_next, max_ts, buffer_t, pop_t)], 2) * gamma + beta     # :343-347
            ctx_tiled = ctx.unsqueeze(2).expand(B, T, neg.shape[2], ctx.shape[-1])                                     # :360
            x_neg = torch.cat([ctx_tiled, self._item_features(neg, max_ts, buffer_t, pop_t)], 3) * gamma + beta       # :356-364
            x_in, x_pos, x_neg = (self._dropout(x_in, self.SITE_INPUT, step), self._dropout(x_pos, self.SITE_POSITIVE, step),
                                  self._dropout(x_neg, self.SITE_NEGATIVE, step))                                       # :338, 352, 368
        with self._stage('CAR'):
            car_in, car_pos, car_neg = self._car(x_in), self._car(x_pos, True), self._car(x_neg, True)                 # :374-405
        with self._stage('RNN'):
            rnn_out = self._rnn(car_in, seq_len)                                                                      # :408
        with self._stage('scorer'):
            fc1 = self._leaky_site('FC1', self._mm(rnn_out, self.w['FC1/kernel']) + self.w['FC1/bias'])                                        # :411
            self._tap('FC1', fc1)
            fc1 = self._dropout(fc1, self.SITE_FC1, step)                                                                 # :418
            pred = torch.tanh(self._mm(fc1, self.w['FC2/kernel']) + self.w['FC2/bias'])                                       # :423
            s_pos = self._scorer(car_pos * pred)                                                                      # :478-485
            s_neg = self._scorer(car_neg * pred.unsqueeze(2)).squeeze(-1)                                             # :493-500
            logits = torch.cat([s_pos, s_neg], 2)                                                                     # :511
            tau = self._c(float(p['softmax_temperature']))
            probs = torch.softmax(logits / tau, dim=-1)                                                               # :514-515
            loss_mask = mask.to(self.dt)
            xe = -(torch.log(probs[:, :, 0]) * loss_mask).sum() / loss_mask.sum()                                     # :660-664
            reg = self.reg_loss()                                                                                     # :655
            total = xe + reg
        if p.get('novelty_reg_factor', 0.0) > 0.0:                                                                # :673-683
            neg_prob = torch.softmax(s_neg / tau, dim=-1)
            neg_nov = -(torch.log(pop_t[neg]) / torch.log(self._c(float(p.get('popularity_smooth_log_base', 2.0)))))       # :544, 1148
            nov = (p['novelty_reg_factor'] * (neg_prob * neg_nov * loss_mask.unsqueeze(-1)).sum(-1)).sum() / loss_mask.sum()
            total = total - nov
        out = dict(total_loss=total, xe_loss=xe, reg_loss=reg, logits=logits, probs=probs, neg_items=neg,
                   mask=mask, car_in=car_in, car_pos=car_pos, car_neg=car_neg, rnn_out=rnn_out, pred=pred,
                   x_in=x_in, x_pos=x_pos, x_neg=x_neg)
        if not train:
            # nar_model.py:777-794 rank_items_by_predicted_prob (tf.nn.top_k: lowest index wins ties)
            ids = torch.cat([label_next.unsqueeze(-1), neg], 2)
            order = torch.sort(probs, dim=-1, descending=True, stable=True).indices
            out['predicted_item_ids'] = torch.gather(ids, 2, order)
            out['predicted_item_probs'] = torch.gather(probs, 2, order)
        return out

    # ------------------------------------------------------------------ one optimizer step
    def train_step(self, features, labels, buffer_ids, pop_norm, neg_items=None, return_grads=False):
        for v in self.w.values():
            v.grad = None
        out = self.forward(features, labels, buffer_ids, pop_norm, 'train', neg_items)
        with self._stage('backward'):
            out['total_loss'].backward()
        with self._stage('Adam'):
            grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v)).clone())
                                for k, v in self.w.items())
            self.adam_update(grads)
        out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        if return_grads:
            out['grads'] = grads
        return out

    def adam_update(self, grads):
        """tf.train.AdamOptimizer(lr, 0.9, 0.999, 1e-8) (nar_model.py:708-722): bias correction folded
        into the step size, epsilon added to sqrt(v) un-corrected; dense over every row of every table."""
        lr, b1, b2, eps = self.p['lr'], 0.9, 0.999, 1e-8
        self.global_step += 1
        t = self.global_step
        lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        with torch.no_grad():
            for k, w in self.w.items():
                g = grads[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                w.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))

    def weights_numpy(self):
        return OrderedDict((k, v.detach().numpy().copy()) for k, v in self.w.items())
