"""Minimal Estimator runtime behind the reference's ``model_fn`` / ``input_fn`` contract.

The reference drives training through tf.estimator.Estimator (nar_trainer_gcom.py:335-386, 511-525): every
``train(input_fn)`` / ``evaluate(input_fn)`` call builds the graph by calling ``input_fn()`` then
``model_fn(features, labels, mode, params)``, restores the variables from ``model_dir``, runs one ``session.run`` per
batch with the spec's SessionRunHooks around it until the input is exhausted (tf.errors.OutOfRangeError), and saves a
checkpoint.  This module keeps exactly that control flow without a graph: ``features`` / ``labels`` are the re-bound
batch handles of datasets.prepare_dataset_iterator, the "variables" are the device-resident NARRuntime kept in the
Estimator's variable store between calls (and checkpointed to model_dir), and a step is ``spec.train_op()``.
"""
import os
import time

import numpy as np
import torch

from . import nar_model
from .nar_model import ModeKeys


class RunConfig:
    """tf.estimator.RunConfig fields the reference sets (nar_trainer_gcom.py:343-348)."""

    def __init__(self, tf_random_seed=None, keep_checkpoint_max=1, save_checkpoints_secs=1200, save_summary_steps=100,
                 log_step_count_steps=100, model_dir=None):
        self.tf_random_seed = tf_random_seed
        self.keep_checkpoint_max = keep_checkpoint_max
        self.save_checkpoints_secs = save_checkpoints_secs
        self.save_summary_steps = save_summary_steps
        self.log_step_count_steps = log_step_count_steps
        self.model_dir = model_dir


class EstimatorSpec:
    def __init__(self, mode, loss=None, train_op=None, eval_metric_ops=None, training_chief_hooks=None, training_hooks=None,
                 evaluation_hooks=None, predictions=None):
        self.mode, self.loss, self.train_op = mode, loss, train_op
        self.eval_metric_ops = eval_metric_ops or {}
        self.training_chief_hooks = list(training_chief_hooks or [])
        self.training_hooks = list(training_hooks or [])
        self.evaluation_hooks = list(evaluation_hooks or [])
        self.predictions = predictions
        if mode == ModeKeys.TRAIN and (loss is None or train_op is None):
            raise ValueError("Missing loss / train_op for ModeKeys.TRAIN")
        if mode == ModeKeys.EVAL and loss is None:
            raise ValueError("Missing loss for ModeKeys.EVAL")


class SessionRunArgs:
    def __init__(self, fetches=None, feed_dict=None):
        self.fetches, self.feed_dict = fetches or {}, feed_dict or {}


class SessionRunValues:
    def __init__(self, results):
        self.results = results


class SessionRunContext:
    def __init__(self, model):
        self.model = model
        self.stop_requested = False

    def request_stop(self):
        self.stop_requested = True


class SessionRunHook:
    def begin(self):
        pass

    def after_create_session(self, session=None, coord=None):
        pass

    def before_run(self, run_context):
        return None

    def after_run(self, run_context, run_values):
        pass

    def end(self, session=None):
        pass


class StreamingMean:
    """(value, update_op) pair of tf.metrics.* as one object: ``update(batch_sum, batch_count)`` / ``result()``."""

    def __init__(self):
        self.total, self.count = 0.0, 0.0

    def update(self, s, n):
        self.total += float(s); self.count += float(n)

    def result(self):
        return self.total / self.count if self.count > 0 else 0.0


class Estimator:
    def __init__(self, model_fn, model_dir=None, config=None, params=None, warm_start_from=None):
        self._model_fn = model_fn
        self.config = config or RunConfig()
        self.model_dir = model_dir or self.config.model_dir
        self.params = dict(params or {})
        seed = self.config.tf_random_seed
        self._store = dict(runtime=None, tf_random_seed=42 if seed is None else int(seed))
        self._last_ckpt_time = time.time()
        self.warm_start_from = warm_start_from
        self.steps_per_sec = None
        if self.model_dir:
            os.makedirs(self.model_dir, exist_ok=True)

    # ---- checkpoints (nar_trainer_gcom.py:343-353: keep_checkpoint_max=1, implicit restore)
    def _ckpt_path(self):
        return os.path.join(self.model_dir, "model.ckpt.pt") if self.model_dir else None

    def latest_checkpoint(self):
        p = self._ckpt_path()
        return p if p and os.path.exists(p) else None

    def _maybe_restore(self, created_now):
        """A freshly created runtime (first call in this process) is restored from model_dir when a checkpoint exists - or, failing
        that, from ``warm_start_from`` (a checkpoint file or a model_dir; the reference copies a previous job's checkpoint into
        model_dir for that, nar_trainer_gcom.py:450-459).  Checkpoints hold tensors, ints and strings only: ``weights_only=True``."""
        if not created_now:
            return
        path = self.latest_checkpoint()
        if path is None and self.warm_start_from:
            w = self.warm_start_from
            path = os.path.join(w, "model.ckpt.pt") if os.path.isdir(w) else w
            if not os.path.exists(path):
                raise FileNotFoundError("warm_start_from: no checkpoint at %s" % path)
        if path is not None:
            sd = torch.load(path, map_location="cpu", weights_only=True)
            self._store['runtime'].load_state_dict(sd)

    def save_checkpoint(self):
        """Checkpoint of the runtime in model_dir.  Under data parallelism state_dict() is a COLLECTIVE in the sharded mode (it
        all-gathers the Adam slots their owner ranks hold): every rank calls it at the same step (see _checkpoint_due), rank 0 alone
        writes the file."""
        p = self._ckpt_path()
        rt = self._store.get('runtime')
        self._check_device_health()           # never persist weights a failed kernel hand-off may have touched
        if p and rt is not None:
            sd = rt.state_dict()
            if getattr(rt, 'dp_rank', 0) == 0:
                tmp = p + ".tmp"
                torch.save(sd, tmp)
                os.replace(tmp, p)
            self._last_ckpt_time = time.time()

    def _coop_timed_out_local(self):
        rt = self._store.get('runtime')
        return bool(rt is not None and getattr(rt, '_rnn_coop_ws', None) and rt.rnn_coop_timed_out())

    _HEALTH_MSG = ("a cooperative recurrent kernel (csrc/rnn_coop.hip) gave up a bounded spin%s: the hidden states of at least one "
                   "step are invalid.  Is another process running cooperative kernels on this GPU?")

    def _check_device_health(self, collective=True):
        """The cooperative recurrent kernels (csrc/rnn_coop.hip) hand h_t from workgroup to workgroup with BOUNDED spins; a spin that gave
        up (a cooperating workgroup never became resident - e.g. several processes' cooperative kernels competing for one GPU) leaves a
        sticky word in their workspace and wrong hidden states behind.  Checked where the loop synchronises anyway (checkpoints, the end of
        train / evaluate, and the data-parallel checkpoint poll): fail loudly rather than train on.  The word is PER PROCESS: under data
        parallelism the flags are MAX-reduced over the data-parallel group first, so that every rank raises together - one rank raising
        alone would leave the others waiting in the next collective (state_dict() all-gathers the Adam slots in the sharded mode).
        Every rank reaches the collective form at the same steps (save_checkpoint is entered on rank 0's broadcast decision; the end of
        train()).  evaluate() passes collective=False: nothing makes every rank evaluate (chief-only or uneven evaluation is legal - the
        evaluation forward issues no collective), so there the check is on the local word only."""
        bad = self._coop_timed_out_local()
        rt = self._store.get('runtime')
        where = ""
        if collective and rt is not None and getattr(rt, 'dp_active', False):
            import torch.distributed as dist
            if dist.is_initialized():
                pg = getattr(rt, 'dp_pg', None)
                flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device=rt.device if dist.get_backend(pg) == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=pg)
                if int(flag.item()) and not bad:
                    where = " on another data-parallel rank"
                bad = bool(int(flag.item()))
        if bad:
            raise RuntimeError(self._HEALTH_MSG % where)

    def _checkpoint_due(self):
        """save_checkpoints_secs elapsed?  A per-rank wall-clock decision would let ranks enter save_checkpoint() - a collective under the
        sharded exchange mode - at different steps (hang, or a gather paired with the next step's reduce-scatter): with an active
        data-parallel group its first rank decides - ONE MAX all-reduce of [rank 0's decision (the others contribute 0), this rank's
        kernel time-out word] on the data-parallel process group, only every CKPT_DECISION_EVERY steps (every rank counts the same
        steps): the read-back is a host-device synchronisation, which every step would undo the enqueue-ahead of the training loop
        (ADVICE r03).  The second word makes a failed cooperative-kernel hand-off on ANY rank stop EVERY rank within
        CKPT_DECISION_EVERY steps instead of at the next checkpoint."""
        secs = self.config.save_checkpoints_secs
        due = bool(secs) and time.time() - self._last_ckpt_time > secs
        rt = self._store.get('runtime')
        if rt is not None and getattr(rt, 'dp_active', False):
            import torch.distributed as dist
            if dist.is_initialized():
                self._ckpt_poll = getattr(self, '_ckpt_poll', 0) + 1
                if self._ckpt_poll % self.CKPT_DECISION_EVERY:
                    return False
                pg = getattr(rt, 'dp_pg', None)
                first = dist.get_rank(pg) == 0
                bad = self._coop_timed_out_local()
                flag = torch.tensor([1 if (due and first) else 0, 1 if bad else 0], dtype=torch.int32,
                                    device=rt.device if dist.get_backend(pg) == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=pg)
                due, any_bad = (bool(int(x)) for x in flag.tolist())
                if any_bad:
                    raise RuntimeError(self._HEALTH_MSG % ("" if bad else " on another data-parallel rank"))
        return due

    CKPT_DECISION_EVERY = 50

    def get_variable_value(self, name):
        """tf.estimator.Estimator.get_variable_value: by the reference graph's TF variable name (layout.tf_variable_names(), ':0' and the
        alias spellings accepted) or by the short logical name; the reference's shapes (un-padded, UGRNN kernel as [I + H, 2 H])."""
        L = self._store['runtime'].layout
        w = self._store['runtime'].logical_weights()
        if name in w:
            return w[name]
        k = name[:-2] if name.endswith(':0') else name
        k = L.tf_variable_aliases().get(k, k)
        return w[L.tf_variable_names()[k]]

    def get_variable_names(self):
        """tf.estimator.Estimator.get_variable_names: the trainable variables of the reference graph under their TF names."""
        return list(self._store['runtime'].layout.tf_variable_names().keys())

    @property
    def global_step(self):
        rt = self._store.get('runtime')
        return rt.global_step if rt is not None else 0

    # ---- the session.run loop
    def _call_model_fn(self, input_fn, mode):
        features, labels = input_fn()
        had_rt = self._store.get('runtime') is not None
        with nar_model.variable_store(self._store):
            spec = self._model_fn(features, labels, mode, self.params)
        self._maybe_restore(not had_rt and self._store.get('runtime') is not None)
        return features, labels, spec

    @staticmethod
    def _run_hooks_step(hooks, ctx, step_fn):
        fetches = []
        feed = {}
        for h in hooks:
            args = h.before_run(ctx)
            fetches.append(args.fetches if args is not None else None)
            if args is not None:
                feed.update(args.feed_dict)
        if feed:
            ctx.model.feed(feed)
        out = step_fn()
        for h, f in zip(hooks, fetches):
            res = {k: (v.eval() if hasattr(v, 'eval') else v) for k, v in f.items()} if f is not None else None
            h.after_run(ctx, SessionRunValues(res))
        return out

    def train(self, input_fn, hooks=None, steps=None, max_steps=None):
        features, labels, spec = self._call_model_fn(input_fn, ModeKeys.TRAIN)
        ds = features.dataset
        all_hooks = list(spec.training_chief_hooks) + list(spec.training_hooks) + list(hooks or [])
        model = getattr(spec.train_op, '__self__', None)
        ctx = SessionRunContext(model)
        for h in all_hooks:
            h.begin()
        n, t0 = 0, time.time()
        last_log, log_every = 0, self.config.log_step_count_steps
        while (steps is None or n < steps) and (max_steps is None or self.global_step < max_steps) and not ctx.stop_requested:
            if not ds.advance():
                break
            self._run_hooks_step(all_hooks, ctx, spec.train_op)
            n += 1
            if hasattr(model, 'stage_next'):
                model.stage_next(ds)          # next batch: H2D copies + negative sampling behind this step (nar_model.presample)
            if log_every and n - last_log >= log_every:
                torch.cuda.synchronize()
                dt = time.time() - t0
                self.steps_per_sec = n / dt
                print("INFO:global_step/sec: %.4g (step %d)" % (n / dt, self.global_step), flush=True)
                last_log = n
            # (called every step whatever save_checkpoints_secs says: under data parallelism it is also the cross-rank poll of the
            # kernel time-out word, every CKPT_DECISION_EVERY steps - ADVICE r05; without a data-parallel group it reads the clock)
            if self._checkpoint_due():
                self.save_checkpoint()
        torch.cuda.synchronize()
        if n:
            self.steps_per_sec = n / max(1e-9, time.time() - t0)
        for h in all_hooks:
            h.end(None)
        ds.close()
        self.save_checkpoint()
        return self

    def evaluate(self, input_fn, steps=None, hooks=None, name=None):
        features, labels, spec = self._call_model_fn(input_fn, ModeKeys.EVAL)
        ds = features.dataset
        all_hooks = list(spec.evaluation_hooks) + list(hooks or [])
        model = None
        for h in all_hooks:
            model = getattr(h, 'model', model)
        ctx = SessionRunContext(model)
        for h in all_hooks:
            h.begin()
        n, loss_sum = 0, 0.0
        while steps is None or n < steps:
            if not ds.advance():
                break
            self._run_hooks_step(all_hooks, ctx, model.run_step)
            loss_sum += float(np.asarray(spec.loss.eval() if hasattr(spec.loss, 'eval') else spec.loss).reshape(-1)[0])
            n += 1
        for h in all_hooks:
            h.end(None)
        ds.close()
        self._check_device_health(collective=False)
        out = {k: (v.result() if hasattr(v, 'result') else v) for k, v in spec.eval_metric_ops.items()}
        out['loss'] = loss_sum / max(1, n)
        out['global_step'] = self.global_step
        return out
