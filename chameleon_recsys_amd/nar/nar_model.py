"""NAR module model on MI355X: host orchestration of the HIP kernels in libchameleon_nar.so.

Mirrors nar_module/nar/nar_model.py of the reference:
  * ``NARModuleModel(mode, inputs, labels, ...)`` keeps the reference constructor (nar_model.py:102-129) and the
    attributes its SessionRunHook reads (:1435-1467);
  * the TF graph (:210-722) becomes an eager sequence of stream-ordered kernel launches over device-resident
    buffers: negative sampling -> de-duplicated feature assembly -> factorised PreCAR -> CAR -> UGRNN -> FCs ->
    scorer -> sampled softmax loss -> full backward -> L2 + TF-Adam.
PyTorch is only used for device memory, streams and (data-parallel) torch.distributed; all arithmetic of the step
runs in the hand-written HIP kernels.  There is no CPU fallback.
"""
import math
import os

import numpy as np
import torch

from .. import _lib
from .. import _roctx
from .._lib import check, ptr
from .clicked_items_state import lane_stream
from .layout import COL_ITEMEMB, ParamLayout

ACT_NONE, ACT_LEAKY, ACT_TANH = 0, 1, 2


class ModeKeys:                       # tf.estimator.ModeKeys
    TRAIN = 'train'
    EVAL = 'eval'
    PREDICT = 'infer'


# Raw handle (hipStream_t as an int) of torch's current stream.  torch.cuda.current_stream() builds a Python Stream object (~8 us);
# a step makes ~100 launches, and with short sessions the host's enqueue time is what bounds the boundary throughput - the private
# C getters cost ~0.3 us.  (Fallback if a torch build lacks them.)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


_raw_checked = False


def _stream():
    global _raw_stream, _raw_checked
    if _raw_stream is not None and _raw_device is not None:
        if not _raw_checked:       # private torch API: validate it once against the public one (a changed signature / meaning -> fall back)
            _raw_checked = True
            try:
                if int(_raw_stream(_raw_device())) != int(torch.cuda.current_stream().cuda_stream):
                    _raw_stream = None
            except Exception:
                _raw_stream = None
            if _raw_stream is None:
                return torch.cuda.current_stream().cuda_stream
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class Tensor:
    """Lazy fetch handle: what a symbolic tf.Tensor attribute of the reference model is to its SessionRunHook
    (nar_model.py:1435-1456).  ``eval()`` returns the numpy value of the LAST executed step."""

    def __init__(self, name, fn):
        self.name, self._fn = name, fn

    def eval(self):
        v = self._fn()
        if torch.is_tensor(v):
            v = v.cpu().numpy()
        return v


class Placeholder:
    """tf.placeholder stand-in (nar_model.py:170-202): fed through ``NARModuleModel.feed(feed_dict)``."""

    def __init__(self, name):
        self.name = name

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, o):
        return isinstance(o, Placeholder) and o.name == self.name


# The Estimator's "variable store": TF re-creates the graph at every train()/evaluate() call and restores the
# variables from the checkpoint in model_dir; here the device-resident NARRuntime simply survives between calls.
_VARIABLE_STORE = None


class variable_store:
    def __init__(self, store):
        self.store = store

    def __enter__(self):
        global _VARIABLE_STORE
        self._prev, _VARIABLE_STORE = _VARIABLE_STORE, self.store
        return self.store

    def __exit__(self, *a):
        global _VARIABLE_STORE
        _VARIABLE_STORE = self._prev


_TORCH_DTYPE = {np.dtype(n).str: t for n, t in (('int64', torch.int64), ('int32', torch.int32), ('float32', torch.float32),
                                                   ('uint8', torch.uint8))}


def pack_offsets(arrays, align=256):
    """Byte offsets of a dict of numpy arrays packed back to back into one staging block (each start aligned: the device views of
    upload_batch reinterpret the block per dtype) and the block's size."""
    offs, total = {}, 0
    for k, a in arrays.items():
        offs[k] = total
        total += (a.nbytes + align - 1) & ~(align - 1)
    return offs, total


class _PinnedRing:
    """Page-locked staging buffers for the per-step host -> device copies: a batch's arrays are packed into one
    pinned arena and copied with truly asynchronous DMA (a copy from pageable memory goes through the driver's bounce buffer and
    blocks the host).  A ring of arenas; an arena is re-used only after the copies issued from it have completed (event)."""

    def __init__(self, n=4, nbytes=4 << 20):
        self.n, self.nbytes = n, nbytes
        self.bufs, self.events, self.k, self.off = [None] * n, [None] * n, -1, 0

    def begin(self, need):
        self.k = (self.k + 1) % self.n
        if self.events[self.k] is not None:
            self.events[self.k].synchronize()
        if self.bufs[self.k] is None or self.bufs[self.k].numel() < need:
            self.bufs[self.k] = torch.empty(max(need, self.nbytes), dtype=torch.uint8, pin_memory=True)
        self.off = 0

    def stage(self, a):
        n = a.nbytes
        if n == 0 or self.off + n > self.bufs[self.k].numel():      # (arena sized too small: this array goes through pageable memory)
            return torch.from_numpy(a)
        view = self.bufs[self.k][self.off:self.off + n]
        self.off += (n + 255) & ~255
        t = view.view(torch.from_numpy(a).dtype).view(a.shape)
        t.copy_(torch.from_numpy(a))
        return t

    def end(self, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.events[self.k] = ev


class NARRuntime:
    """Everything that outlives a single train()/evaluate() call: weights + Adam slots in ONE flat HBM buffer,
    resident article tables (ACE matrix, metadata - re-fed from numpy every step by the reference,
    nar_model.py:1458-1467), global step, workspaces."""

    def __init__(self, params, device='cuda:0', seed=42, weights=None):
        self.lib = _lib.load()
        self._views = {}
        if not torch.cuda.is_available():
            raise _lib.ChameleonLibError("no ROCm device visible: the NAR step has no CPU path")
        self.device = torch.device(device)
        self.params = params
        acfg = params['articles_features_config']
        ace = np.ascontiguousarray(params['content_article_embeddings_matrix'], dtype=np.float32)
        self.n_items = acfg['article_id'].get('cardinality', ace.shape[0]) if 'article_id' in acfg else ace.shape[0]
        if self.n_items != ace.shape[0]:
            raise ValueError("article_id cardinality (%d) != ACE rows (%d)" % (self.n_items, ace.shape[0]))
        self.item_id_bits = max(1, int(self.n_items - 1).bit_length())      # radix passes of the row grouping (cham_group_rows)
        self.layout = ParamLayout(params['session_features_config'], acfg, self.n_items, ace.shape[1],
                                  params['CAR_embedding_size'], params['rnn_units'],
                                  params.get('rnn_num_layers', 1), params.get('rnn_cell', 'ugrnn'),
                                  params.get('internal_features_config'), params.get('max_cardinality_for_ohe', 10))
        L = self.layout
        logical = weights if weights is not None else L.init_logical(seed)
        dev = self.device
        self.flat = torch.from_numpy(L.pack(logical)).to(dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.grads = torch.zeros_like(self.flat)
        self.global_step = 0
        # The launch parameters that change from step to step (sampler key, the batch's max time stamp, sum(mask), Adam's lr_t) live in a
        # 32-byte DEVICE record (csrc/common.h ChamStepScalars, include/chameleon_nar.h "STEP SCALARS"), written by set_step_scalars() in
        # front of the step: the step's launches then carry no per-step host value and can be captured in a hipGraph (GraphedTrainStep).
        # CHAM_DEV_SCALARS=0: the by-value entry points (A/B arm; bit-identical results).  capturing: a graph capture is in progress - the
        # record is NOT written by the step's code (the replay writes it in front of every launch of the graph).
        self.dev_scalars = os.environ.get("CHAM_DEV_SCALARS", "1") == "1"
        # round 6: the loss record's kernels (L2 term, finalize) and the zero fill of the embedding gradients off the main lane (CHAM_LOSS_SIDE=0: on it)
        self.loss_side = os.environ.get("CHAM_LOSS_SIDE", "1") == "1"
        # round 6: per-click sums of dZ1 in the CAR dgrad's epilogue instead of a second pass over its 1 GB (CHAM_DGRAD_GROUPSUM=0: k_combine_bwd_u)
        self.dgrad_groupsum = os.environ.get("CHAM_DGRAD_GROUPSUM", "1") == "1"
        # round 6, the serial tail of the backward pass (profiles/r06_notes.md section 5b): dgamma / dbeta column sums with coalesced reads
        # through a workspace (CHAM_FEATURE_BWD_WS), the user-context half of the PreCAR input's backward on the third lane (CHAM_TAIL_SPLIT)
        self.feature_bwd_ws = os.environ.get("CHAM_FEATURE_BWD_WS", "1") == "1"
        self.tail_split = os.environ.get("CHAM_TAIL_SPLIT", "1") == "1"
        self.capturing = False
        self.scalars = None
        # 'f32': exact fp32 MFMA (BASELINE config 2, the default); 'bf16': operands of every Dense / matmul rounded to bf16 on
        # the fly, fp32 accumulation, fp32 storage / softmax / loss / Adam (BASELINE config 3)
        self.gemm_dtype = params.get('gemm_dtype', 'f32')
        # 'f32': fp32 operands; the wide GEMMs (N > 64) run as six bf16-plane products per fp32 product on the bf16 matrix cores
        # (csrc/gemm_x3.hip, fp32-grade error, 6/16 of the native fp32 matrix time), the narrow ones on v_mfma_f32_32x32x2_f32.
        # 'f32_native': every GEMM on the native fp32 MFMA (csrc/gemm.hip) - the reference arm of the parity tests.
        # 'bf16': BASELINE config 3 (bf16-resident candidate-row matrices, fp32 accumulate).
        if self.gemm_dtype not in ('f32', 'f32_native', 'bf16'):
            raise ValueError("gemm_dtype must be 'f32', 'f32_native' or 'bf16'")
        self.x3 = self.gemm_dtype == 'f32' and os.environ.get("CHAM_GEMM_X3", "1") == "1"
        # plane-resident CAR GEMMs (csrc/gemm_p3.hip): the candidate rows of Z1 and dZ2 live in HBM as three bf16 planes written by
        # their producers, W2 / W2^T get plane shadows once per step; same six plane products as gemm_x3.hip.  CHAM_GEMM_P3=0: the
        # on-the-fly split for those three GEMMs too (A/B arm).  The only shape gate is C % 256 == 0 below (decided once, here): any other
        # argument the plane kernels reject is an error of this module and raises from check() mid-step - loudly, there is no silent
        # fall-back to another arithmetic
        self.p3 = self.x3 and os.environ.get("CHAM_GEMM_P3", "1") == "1" and self.layout.C % 256 == 0
        # two-fp16-plane form of those three GEMMs (csrc/gemm_h2.hip, round 4): planes (h, l) x a power-of-two scale derived on the device
        # from a bound of the matrix, THREE plane products instead of six - the kernels were power-limited, so halving the MFMA count is
        # what moves them; same float64-error bar as the six-product form (tests/test_gemm_h2_gpu.py).  CHAM_GEMM_H2=0: the three-bf16-plane arm
        self.h2 = self.p3 and os.environ.get("CHAM_GEMM_H2", "1") == "1"
        # ... optionally with the candidate-row planes TILE-BLOCKED (round 6; [row tiles of 256][C / 32][256][32], csrc/common.h h2b_index): an
        # LDS-DMA request of the NT forms (16 rows x 32 k) is then 1 KB of whole 128-byte lines instead of sixteen half lines.  Bit-identical
        # results (same pieces, same products).  MEASURED (profiles/r06_notes.md section 2): the NT forms gain 4-5 % (stand-alone 1.41 -> 1.36 and
        # 1.44 -> 1.375 ms), the TN weight gradient loses 2.5 % stand-alone and 6-19 % in the step, the plane-writing combine 0.09 ms: the step is
        # 0.05 ms SLOWER (7.89 vs 7.84 ms) - the NT forms were power-limited, not line-fill-limited.  Hence OFF by default; CHAM_H2_BLOCKED=1 is
        # the arm (needs the 64-byte-piece NT kernel, CHAM_H2_NT_WIDE=1).
        blk = os.environ.get("CHAM_H2_BLOCKED", "0")          # 1 = both matrices, dz2 / z1 = only that one (A/B arms)
        self.h2_blocked = (self.h2 and blk in ("1", "dz2", "z1") and os.environ.get("CHAM_H2_NT_WIDE", "1") == "1" and self.layout.C % 32 == 0)
        self.h2_blocked_which = blk if self.h2_blocked else "0"
        # NT forms of those GEMMs (CAR forward, CAR dgrad) on the 64-byte-source-piece kernel (csrc/gemm_h2.hip gemm_h2w_kernel, round 5);
        # CHAM_H2_NT_WIDE=0: the 32-byte-piece kernel of round 4 (bit-identical results, A/B arm).  A LIBRARY-wide setting, made once per
        # process when the library is loaded (chameleon_recsys_amd/_lib.py): constructing a second runtime never flips the kernels of the first
        # the scorer's first layer on two fp16 planes split while staged (round 5) - THREE plane products instead of six for the 3 x 65 GFLOP that were
        # left on six-product kernels: its weight gradient (csrc/gemm_x3.hip NP = 2, cham_gemm_f32x2h; |cand (.) pred| <= 1: a constant scale record,
        # dS1 by its max row norm) and the products inside its fused dgrad (csrc/dm_fused.hip MODE 3, cham_dm_mulpred_h2h; Ws1's planes by its max
        # row norm).  CHAM_S1_H2=1 (default): those two BACKWARD kernels; =a: the forward GEMM as well (another 0.1 ms, but the 200-step loss
        # curve then left the float64 curve earlier than every other fp32 arm - one trajectory; the CPU oracle with the same rounding: 1.1-1.4 x -
        # profiles/r05_notes.md section 9 - so the forward keeps its exact 24-bit operands); =0: six bf16 products in all three
        mode = os.environ.get("CHAM_S1_H2", "1") if self.h2 else "0"
        self.s1_h2 = mode in ("1", "a")          # the weight gradient
        self.dm_f16 = self.s1_h2                  # the fused dgrad's own products
        self.s1_h2_fwd = mode == "a"              # the forward GEMM
        self.tf_random_seed = int(params.get('tf_random_seed', 42))
        # resident article tables
        meta = params['articles_metadata']
        self.ace = torch.from_numpy(ace).to(dev)
        self.created = torch.from_numpy(np.ascontiguousarray(meta['created_at_ts'], dtype=np.int64)).to(dev)
        def meta_column(n):
            # metadata columns live in ONE int64 table.  A float-valued numerical article feature (config dtype 'float',
            # nar_model.py:755-757) is stored as its float32 bit pattern and read back bit-exactly by the assemble kernels
            # (layout.meta_is_float -> descriptor sub-field 1); everything else as the integer it is.
            a = np.asarray(meta[n])
            if n in L.meta_is_float:
                return np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
            if acfg[n]['type'] == 'numerical' and not np.issubdtype(a.dtype, np.integer) and not np.array_equal(a, np.round(a)):
                raise ValueError("article feature %r holds non-integer values but its config says dtype %r: declare it "
                                 "{'type': 'numerical', 'dtype': 'float'}" % (n, acfg[n].get('dtype')))
            return a.astype(np.int64)
        if L.meta_names:
            mc = np.stack([meta_column(n) for n in L.meta_names])
        else:
            mc = np.zeros((1, self.n_items), np.int64)
        self.meta_cat = torch.from_numpy(np.ascontiguousarray(mc)).to(dev)
        self.ctx_desc = torch.from_numpy(L.ctx_descriptors()).to(dev)
        self.ctx_emb_groups, self.item_emb_groups = L.ctx_emb_groups(), L.item_emb_groups()
        segs, singles = L.item_segments()
        self.item_segs, self.n_item_segs = torch.from_numpy(np.ascontiguousarray(segs)).to(dev), int(segs.shape[0])
        self.item_singles, self.n_item_singles = torch.from_numpy(np.ascontiguousarray(singles)).to(dev), int(singles.shape[0])
        # item rows through LDS tiles (csrc/features.hip k_item_assemble_lds); rows too wide for the tile: one thread per element
        self.item_lds = L.Fi * 4 * 8 <= 64 * 1024
        self.item_desc = torch.from_numpy(L.item_descriptors()).to(dev)
        self.gemm_ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)        # 256 MB split-K partials (main stream)
        self.colsum_ws = torch.empty(4 << 20, dtype=torch.float32, device=dev)
        # the recurrent branch (8 CUs busy) runs on a side stream, overlapped with the candidate-row CAR GEMMs (normal priority: a
        # high-priority side lane was needed by the early builds only, profiles/r01_notes.md items 2 and 25)
        self.side_stream = lane_stream(dev, "side", 0)
        self._side_raw = self.side_stream.cuda_stream
        self.gemm_ws_side = torch.empty(32 << 20, dtype=torch.float32, device=dev)       # split-K partials of the side lane
        self.colsum_ws_side = torch.empty(2 << 20, dtype=torch.float32, device=dev)
        # third lane (round 5, CHAM_WGRAD_AUX): the small weight / bias gradients of the backward pass leave the side lane, whose W2 weight
        # gradient then starts the moment the main lane's CAR dgrad has finished instead of behind ~1 ms of small launches
        self.wgrad_aux = os.environ.get("CHAM_WGRAD_AUX", "1") == "1"
        self.aux_stream = lane_stream(dev, "aux", 0)
        self._aux_raw = self.aux_stream.cuda_stream
        self.gemm_ws_aux = torch.empty(16 << 20, dtype=torch.float32, device=dev)
        self.colsum_ws_aux = torch.empty(2 << 20, dtype=torch.float32, device=dev)
        # head of the step (round 5, CHAM_HEAD_SPLIT): the clicked rows' PreCAR combine + CAR layer 2 feed the recurrent branch only - they
        # run at the head of the side lane instead of in front of the main lane's plane producer and CAR forward GEMM
        self.head_split = os.environ.get("CHAM_HEAD_SPLIT", "1") == "1"
        self.overlap = os.environ.get("CHAM_OVERLAP", "1") == "1"           # CHAM_OVERLAP=0: the same program order on one stream
        # row-wise stages on the non-padded (session, time) positions only (upload_batch); CHAM_COMPACT=0 computes the padded
        # positions too and masks them, like the reference graph does
        self.compact = os.environ.get("CHAM_COMPACT", "1") == "1"
        # Measured schedule constants (the A/B arms they won against are in profiles/r01-r04_notes.md, not in the code):
        #   * K-splits of the plane-resident W2 weight gradient: 32 = two rounds of shorter workgroups, the main lane's PreCAR backward
        #     slips in between (16 = one workgroup per CU for the whole kernel, nothing else gets a CU meanwhile);
        #   * short (ragged) batches, at most this many candidate rows: the W2 weight gradient runs on the MAIN lane while that lane waits
        #     for the clicked-row gradient out of the side lane's recurrent chain (the chain does not shrink with the valid positions);
        #   * otherwise it waits for the main lane's CAR dgrad (two one-workgroup-per-CU matrix kernels only time-slice the chip) and is
        #     the side lane's LAST work: the small weight / bias gradients go first and run under that dgrad.
        self.p3_w2_splits = 32
        self.w2_main_rows = 131072
        #   * the same threshold selects the recurrent kernels: steps with at most this many candidate rows (ragged batches, the shard of
        #     a strong-scaling rank) run the UGRNN time steps on eight cooperating workgroups per 32 sessions with W_h resident in LDS
        #     (csrc/rnn_coop.hip: ~8 us per time step instead of ~33, but 120-155 KB of LDS per workgroup - no CU shared with a plane-GEMM
        #     workgroup); a full batch hides the single-workgroup kernels (csrc/rnn.hip, 33 KB of LDS) behind its big GEMMs
        self.rnn_coop_rows = 131072 if (L.cell == 'ugrnn' and L.Hp == 256 and not L.rnn_stepwise) else -1
        self._rnn_coop_ws = {}
        self.presample = os.environ.get("CHAM_PRESAMPLE", "1") == "1"       # NARModuleModel.presample (A/B switch)
        self.upload_stream = lane_stream(dev, "upload")       # H2D copies of a batch on their own stream, from a page-locked staging ring
        self.pinned = _PinnedRing()
        self.sumsq = torch.zeros(1024, dtype=torch.float32, device=dev)
        # bf16 configuration: bf16 shadows (plain + transposed) of the weights whose GEMMs run over the candidate rows, refreshed
        # whenever the fp32 master copy changed (csrc/gemm_b16.hip wants both operands k-contiguous: W^T forward, W dgrad)
        self.b16 = self.gemm_dtype == 'bf16'
        self.shadow, self._shadow_key, self.weights_version = {}, None, 0
        if self.b16:
            if L.C % 128:
                raise ValueError("gemm_dtype='bf16' needs CAR_embedding_size % 128 == 0")
            for name in ('W2', 'Ws1', 'Ws2', 'Ws3'):
                r, c = L.entries[name].shape
                self.shadow[name] = torch.zeros(r, c, dtype=torch.bfloat16, device=dev)
                self.shadow[name + 'T'] = torch.zeros(c, r, dtype=torch.bfloat16, device=dev)
        # bf16 configuration: the three candidate-row CAR GEMMs on the LDS-DMA core (csrc/gemm_p3.hip, gemm_b1_kernel) where it takes the
        # shape; the register-staged kernels of csrc/gemm_b16.hip otherwise (tests switch rt.b16_dma off to cover those at the G1 shape)
        self.b16_dma = self.gemm_dtype == 'bf16' and L.C % 256 == 0
        # ... and the scorer layer-1 dgrad fused with the cand (.) pred / CAR-tanh backward (csrc/dm_fused.hip, MODE 2) where the kernel takes
        # the shape (1 + N in [32, 256] is checked per step)
        self.dm_fused_b16 = self.gemm_dtype == 'bf16' and L.entries['Ws1'].shape == (L.C, 128) and L.C % 64 == 0
        if self.p3:
            C_ = L.C
            npl, pdt = (2, torch.float16) if self.h2 else (3, torch.bfloat16)
            self.w2p = torch.zeros(npl, C_, C_, dtype=pdt, device=dev)       # planes of W2 as stored (CAR dgrad)
            self.w2tp = torch.zeros(npl, C_, C_, dtype=pdt, device=dev)      # planes of W2^T (CAR forward)
            if self.h2:      # H2Scale records (32 bytes each, zero-initialised): W2's scale; max row norm of Ws1 (factor of the dZ2 bound)
                self.sc_w2 = torch.zeros(8, dtype=torch.float32, device=dev)
                self.sc_ws1n = torch.zeros(8, dtype=torch.float32, device=dev)
                # constant record of an operand bounded by 1 (cand (.) pred: tanh x tanh): scale 2^14 - a factor of two inside fp16's range
                self.sc_unit = torch.tensor([16384.0, 1.0 / 16384.0, 1.0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev)
            # scorer layer-1 dgrad fused with the cand (.) pred backward (csrc/dm_fused.hip) where the kernel takes the shape; otherwise
            # (and for 1 + N outside [32, 256]) the two separate kernels
            self.dm_fused = L.entries['Ws1'].shape == (C_, 128) and C_ % 64 == 0
            self.ws1p = torch.zeros(3, C_, 128, dtype=torch.bfloat16, device=dev)     # planes of Ws1 as stored
            if self.h2:       # ... and its two fp16 planes under the scale of sc_ws1n (the fused dgrad's own products, CHAM_S1_H2)
                self.ws1h = torch.zeros(2, C_, 128, dtype=torch.float16, device=dev)
        self._plans = {}
        self.max_plans = 24                       # padded lengths T seen in a run (seq_len - 1 = 19 at most for G1)
        self.plan_bytes_budget = 128 << 30        # of the 288 GB: activations of the cached shapes
        self.profile = None           # list -> per-GEMM-launch HIP-event timing (bench.py roofline leg)
        # data-parallel context (set by parallel.DataParallelNAR)
        self.dp_rank, self.dp_world, self.dp_allreduce, self.dp_sharded = 0, 1, None, None
        self.dp_early_bucket, self.dp_gather_slots, self.dp_mode, self.dp_active = None, None, 'allreduce', False

    # ---- views into the flat buffers
    def view(self, flat, name):
        """Tensor view of one logical tensor inside a flat buffer (cached per buffer: a step asks ~70 times)."""
        cache = self._views.get(id(flat))
        if cache is None or cache[0] is not flat:
            cache = self._views[id(flat)] = (flat, {})
        v = cache[1].get(name)
        if v is None:
            e = self.layout.entries[name]
            v = cache[1][name] = flat[e.offset:e.offset + int(np.prod(e.shape))].view(*e.shape)
        return v

    def p(self, name):
        return self.view(self.flat, name)

    def g(self, name):
        return self.view(self.grads, name)

    def logical_weights(self):
        return self.layout.unpack(self.flat.cpu().numpy())

    def logical_grads(self):
        return self.layout.unpack(self.grads.cpu().numpy())

    def load_logical_weights(self, logical):
        self.flat.copy_(torch.from_numpy(self.layout.pack(logical)))
        self.weights_version += 1

    def shadows_stale(self):
        return (self.b16 or self.p3) and (self.global_step, self.weights_version) != self._shadow_key

    def refresh_shadows(self):
        """bf16 shadows (bf16 configuration) / bf16 plane shadows of W2 (plane-resident CAR GEMMs) of the candidate-row GEMM weights,
        once per weight version (one small launch per weight)."""
        if not self.shadows_stale():
            return
        key = (self.global_step, self.weights_version)
        if self.h2:
            C = self.layout.C
            check(self.lib.cham_split2h(ptr(self.p('W2')), C, C, C, ptr(self.w2p), C * C, C, ptr(self.w2tp), C * C, C, ptr(self.sc_w2), 1, _stream()),
                  "cham_split2h")
            k1 = self.layout.entries['Ws1'].shape[1]
            check(self.lib.cham_h2_scale_rownorm(ptr(self.p('Ws1')), C, k1, k1, None, ptr(self.sc_ws1n), _stream()), "cham_h2_scale_rownorm")
        elif self.p3:
            C = self.layout.C
            check(self.lib.cham_split3(ptr(self.p('W2')), C, C, C, ptr(self.w2p), C * C, C, ptr(self.w2tp), C * C, C, _stream()), "cham_split3")
        if self.p3:
            C = self.layout.C
            if self.dm_fused:
                check(self.lib.cham_split3(ptr(self.p('Ws1')), C, 128, 128, ptr(self.ws1p), C * 128, 128, None, 0, 0, _stream()), "cham_split3")
                if self.dm_f16:     # (scale: the row-norm record derived just above - no second pass over the weight)
                    check(self.lib.cham_split2h(ptr(self.p('Ws1')), C, 128, 128, ptr(self.ws1h), C * 128, 128, None, 0, 0, ptr(self.sc_ws1n), 0,
                                                _stream()), "cham_split2h")
        if self.b16:
            for name in ('W2', 'Ws1', 'Ws2', 'Ws3'):
                r, c = self.layout.entries[name].shape
                check(self.lib.cham_cast_b16(ptr(self.p(name)), r, c, ptr(self.shadow[name]), ptr(self.shadow[name + 'T']), _stream()),
                      "cham_cast_b16")
        self._shadow_key = key

    def state_dict(self):
        """Weights + Adam slots + step.  Under the sharded data-parallel mode a rank only maintains the slots of the
        parameter slice it owns: they are all-gathered first (a COLLECTIVE - every rank must call state_dict()), so that any
        rank's checkpoint resumes the same optimizer trajectory in any mode (the mode is recorded)."""
        m, v = self.m, self.v
        if getattr(self, 'dp_gather_slots', None) is not None:
            m, v = self.dp_gather_slots(m, v)
        return {'flat': self.flat.cpu(), 'm': m.cpu(), 'v': v.cpu(), 'global_step': self.global_step,
                'dp_mode': getattr(self, 'dp_mode', 'allreduce'), 'dp_world': self.dp_world, 'layout': self.layout.fingerprint()}

    def load_state_dict(self, sd):
        """Restores weights + Adam slots; refuses a checkpoint written for another parameter layout (a changed feature config, cell
        type or CAR / RNN width with the same element count would otherwise be restored into the wrong offsets)."""
        if sd.get('layout') is not None and sd['layout'] != self.layout.fingerprint():
            raise ValueError("checkpoint was written for a different parameter layout (feature config / rnn_cell / sizes changed)")
        if tuple(sd['flat'].shape) != tuple(self.flat.shape):
            raise ValueError("checkpoint holds %d parameters, the model %d" % (sd['flat'].numel(), self.flat.numel()))
        self.flat.copy_(sd['flat']); self.m.copy_(sd['m']); self.v.copy_(sd['v'])
        self.global_step = int(sd['global_step'])
        self.weights_version += 1

    def rnn_coop_ws(self, B):
        """Exchange buffers + flags of the cooperative recurrent kernels for batches of B sessions (zero-initialised once; the forward and
        the backward of a step run on the same lane and share it)."""
        ws = self._rnn_coop_ws.get(B)
        if ws is None:
            nb = int(self.lib.cham_rnn_coop_workspace_bytes(B, self.layout.Hp))
            ws = self._rnn_coop_ws[B] = torch.zeros(nb, dtype=torch.uint8, device=self.device)
        return ws

    def rnn_coop_timed_out(self):
        """True if a cooperating recurrent workgroup ever gave up a bounded spin (synchronises; tests / bench / end of Estimator.train)."""
        return any(int(self.lib.cham_rnn_coop_timeouts(ptr(ws), B, self.layout.Hp, _stream())) != 0 for B, ws in self._rnn_coop_ws.items())

    def plan(self, B, T, N, n_buf, Bg=None):
        """Buffers for one batch shape, cached: ragged hourly files produce a handful of padded lengths T.  Least-recently-used
        plans are dropped one at a time (count / byte budget), never all at once."""
        key = (B, T, N, n_buf, Bg or B, self.p3, self.h2)      # (the plane buffers of a plan follow the arithmetic it was built for)
        pl = self._plans.pop(key, None)
        if pl is None:
            need = StepPlan.estimate_bytes(self.layout, B, T, N, self)
            while self._plans and (len(self._plans) >= self.max_plans or
                                   need + sum(p.nbytes for p in self._plans.values()) > self.plan_bytes_budget):
                self._plans.pop(next(iter(self._plans)))          # oldest entry (dicts keep insertion order)
            pl = StepPlan(self, B, T, N, n_buf, Bg or B)
            pl.nbytes = pl.allocated_bytes()                       # what the eviction above counts for the plans already cached
            self.plans_created = getattr(self, 'plans_created', 0) + 1      # (bookkeeping: a first-seen padded length costs GBs of allocation)
        self._plans[key] = pl                                      # (re)insert as most recently used
        return pl

    def _lane_ws(self, name):
        """The split-K / column-sum workspace of the lane (stream) the caller is enqueuing on: lanes run concurrently."""
        st = _stream()
        if st == self._side_raw:
            return getattr(self, name + '_side')
        if st == self._aux_raw:
            return getattr(self, name + '_aux')
        return getattr(self, name)

    def warm_plans(self, B, Ts, N, n_buf, Bg=None):
        """Builds the buffer set of every padded length in ``Ts`` now (a trainer's warm-up): hourly session files produce a handful of
        padded lengths T <= truncate_session_length - 1, and the first batch of each would otherwise allocate and zero-fill several GB
        in the middle of training (tens of ms).  Bounded by ``max_plans`` / ``plan_bytes_budget`` like any other plan."""
        for T in Ts:
            self.plan(B, int(T), N, n_buf, Bg)

    # ---- thin kernel wrappers -------------------------------------------------------------------------
    def gemm(self, A, B, C, M, N, K, lda, ldb, ldc, transA=0, transB=0, bias=None, act=ACT_NONE, dref=None, ldr=0,
             dact=ACT_NONE, rowscale=None, ldrs=0, rs_div=1, accumulate=0, splits=1, force_f32=False, h2scales=None):
        """h2scales = (record of A [x rowscale], record of B): the two-fp16-plane form of the bf16x3 kernel (cham_gemm_f32x2h; default
        arithmetic only - the other arithmetics ignore it)."""
        ws = None
        if splits != 1:
            ws = self._lane_ws('gemm_ws')
        prof = self.profile
        bf16 = self.gemm_dtype == 'bf16' and not force_f32
        x3 = self.x3 and not force_f32
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0 = self._tile_counts()
            x0 = self._tile_counts_x3() if x3 else None
            e0.record()       # torch's current stream == the stream the kernel is launched on (_stream())
        x2h = h2scales is not None and x3 and not bf16 and self.s1_h2
        if x2h:
            if dref is not None or transB:
                raise ValueError("the two-plane form takes the NN and TN shapes only")
            check(self.lib.cham_gemm_f32x2h(ptr(A), lda, transA, ptr(B), ldb, transB, ptr(C), ldc, M, N, K, ptr(bias), act, ptr(rowscale), ldrs,
                                            rs_div, accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, ptr(h2scales[0]),
                                            ptr(h2scales[1]), _stream()), "cham_gemm_f32x2h")
        else:
            fn = self.lib.cham_gemm_bf16 if bf16 else (self.lib.cham_gemm_f32x3 if x3 else self.lib.cham_gemm_f32)
            check(fn(ptr(A), lda, transA, ptr(B), ldb, transB, ptr(C), ldc, M, N, K, ptr(bias), act,
                     ptr(dref), ldr, dact, ptr(rowscale), ldrs, rs_div, accumulate, ptr(ws),
                     ws.numel() * 4 if ws is not None else 0, splits, _stream()), "cham_gemm_f32")
        if prof is not None:
            e1.record()
            c1 = self._tile_counts()
            tile = next((i for i in range(13) if c1[i] != c0[i]), -1)
            rec = dict(M=M, N=N, K=K, transA=transA, transB=transB, splits=int(c1[15]), act=act, dref=dref is not None, dact=dact,
                       bias=bias is not None, rowscale=rowscale is not None, bf16=bf16, tile=tile, epi=int(c1[14]), ev=(e0, e1))
            if x3:
                x1 = self._tile_counts_x3()
                xt = next((i for i in (0, 1, 2, 4, 5) if x1[i] != x0[i]), -1)
                if xt >= 0:          # ran on a bf16x3 instance (otherwise: delegated to the native kernels, recorded above)
                    rec.update(x3=True, tile=xt & 3, x2h=xt >= 4, epi=int(x1[6]), splits=int(x1[7]))
            prof.append(rec)

    def _tile_counts(self):
        import ctypes
        out = (ctypes.c_longlong * 16)()
        self.lib.cham_gemm_launch_counts(out, 0)
        return list(out)

    def _tile_counts_x3(self):
        import ctypes
        out = (ctypes.c_longlong * 8)()
        self.lib.cham_gemm_f32x3_launch_counts(out, 0)
        return list(out)

    def gemm_b16(self, A, lda, transA, B, ldb, transB, C, ldc, out_f32, M, N, K, bias=None, act=ACT_NONE, dref=None, ldr=0,
                 dact=ACT_NONE, accumulate=0, splits=1, dma=False):
        """bf16-resident GEMM (csrc/gemm_b16.hip): NT (transA=0, transB=1) or TN (transA=1, transB=0).  dma: the LDS-DMA core
        (cham_gemm_b16_dma; NT with bf16 output / TN with fp32 output, M and N multiples of 256 for TN)."""
        ws = None
        if splits != 1:
            ws = self._lane_ws('gemm_ws')
        prof = self.profile
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0 = self._tile_counts_b16()
            e0.record()
        if dma:
            if prof is not None:
                import ctypes
                cw0 = (ctypes.c_longlong * 8)()
                self.lib.cham_gemm_p3_launch_counts(cw0, 0)
            check(self.lib.cham_gemm_b16_dma(ptr(A), lda, ptr(B), ldb, transA, ptr(C), ldc, M, N, K, ptr(bias), act, ptr(dref), ldr, dact,
                                             accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, _stream()), "cham_gemm_b16_dma")
            if prof is not None:
                e1.record()
                import ctypes
                c = (ctypes.c_longlong * 8)()
                self.lib.cham_gemm_p3_launch_counts(c, 0)
                prof.append(dict(M=M, N=N, K=K, transA=transA, transB=transB, splits=int(c[7]), act=act, dref=dref is not None, dact=dact,
                                 bias=bias is not None, rowscale=False, bf16=True, b1=True, b1w=bool(c[4] > cw0[4]), out_f32=int(transA), tile=0, epi=int(c[6]),
                                 ev=(e0, e1)))
            return
        check(self.lib.cham_gemm_b16(ptr(A), lda, transA, ptr(B), ldb, transB, ptr(C), ldc, out_f32, M, N, K, ptr(bias), act, ptr(dref),
                                     ldr, dact, accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, _stream()),
              "cham_gemm_b16")
        if prof is not None:
            e1.record()
            c1 = self._tile_counts_b16()
            tile = next((i for i in range(5) if c1[i] != c0[i]), -1)
            prof.append(dict(M=M, N=N, K=K, transA=transA, transB=transB, splits=int(c1[7]), act=act, dref=dref is not None, dact=dact,
                             bias=bias is not None, rowscale=False, bf16=True, b16=True, out_f32=int(c1[5]), tile=tile, epi=int(c1[6]),
                             ev=(e0, e1)))

    def gemm_p3(self, A, a_ps, lda, B, b_ps, ldb, tn, C, ldc, M, N, K, bias=None, act=ACT_NONE, dref_h=None, ldr=0, dact=ACT_NONE,
                accumulate=0, splits=1):
        """Plane-product GEMM over pre-split bf16 planes (csrc/gemm_p3.hip): NT (tn=0) or TN (tn=1, split-K)."""
        ws = None
        if splits != 1:
            ws = self._lane_ws('gemm_ws')
        prof = self.profile
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(self.lib.cham_gemm_p3(ptr(A), a_ps, lda, ptr(B), b_ps, ldb, tn, ptr(C), ldc, M, N, K, ptr(bias), act, ptr(dref_h), ldr, dact,
                                    accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, _stream()), "cham_gemm_p3")
        if prof is not None:
            e1.record()
            import ctypes
            c = (ctypes.c_longlong * 8)()
            self.lib.cham_gemm_p3_launch_counts(c, 0)
            prof.append(dict(M=M, N=N, K=K, transA=tn, transB=0 if tn else 1, splits=int(c[7]), act=act, dref=dref_h is not None, dact=dact,
                             bias=bias is not None, rowscale=False, bf16=False, p3=True, tile=0, epi=int(c[6]), ev=(e0, e1)))

    def gemm_h2(self, A, a_ps, lda, a_sc, B, b_ps, ldb, b_sc, tn, C, ldc, M, N, K, bias=None, act=ACT_NONE, dref_h=None, ldr=0, dact=ACT_NONE,
                accumulate=0, splits=1, a_tiles=0, b_tiles=0, dref_blocked=0, group_rows=0, groupsum=None):
        """Plane-product GEMM over two fp16 planes + scale records (csrc/gemm_h2.hip): NT (tn=0) or TN (tn=1, split-K).  a_tiles / b_tiles > 0:
        that operand is TILE-BLOCKED with this many row tiles per plane (include/chameleon_nar.h cham_gemm_h2b), dref_blocked: dref_h likewise.
        groupsum (dgrad only): the epilogue also leaves the column sums of every group of `group_rows` rows (cham_gemm_h2_dgrad_gs)."""
        ws = None
        if splits != 1:
            ws = self._lane_ws('gemm_ws')
        prof = self.profile
        if prof is not None:
            import ctypes
            c0 = (ctypes.c_longlong * 8)()
            self.lib.cham_gemm_h2_launch_counts(c0, 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if groupsum is not None:
            check(self.lib.cham_gemm_h2_dgrad_gs(ptr(A), a_ps, lda, ptr(a_sc), ptr(B), b_ps, ldb, ptr(b_sc), ptr(C), ldc, M, N, K, ptr(dref_h), ldr,
                                                 a_tiles, dref_blocked, group_rows, ptr(groupsum), groupsum.numel() * 4, _stream()), "cham_gemm_h2_dgrad_gs")
        else:
            check(self.lib.cham_gemm_h2b(ptr(A), a_ps, lda, ptr(a_sc), ptr(B), b_ps, ldb, ptr(b_sc), tn, ptr(C), ldc, M, N, K, ptr(bias), act, ptr(dref_h),
                                         ldr, dact, accumulate, ptr(ws), ws.numel() * 4 if ws is not None else 0, splits, a_tiles, b_tiles, dref_blocked,
                                         _stream()), "cham_gemm_h2b")
        if prof is not None:
            e1.record()
            c = (ctypes.c_longlong * 8)()
            self.lib.cham_gemm_h2_launch_counts(c, 0)
            prof.append(dict(M=M, N=N, K=K, transA=tn, transB=0 if tn else 1, splits=int(c[7]), act=act, dref=dref_h is not None, dact=dact,
                             bias=bias is not None, rowscale=False, bf16=False, h2=True, h2w=bool(c[2] > c0[2]), h2blk=bool(c[3] > c0[3]), tile=0, epi=int(c[6]), ev=(e0, e1)))

    def _tile_counts_b16(self):
        import ctypes
        out = (ctypes.c_longlong * 8)()
        self.lib.cham_gemm_b16_launch_counts(out, 0)
        return list(out)

    def colsum(self, X, ld, R, F, out, w=None, accumulate=0, b16=False):
        ws = self._lane_ws('colsum_ws')
        check((self.lib.cham_colsum_b16 if b16 else self.lib.cham_colsum)(ptr(X), ld, R, F, ptr(w), ptr(out), accumulate, ptr(ws),
                                                                           ws.numel() * 4, _stream()), "cham_colsum")

    def set_step_scalars(self, step=0, max_ts=0, sum_mask=1.0, lr_t=0.0, fields=7, stream=None):
        """Writes the selected fields of the step-scalar record (fields: 1 = sampler keys (step, step + 1), 2 = max_ts + sum_mask, 4 = lr_t) on
        `stream` (default: the current one), in stream order in front of the launches that read it.  No-op during a graph capture."""
        if self.capturing:
            return
        if self.scalars is None:
            self.scalars = torch.zeros(max(8, self.lib.cham_step_scalars_bytes() // 4), dtype=torch.int32, device=self.device)
        check(self.lib.cham_step_scalars_set(ptr(self.scalars), int(step) & 0xFFFFFFFF, (int(step) + 1) & 0xFFFFFFFF, int(max_ts), float(sum_mask),
                                             float(lr_t), fields, _stream() if stream is None else stream), "cham_step_scalars_set")

    def side(self):
        """Context manager: run the enclosed launches on the side stream (or inline when overlap is disabled)."""
        import contextlib
        return torch.cuda.stream(self.side_stream) if self.overlap else contextlib.nullcontext()

    def fork(self):
        if self.overlap:
            self.side_stream.wait_stream(torch.cuda.current_stream())

    def join(self):
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.side_stream)


def _h2b_block():
    from .. import _lib
    return int(_lib.load().cham_h2b_block_elements())


def blocked_plane_elements(tiles, C):
    """Elements of ONE tile-blocked plane of `tiles` row tiles and leading dimension C (csrc/common.h h2b_index)."""
    return tiles * (C // 32) * _h2b_block()


def planes_from_blocked(t, tiles, C=None):
    """[planes, >= tiles * (C / 32) * block] memory in the TILE-BLOCKED layout ([row tile][C / 32][256][32] (+ padding), csrc/common.h
    h2b_index) -> the row-major [planes, tiles * 256, C] matrix (a copy; tests and debugging)."""
    npl = t.shape[0]
    C = t.shape[2] if C is None else C
    blk = _h2b_block()
    flat = t.reshape(npl, -1)[:, :tiles * (C // 32) * blk]
    return flat.reshape(npl, tiles, C // 32, blk)[..., :8192].reshape(npl, tiles, C // 32, 256, 32).permute(0, 1, 3, 2, 4).reshape(npl, tiles * 256, C)


def planes_to_blocked(t):
    """Row-major [planes, R, C] (C % 32 == 0) -> (tile-blocked [planes, tiles * (C / 32) * block] with the rows beyond R zero, tiles)."""
    npl, R, C = t.shape
    tiles = -(-R // 256)
    blk = _h2b_block()
    pad = torch.zeros(npl, tiles * 256, C, dtype=t.dtype, device=t.device)
    pad[:, :R] = t
    out = torch.zeros(npl, tiles, C // 32, blk, dtype=t.dtype, device=t.device)
    out[..., :8192] = pad.reshape(npl, tiles, 256, C // 32, 32).permute(0, 1, 3, 2, 4).reshape(npl, tiles, C // 32, 8192)
    return out.reshape(npl, -1), tiles


class StepPlan:
    """Device buffers for one (B, T, N) shape.  Row layouts: see csrc/scorer.hip."""

    @staticmethod
    def estimate_bytes(L, B, T, N, rt=None):
        """Dominant buffers only (decides what to evict BEFORE the plan exists; the plan then records what it really allocated):
        Z1, Z2, dZ1, dZ2 over all CAR rows + the scorer activations over the candidate rows + the plane-resident operands of the
        candidate-row CAR GEMMs (Z1 and dZ2 as two fp16 / three bf16 planes), the b2 partial sums and the segment table."""
        rows = B * T * (N + 2)
        need = 4 * (4 * rows * L.C + 2 * rows * (128 + 64 + 32))        # (the bf16 configuration needs ~5/8 of this)
        if rt is not None and rt.p3:
            rc = B * T * (N + 1)
            need += 2 * (2 if rt.h2 else 3) * 2 * rc * L.C + 4 * B * T * L.C
            need += 4 * int(rt.lib.cham_group_rows_segments_len(2 * B * T + 20 * N + 1))
        return need

    def allocated_bytes(self):
        """Bytes of device memory this plan holds (every tensor attribute, lists and dicts of tensors included)."""
        seen, total = set(), 0

        def walk(x):
            nonlocal total
            if torch.is_tensor(x):
                st = x.untyped_storage()
                if st.data_ptr() not in seen:
                    seen.add(st.data_ptr())
                    total += st.nbytes()
            elif isinstance(x, (list, tuple)):
                for y in x:
                    walk(y)
            elif isinstance(x, dict):
                for y in x.values():
                    walk(y)
        for v in vars(self).values():
            walk(v)
        return total

    def __init__(self, rt, B, T, N, n_buf, Bg):
        L, dev = rt.layout, rt.device
        self.B, self.T, self.N, self.n_buf, self.Bg = B, T, N, n_buf, Bg
        self.BT = BT = B * T
        self.NC = NC = N + 1
        self.Rc = Rc = BT * NC
        self.Rall = Rall = BT + Rc
        self.pmax = pmax = 20 * N
        self.RV = RV = 2 * BT + pmax + 1
        C, Hp, Fc, Fi = L.C, L.Hp, L.Fc, L.Fi
        self.C = C
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        i64 = lambda *s: torch.zeros(*s, dtype=torch.int64, device=dev)
        # sampler
        # two sets of sampler outputs: the next step's negatives can be drawn (NARModuleModel.presample) while this step's
        # backward still reads its own
        self._samp = [dict(neg_ids=i64(B, T, N), neg_slot=torch.zeros(B, T, N, dtype=torch.int32, device=dev), pool=i64(pmax),
                           canon=torch.zeros(pmax, dtype=torch.int32, device=dev), meta=torch.zeros(4, dtype=torch.int32, device=dev))
                      for _ in range(2)]
        self.use_sampler_set(0)
        need = rt.lib.cham_combine_bwd_workspace_bytes(L.C, B * T, N, 20 * N)
        if need > rt.gemm_ws.numel() * 4:
            rt.gemm_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        if need > rt.gemm_ws_side.numel() * 4:
            rt.gemm_ws_side = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        self.ws_bytes = rt.lib.cham_neg_sample_workspace_bytes(Bg * (T + 1), int(rt.params['recent_clicks_buffer_max_size']), n_buf)
        self.sampler_ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        # features
        self.ids_all = i64(RV)
        self.ref_ts = i64(RV)
        self.rec_raw, self.nov_raw = f32(RV), f32(RV)
        self.stats = f32(3, 8)
        self.stat_scratch = f32(3 * max(1, int(rt.params['recent_clicks_for_normalization'])))
        self.w_rows = f32(RV)
        self.perm = torch.zeros(RV, dtype=torch.int32, device=dev)          # item rows grouped by id (cham_group_rows)
        self.seg = torch.zeros(int(rt.lib.cham_group_rows_segments_len(RV)), dtype=torch.int32, device=dev)      # its segment table
        self.group_ws = torch.zeros(int(rt.lib.cham_group_rows_workspace_bytes(RV)) // 4, dtype=torch.int32, device=dev)
        self.Xc_raw, self.Xc_s, self.dXc = f32(BT, Fc), f32(BT, Fc), f32(BT, Fc)
        self.Xi_raw, self.Xi_s, self.dXi = f32(RV, Fi), f32(RV, Fi), f32(RV, Fi)
        # CAR
        self.U, self.dU = f32(BT, C), f32(BT, C)
        self.V, self.dV = f32(RV, C), f32(RV, C)
        b16 = rt.b16
        bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        cand = bf if b16 else f32                 # storage of the matrices with one row per candidate
        if b16:       # clicked-input rows stay fp32 (they feed / come from the fp32 recurrent branch); candidate rows are bf16
            self.Z1, self.Z2, self.dZ2, self.dZ1 = f32(BT, C), f32(BT, C), f32(BT, C), f32(BT, C)
            self.Z1c, self.Z2c, self.dZ2c, self.dZ1c, self.Mc = bf(Rc, C), bf(Rc, C), bf(Rc, C), bf(Rc, C), bf(Rc, C)
            self.b2part = f32(BT, C)
        else:         # one [BT + Rc, C] matrix each: clicked-input rows first (Z1c ... are views taken per step: BT = valid positions)
            # With the plane-resident CAR GEMMs the candidate rows of Z1 exist as planes only, and with the fused scorer dgrad those of
            # dZ2 too: the fp32 matrices keep the clicked-input rows (ensure_rows() grows them on the paths that do need all rows -
            # dropout, NC outside the fused kernel's range)
            fused = rt.p3 and rt.dm_fused and 32 <= NC <= 256
            self.Z1 = f32(BT if rt.p3 else Rall, C)
            self.Z2 = f32(Rall, C)
            self.dZ2 = f32(BT if fused else Rall, C)
            self.dZ1 = f32(Rall, C)
        self.p3 = rt.p3
        self.h2 = rt.h2
        if rt.p3:     # plane-resident operands of the three candidate-row CAR GEMMs (planes Rc * C elements apart) + b2 partial sums
            self.z1_tiles = self.dz2_tiles = 0
            if rt.h2:      # two fp16 planes + the scale records of the two matrices
                # TILE-BLOCKED planes (round 6; csrc/common.h h2b_index): [row tiles of 256][C / 32][256][32] - what the NT GEMMs fetch per
                # request is whole 128-byte lines.  Z1's planes always (cham_combine_fwd_h2b writes them); dZ2's when the fused scorer dgrad
                # is their producer (cham_dm_mulpred_h2_blk; the unfused cham_mulpred_bwd_h2 writes row-major).  Row tiles are allocated
                # whole (+ one: a workgroup of the fused dgrad addresses two tiles from its first row's), zero-initialised.
                tiles = -(-Rc // 256) + 1
                if rt.h2_blocked:
                    self.z1_tiles = tiles if rt.h2_blocked_which in ("1", "z1") else 0
                    self.dz2_tiles = tiles if (rt.dm_fused and 32 <= NC <= 256 and rt.h2_blocked_which in ("1", "dz2")) else 0
                self.Z1p = (torch.zeros(2, blocked_plane_elements(self.z1_tiles, C), dtype=torch.float16, device=dev) if self.z1_tiles
                            else torch.zeros(2, Rc, C, dtype=torch.float16, device=dev))
                self.dZ2p = (torch.zeros(2, blocked_plane_elements(self.dz2_tiles, C), dtype=torch.float16, device=dev) if self.dz2_tiles
                             else torch.zeros(2, Rc, C, dtype=torch.float16, device=dev))
                self.sc_z1 = torch.zeros(8, dtype=torch.float32, device=dev)
                self.sc_dz2 = torch.zeros(8, dtype=torch.float32, device=dev)
                self.sc_ds1 = torch.zeros(8, dtype=torch.float32, device=dev)          # dS1 itself (operand of the Ws1 weight gradient)
                # per-click column sums of dZ1 from the CAR dgrad's epilogue (round 6; csrc/gemm_h2.hip H2Params::gsum): dU without a second
                # pass over the 1 GB of candidate rows
                self.gsum = (torch.empty(int(rt.lib.cham_gemm_h2_groupsum_bytes(Rc, C, NC)) // 4, dtype=torch.float32, device=dev)
                             if rt.dgrad_groupsum and NC >= 32 and Rc > 0 else None)
            else:
                self.Z1p, self.dZ2p = bf(3, Rc, C), bf(3, Rc, C)
            self.p3_ps = Rc * C                                   # plane stride of a row-major operand ...
            self.z1_ps = self.Z1p[0].numel()                      # ... and of each operand as allocated
            self.dz2_ps = self.dZ2p[0].numel()
            self.b2part = f32(BT, C)
        # RNN
        self.seq_len = torch.zeros(B, dtype=torch.int32, device=dev)
        NG = L.NG
        self.xproj = [f32(BT, NG * Hp) for _ in range(L.L)]
        self.dxproj = f32(BT, NG * Hp)
        self.rnn_out = [f32(BT, Hp) for _ in range(L.L)]
        self.hprev = [f32(BT, Hp) for _ in range(L.L)]
        self.G = [f32(BT, Hp) for _ in range(L.L)]
        self.Cc = [f32(BT, Hp) for _ in range(L.L)]
        gru = L.cell == 'gru'
        self.R = [f32(BT, Hp) if gru else None for _ in range(L.L)]
        self.RH = [f32(BT, Hp) if gru else None for _ in range(L.L)]
        self.drnn = f32(BT, Hp)
        self.WhT = f32(NG * Hp, Hp)
        if L.rnn_stepwise:
            self.h_state, self.zh, self.carry = f32(B, Hp), f32(B, 2 * Hp), f32(B, Hp)
            self.dzs, self.direct = f32(B, 2 * Hp), f32(B, Hp)
        # FCs / scorer
        self.FC1, self.dFC1 = f32(BT, 512), f32(BT, 512)
        self.pred, self.dpred = f32(BT, C), f32(BT, C)
        self.S1, self.dS1 = cand(Rc, 128), cand(Rc, 128)
        self.S2, self.dS2 = cand(Rc, 64), cand(Rc, 64)
        self.S3, self.dS3 = cand(Rc, 32), cand(Rc, 32)
        self.ds = f32(Rc)
        self.logits, self.probs = f32(BT, NC), f32(BT, NC)
        self.nll = f32(BT)
        self.nov_aux = f32(BT, 3)
        self.mask = torch.zeros(BT, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(3, dtype=torch.float32, device=dev)
        # valid-position compaction (see NARModuleModel.upload_batch): compact copies of the sampler output, and the
        # [B, T]-layout staging buffers either side of the recurrent stack
        self.neg_ids_c = i64(BT, N)
        self.neg_slot_c = torch.zeros(BT, N, dtype=torch.int32, device=dev)
        self.Z2f, self.rnn_c, self.drnn_c, self.dxproj_c = f32(BT, C), f32(BT, Hp), f32(BT, Hp), f32(BT, NG * Hp)
        self.pos, self.P = None, BT
        # The zero fills above run on the stream that built the plan.  A stream that touches these buffers without being ordered behind
        # that point must wait for this event first: presample() may be the FIRST user of a plan (a padded length T seen for the first
        # time), it writes the sampler outputs on the state's stream - and the fills, still queued behind the running step, would wipe them.
        self.created = torch.cuda.Event()
        self.created.record()

    def ensure_rows(self, name):
        """The fp32 matrix `name` with one row per CAR row ([BT + Rc, C]): allocated in full on first need (see __init__).  Called before
        the step's first write to it."""
        t = getattr(self, name)
        if t.shape[0] < self.Rall:
            t = torch.empty(self.Rall, t.shape[1], dtype=t.dtype, device=t.device)
            setattr(self, name, t)
            if hasattr(self, 'nbytes'):         # what the plan cache's byte budget counts (NARRuntime.plan)
                self.nbytes = self.allocated_bytes()
        return t

    def dropout_buffers(self, rt):
        """Buffers of the dropout path (keep_prob < 1: dense PreCAR input rows, dropped FC1 / recurrent outputs) - allocated on first use."""
        if not rt.b16:
            self.ensure_rows('Z1'); self.ensure_rows('dZ2')
        if getattr(self, 'Xd', None) is None:
            L, dev = rt.layout, rt.device
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            Fw = L.Fc + L.Fi
            self.Xd, self.dXd = f32(self.Rall, Fw), f32(self.Rall, Fw)
            self.dUx, self.dVx = f32(self.BT, Fw), f32(self.RV, Fw)
            self.FC1d = f32(self.BT, 512)
            self.rnn_drop = [f32(self.BT, L.Hp) for _ in range(L.L)]
            need = rt.lib.cham_combine_bwd_workspace_bytes(Fw, self.B * self.T, self.N, self.pmax)
            self.drop_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
            if rt.b16:      # bf16 configuration: the dense PreCAR layer computes / consumes fp32 images of the bf16-resident candidate rows
                self.Z1f, self.dZ1f = f32(self.Rall, L.C), f32(self.Rall, L.C)

    def cand_Z1(self, P=None):
        """Candidate rows of the PreCAR output of the last step as fp32 [P * NC, C] (tests): the fp32 matrix, or the sum of its planes."""
        P = self.P if P is None else P
        n = P * self.NC
        if getattr(self, 'used_p3', False):
            if getattr(self, 'z1_tiles', 0):       # tile-blocked planes -> row-major
                z = planes_from_blocked(self.Z1p, self.z1_tiles, self.C)[:, :n].float()
            else:
                z = self.Z1p[:, :n].float()
            if self.h2:
                return (z[0] + z[1]) * self.sc_z1[1]
            return z[0] + z[1] + z[2]
        if self.Z1.shape[0] == self.BT and hasattr(self, 'Z1c'):
            return self.Z1c[:n].float()
        return self.Z1[P:P + n]

    def use_sampler_set(self, k):
        self._samp_cur = k
        for name, t in self._samp[k].items():
            setattr(self, name, t)

    def full_rows(self, x, group=1):
        """Debug / test helper: rows of the current step (one group of ``group`` rows per valid position) -> the [B*T*group, ...]
        layout with zeros at padded positions."""
        x = x[:self.P * group]
        if self.pos is None:
            return x
        out = torch.zeros((self.BT * group,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        out.view(self.BT, -1)[self.pos.long()] = x.reshape(self.P, -1)
        return out


class NARModuleModel:
    """Reference constructor signature: nar_module/nar/nar_model.py:102-129 (+ ``runtime``, the device-resident
    state that TF keeps in its checkpoint / session)."""

    def __init__(self, mode, inputs, labels, session_features_config, articles_features_config, batch_size, lr, keep_prob,
                 negative_samples, negative_sample_from_buffer, content_article_embeddings_matrix, rnn_num_layers=1,
                 softmax_temperature=1.0, reg_weight_decay=0.0, recent_clicks_buffer_hours=1.0,
                 recent_clicks_buffer_max_size=1000, recent_clicks_for_normalization=1000, articles_metadata=None,
                 plot_histograms=False, metrics_top_n=5, elapsed_days_smooth_log_base=1.3, popularity_smooth_log_base=2.0,
                 CAR_embedding_size=256, rnn_units=256, max_cardinality_for_ohe=10, novelty_reg_factor=0.0,
                 diversity_reg_factor=0.0,
                 internal_features_config={'recency': True, 'novelty': True, 'article_content_embeddings': True,
                                           'item_clicked_embeddings': True},
                 eval_cold_start=False, runtime=None, rnn_cell='ugrnn', gemm_dtype='f32'):
        # log_base / log_1p bases of the recency and novelty features (nar_model.py:28-34, 1071-1075, 1148): launch scalars of the kernels
        for b in (elapsed_days_smooth_log_base, popularity_smooth_log_base):
            if not (b > 0.0) or b == 1.0:
                raise ValueError("log bases must be > 0 and != 1")
        self.elapsed_days_smooth_log_base = float(elapsed_days_smooth_log_base)
        self.popularity_smooth_log_base = float(popularity_smooth_log_base)
        self.novelty_reg_factor = float(novelty_reg_factor)
        self.is_training = (mode == ModeKeys.TRAIN)
        if self.is_training and not (0.0 < keep_prob <= 1.0):
            raise ValueError("dropout_keep_prob must be in (0, 1]")
        self.mode = mode
        self.inputs, self.labels = inputs, labels
        self.lr, self.keep_prob = lr, keep_prob
        self.negative_samples = negative_samples
        self.negative_sample_from_buffer = negative_sample_from_buffer
        self.softmax_temperature = softmax_temperature
        self.reg_weight_decay = reg_weight_decay
        self.metrics_top_n = metrics_top_n
        self.recent_clicks_for_normalization = recent_clicks_for_normalization
        self.recent_clicks_buffer_max_size = recent_clicks_buffer_max_size
        self.eval_cold_start = eval_cold_start
        if runtime is None and _VARIABLE_STORE is not None and _VARIABLE_STORE.get('runtime') is not None:
            runtime = _VARIABLE_STORE['runtime']
        if runtime is None:
            seed = _VARIABLE_STORE.get('tf_random_seed', 42) if _VARIABLE_STORE is not None else 42
            runtime = NARRuntime(dict(tf_random_seed=seed,session_features_config=session_features_config,
                                      articles_features_config=articles_features_config,
                                      content_article_embeddings_matrix=content_article_embeddings_matrix,
                                      articles_metadata=articles_metadata, CAR_embedding_size=CAR_embedding_size,
                                      rnn_units=rnn_units, rnn_num_layers=rnn_num_layers, rnn_cell=rnn_cell, gemm_dtype=gemm_dtype,
                                      internal_features_config=internal_features_config,
                                      max_cardinality_for_ohe=max_cardinality_for_ohe,
                                      recent_clicks_buffer_max_size=recent_clicks_buffer_max_size,
                                      recent_clicks_for_normalization=recent_clicks_for_normalization), seed=seed)
            if _VARIABLE_STORE is not None:
                _VARIABLE_STORE['runtime'] = runtime
        self.rt = runtime
        # state "placeholders" (nar_model.py:195-202): fed by ItemsStateUpdaterHook.before_run
        self.articles_recent_pop_norm = None
        self.pop_recent_items_buffer = None
        self._dev_state = None
        self.total_loss = None
        self._accumulating = False             # train_step_microbatched: gradients of a micro-batch are not final
        self.train = self.train_step           # the reference's ``model.train`` op
        self._eval_iter = 0
        self._eval = None
        # attributes ItemsStateUpdaterHook fetches / feeds (nar_model.py:1435-1467)
        inp = lambda k: Tensor(k, lambda: np.asarray(self.inputs[k]))
        self.item_clicked = inp('item_clicked')
        self.event_timestamp = Tensor('event_timestamp', lambda: np.asarray(self.inputs['event_timestamp'])[..., None])   # :233
        self.session_id, self.user_id = inp('session_id'), inp('user_id')
        self.next_item_label = Tensor('next_item_label', lambda: np.asarray(self.labels['label_next_item']))
        self.label_last_item = Tensor('label_last_item', lambda: np.asarray(self.labels['label_last_item']))
        self.batch_negative_items = Tensor('batch_negative_items', lambda: self._plan.neg_ids)
        self.predicted_item_ids = Tensor('predicted_item_ids', lambda: self._eval['pred_ids'])
        self.predicted_item_probs = Tensor('predicted_item_probs', lambda: self._eval['pred_probs'])
        self.label_rank = Tensor('label_rank', lambda: self._eval['label_rank'])
        self.batch_items_count = Tensor('batch_items_count', lambda: self._batch_counts()[0])          # :258
        self.batch_unique_items_count = Tensor('batch_unique_items_count', lambda: self._batch_counts()[1])   # :261
        self.loss_t = Tensor('total_loss', lambda: self.total_loss)
        self.ph_articles_recent_pop_norm = Placeholder('articles_recent_pop_norm')      # :195
        self.ph_pop_recent_items_buffer = Placeholder('pop_recent_items_buffer')        # :200
        self.ph_content_article_embeddings_matrix = Placeholder('content_article_embeddings_matrix')   # :170
        self.ph_articles_metadata = {n: Placeholder('articles_metadata/' + n) for n in (articles_metadata or {})}

    def _batch_counts(self):
        ic = np.asarray(self.inputs['item_clicked'])
        nz = ic[ic != 0]
        return int(nz.shape[0]), int(np.unique(nz).shape[0])

    def feed(self, feed_dict):
        """session.run(feed_dict=...) of the hook (nar_model.py:1458-1467).  The ACE matrix / metadata placeholders are
        accepted and ignored when they are the arrays already resident in HBM (the reference re-feeds them every step
        only to keep them out of the checkpoint)."""
        by_name = {(k.name if isinstance(k, Placeholder) else k): v for k, v in feed_dict.items()}
        self.feed_state(by_name['articles_recent_pop_norm'], by_name['pop_recent_items_buffer'])

    # ------------------------------------------------------------------ host -> device
    def feed_device_state(self, state):
        """Device-resident state (clicked_items_state.DeviceClickedItemsState): nothing is uploaded."""
        state.sync_to_current()           # the previous batch's update runs on the state's own stream
        self.articles_recent_pop_norm = state
        self.pop_recent_items_buffer = state
        self._dev_state = dict(buffer=state.buf_ids, pop_norm=state.pop_norm, last=state.buf_ids,
                               n_last=min(self.recent_clicks_for_normalization, state.buf_ids.numel()) if state.n_updates > 0 else 0,
                               device=True)

    def feed_state(self, pop_norm, buffer_ids):
        """The hook's feed_dict (nar_model.py:1458-1463)."""
        if getattr(pop_norm, 'is_device', False):
            return self.feed_device_state(pop_norm)
        self.articles_recent_pop_norm = pop_norm
        self.pop_recent_items_buffer = buffer_ids
        dev = self.rt.device
        buf = np.ascontiguousarray(buffer_ids, dtype=np.int64)
        nz = buf[buf != 0][: self.recent_clicks_for_normalization]           # nar_model.py:1041-1044
        self._dev_state = dict(
            buffer=torch.from_numpy(buf).to(dev, non_blocking=True),
            pop_norm=torch.from_numpy(np.ascontiguousarray(pop_norm, dtype=np.float32)).to(dev, non_blocking=True),
            last=torch.from_numpy(np.ascontiguousarray(nz)).to(dev, non_blocking=True), n_last=int(nz.shape[0]))

    def upload_batch(self, features, labels, global_features=None, global_labels=None, row_begin=0):
        """numpy batch (input_fn output) -> device tensors + the few host scalars the launch parameters need."""
        L, dev = self.rt.layout, self.rt.device
        item_clicked = np.ascontiguousarray(features['item_clicked'], dtype=np.int64)
        B, T = item_clicked.shape
        gf = features if global_features is None else global_features
        gl = labels if global_labels is None else global_labels
        aci = np.concatenate([np.asarray(gf['item_clicked'], np.int64), np.asarray(gl['label_last_item'], np.int64).reshape(-1, 1)], 1)
        ssz = np.asarray(features['session_size'], dtype=np.int64).reshape(-1)
        seq_len = (ssz - 1).astype(np.int32)                                   # nar_model.py:227
        mask = (np.arange(T)[None, :] < seq_len[:, None])                      # :231
        g_ssz = np.asarray(gf['session_size'], dtype=np.int64).reshape(-1)
        g_T = np.asarray(gf['item_clicked']).shape[1]
        sum_mask = float(np.minimum(np.maximum(g_ssz - 1, 0), g_T).sum())      # global denominator (:664)
        ets = np.ascontiguousarray(features['event_timestamp'], dtype=np.int64)
        max_ts = int(np.asarray(gf['event_timestamp']).max())                  # :235
        cat = np.stack([np.asarray(features[n], np.int64).reshape(-1) for n in L.ctx_cat_names]) if L.ctx_cat_names \
            else np.zeros((1, B * T), np.int64)
        num = np.stack([np.asarray(features[n], np.float32).reshape(-1) for n in L.ctx_num_names]) if L.ctx_num_names \
            else np.zeros((1, B * T), np.float32)
        g_ets = ets if global_features is None else np.ascontiguousarray(gf['event_timestamp'], dtype=np.int64)
        label_next = np.ascontiguousarray(labels['label_next_item'], dtype=np.int64)
        # Valid-position compaction.  The reference computes every padded (session, time) position and multiplies it
        # away with sequence_mask (nar_model.py:231, 664); here the row-wise stages (features, CAR, scorer, softmax and
        # their gradients) run on the P non-padded positions only - same loss and gradients, P/(B*T) of the work
        # (G1-like session lengths: ~1/3).  Only the sampler (keyed by (row, position)) and the recurrent stack keep
        # the [B, T] layout.
        mrows = mask.reshape(-1)
        pos = np.flatnonzero(mrows).astype(np.int32)
        P = int(pos.shape[0])
        host = dict(g_event_ts=g_ets, aci=aci, item_clicked=item_clicked, label_next=label_next, event_ts=ets, seq_len=seq_len)
        compacted = self.rt.compact and 0 < P < B * T
        if compacted:
            host.update(pos=pos, ic_rows=item_clicked.reshape(-1)[pos], ln_rows=label_next.reshape(-1)[pos], ets_rows=ets.reshape(-1)[pos],
                        mask=np.ones(P, np.uint8), cat=cat[:, pos], num=num[:, pos])
        else:
            host.update(mask=mrows.astype(np.uint8), cat=cat, num=num)
        d = dict(B=B, T=T, Bg=aci.shape[0], row_begin=row_begin, sum_mask=sum_mask, max_ts=max_ts, P=P if compacted else B * T, pos=None)
        d.update(self._h2d(host))
        if not compacted:
            d.update(ic_rows=d['item_clicked'].view(-1), ln_rows=d['label_next'].view(-1), ets_rows=d['event_ts'].view(-1))
        return d

    def _h2d(self, host):
        """The batch's arrays -> device tensors + the 'uploaded' event its consumers (forward, presample) wait for.  The copies
        run on their own stream: queued behind the running step's kernels on the compute stream, a copy from pageable host
        memory would block the host until that step has finished (one implicit synchronisation per step).  With the pinned
        ring (default) the arrays are packed into ONE page-locked arena and cross in ONE copy into one device block the
        returned tensors are views of (13 copies, allocations and stream records fewer per step on the host)."""
        dev = self.rt.device
        main, up, ring = torch.cuda.current_stream(), self.rt.upload_stream, self.rt.pinned
        arrs = {k: np.ascontiguousarray(a) for k, a in host.items()}
        out = {}
        if ring is not None:
            offs, total = pack_offsets(arrs)
            ring.begin(total)
            arena = ring.bufs[ring.k]
            bytes_view = arena.numpy()
            for k, a in arrs.items():
                if a.nbytes:
                    bytes_view[offs[k]:offs[k] + a.nbytes] = a.reshape(-1).view(np.uint8)
            with torch.cuda.stream(up if up is not None else main):
                block = torch.empty(max(total, 256), dtype=torch.uint8, device=dev)
                block[:total].copy_(arena[:total], non_blocking=True)
            if up is not None:
                block.record_stream(main)         # allocated on the upload stream, consumed on the compute stream
            for k, a in arrs.items():
                out[k] = block[offs[k]:offs[k] + a.nbytes].view(_TORCH_DTYPE[a.dtype.str]).view(a.shape)
            # the packed device block and its layout: a batch of the same shape can be copied into a persistent input slot with ONE
            # device-to-device copy (GraphedTrainStep)
            out['_block'], out['_total'] = block, total
            out['_layout'] = tuple((k, offs[k], a.dtype.str, a.shape) for k, a in arrs.items())
        else:
            for k, a in arrs.items():
                x = torch.from_numpy(a)
                if up is None:
                    out[k] = x.to(dev, non_blocking=True)
                    continue
                with torch.cuda.stream(up):
                    x = x.to(dev, non_blocking=True)
                x.record_stream(main)
                out[k] = x
        out['uploaded'] = torch.cuda.Event()
        out['uploaded'].record(up if up is not None else main)
        if ring is not None:
            ring.end(up if up is not None else main)
        return out

    def _neg_sample(self, pl, d, step, k, stream, which=None):
        """K0 negative sampling (nar_model.py:265-276) of batch d with key `step` into the plan's sampler-output set k.  which = 0 / 1: the key
        is the .step / .step_next field of the runtime's step-scalar record (device) instead of the argument."""
        rt, st, o = self.rt, self._dev_state, pl._samp[k]
        if which is not None:
            check(rt.lib.cham_neg_sample_dev(ptr(d['aci']), d['Bg'], d['T'] + 1, ptr(st['buffer']), st['buffer'].numel(),
                                             rt.tf_random_seed, ptr(rt.scalars), which, d['row_begin'], d['B'], self.negative_samples,
                                             self.negative_sample_from_buffer, ptr(o['neg_ids']), ptr(o['neg_slot']), ptr(o['pool']),
                                             ptr(o['canon']), ptr(o['meta']), ptr(pl.sampler_ws), pl.ws_bytes, stream), "cham_neg_sample_dev")
            return
        check(rt.lib.cham_neg_sample(ptr(d['aci']), d['Bg'], d['T'] + 1, ptr(st['buffer']), st['buffer'].numel(),
                                     rt.tf_random_seed, step, d['row_begin'], d['B'], self.negative_samples,
                                     self.negative_sample_from_buffer, ptr(o['neg_ids']), ptr(o['neg_slot']), ptr(o['pool']),
                                     ptr(o['canon']), ptr(o['meta']), ptr(pl.sampler_ws), pl.ws_bytes, stream), "cham_neg_sample")

    def presample(self, d):
        """Draw the negatives of the NEXT training step now.  Sampling depends on the recent-clicks state and the batch's ids,
        not on the weights, so it can run on the device state's stream right behind the state update of the current batch,
        while the current step's backward is still executing (0.3 ms of latency-bound kernels off the head of every step).
        ``d`` = the uploaded next batch; forward() picks the result up when the batch and the sampler key match."""
        rt, state = self.rt, self.articles_recent_pop_norm
        if not (rt.presample and self.is_training and getattr(state, 'is_device', False) and state.stream is not None
                and self._dev_state is not None and self._dev_state.get('device')):
            return False
        pl = rt.plan(d['B'], d['T'], self.negative_samples, self.negative_sample_from_buffer, d['Bg'])
        k, step = 1 - pl._samp_cur, rt.global_step
        if not rt.capturing:      # (events recorded outside a capture are not waited for inside one: the replay is stream-ordered behind them)
            state.stream.wait_event(pl.created)
            if d.get('uploaded') is not None:
                state.stream.wait_event(d['uploaded'])
            d['aci'].record_stream(state.stream)
        with torch.cuda.stream(state.stream):      # in order behind the state update it must see
            # (captured step: the key is the record's .step_next - this call runs behind the step whose key is .step)
            self._neg_sample(pl, d, step, k, state.stream.cuda_stream, which=1 if rt.capturing else None)
            ev = torch.cuda.Event()
            ev.record()
        d['_presampled'] = (pl, k, step, ev)
        return True

    # ------------------------------------------------------------------ forward
    def forward(self, d, step=None):
        depth = _roctx.depth()
        try:
            return self._forward(d, step)
        finally:
            _roctx.unwind(depth)          # (an error raised mid-step leaves no stage range open - ADVICE r05)

    def _forward(self, d, step):
        rt, lib, L = self.rt, self.rt.lib, self.rt.layout
        st = self._dev_state
        if st is None:
            raise RuntimeError("feed_state() must be called before every step (ItemsStateUpdaterHook.before_run)")
        B, T, N = d['B'], d['T'], self.negative_samples
        pl = rt.plan(B, T, N, self.negative_sample_from_buffer, d['Bg'])
        self._plan, self._d = pl, d
        pl.used_p3 = False
        check(lib.cham_set_log_bases(self.elapsed_days_smooth_log_base, self.popularity_smooth_log_base), "cham_set_log_bases")
        if d.get('uploaded') is not None:
            torch.cuda.current_stream().wait_event(d['uploaded'])
        s = _stream()
        # weight shadows (planes of W2 / W2^T, the row-norm bound of Ws1, bf16 shadows): four small launches that depend on the weights only -
        # on the side lane (idle at the head of a step) behind the previous step's Adam, beside this lane's feature kernels (round 4: the
        # head of a step is ~35 latency-bound launches in a row on one lane)
        shadows_ev = None
        if rt.overlap and rt.shadows_stale():
            rt.side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(rt.side_stream):
                rt.refresh_shadows()
                shadows_ev = torch.cuda.Event()
                shadows_ev.record()
        # BT = rows of the row-wise stages = the P valid positions (all B*T when nothing is padded); BTf = the [B, T] layout
        pos, BT, BTf, NC, pmax = d['pos'], d['P'], pl.BT, pl.NC, pl.pmax
        pl.pos, pl.P = pos, BT
        Rc = BT * NC; Rall = BT + Rc; RV = 2 * BT + pmax + 1
        C, Hp, Fc, Fi = L.C, L.Hp, L.Fc, L.Fi
        cell, NGH = (1 if L.cell == 'gru' else 0), L.NG * L.Hp
        if step is None:
            step = rt.global_step if self.is_training else self.eval_step_key(rt.global_step, self._eval_iter)
        p = rt.p
        # K0 negative sampling (nar_model.py:265-276) - unless presample() already drew this batch's negatives for this key
        _roctx.push("K0 negative sampling")
        ps = d.pop('_presampled', None)
        dsc = rt.dev_scalars          # per-step launch scalars from the device record (NARRuntime.__init__)
        if dsc:
            rt.set_step_scalars(step, d['max_ts'], d['sum_mask'], fields=3, stream=s)
        if ps is not None and ps[0] is pl and ps[2] == step and st.get('device'):
            pl.use_sampler_set(ps[1])
            if ps[3] is not None:
                torch.cuda.current_stream().wait_event(ps[3])
        else:
            self._neg_sample(pl, d, step, pl._samp_cur, s, which=0 if dsc else None)
        if rt.dp_mode in ('sparse', 'sparse_rs') and getattr(rt, 'dp_active', rt.dp_world > 1):
            # item rows this step can touch on ANY rank (parallel.py, mode "sparse"): GLOBAL clicked ids + candidate pool + pad item,
            # as int32 row indices in a buffer of the plan (nothing is allocated or freed around the collective)
            n1 = d['aci'].numel()
            if getattr(pl, 'touched', None) is None or pl.touched.numel() < n1 + pmax + 1:
                pl.touched = torch.zeros(n1 + pmax + 1, dtype=torch.int32, device=rt.device)
            pl.touched[:n1].copy_(d['aci'].reshape(-1))
            pl.touched[n1:n1 + pmax].copy_(pl.pool)
            pl.touched[n1 + pmax:n1 + pmax + 1].zero_()
            rt.dp_touched = pl.touched[:n1 + pmax + 1]
        neg_ids, neg_slot = pl.neg_ids, pl.neg_slot
        if pos is not None:
            neg_ids, neg_slot = pl.neg_ids_c, pl.neg_slot_c
            check(lib.cham_rows_gather(ptr(pl.neg_ids), ptr(pos), BT, 2 * N, ptr(neg_ids), s), "cham_rows_gather")
            check(lib.cham_rows_gather(ptr(pl.neg_slot), ptr(pos), BT, N, ptr(neg_slot), s), "cham_rows_gather")
        pl.cur_neg_ids, pl.cur_neg_slot = neg_ids, neg_slot
        # a training-mode forward on this plan that was NOT followed by backward() (a loss-only call, an exception in between) leaves the
        # side lane's row grouping un-awaited: it reads ids_all and writes perm / seg - wait for it before rewriting ids_all
        if getattr(pl, 'grouped_ev', None) is not None:
            torch.cuda.current_stream().wait_event(pl.grouped_ev)
        # K1 item row set = [clicked ; positives ; pool slots ; pad item 0]
        _roctx.pop(); _roctx.push("K1 features (gather, normalise, scale / center)")
        # (one launch - round 5: eight copy / fill launches before - also seq_len and the position mask of the stages below)
        if dsc:
            check(lib.cham_step_ints_dev(ptr(d['ic_rows']), ptr(d['ln_rows']), ptr(pl.pool), ptr(d['ets_rows']), ptr(rt.scalars), BT, pmax, ptr(d['seq_len']),
                                         B, ptr(d['mask']), ptr(pl.ids_all), ptr(pl.ref_ts), ptr(pl.seq_len), ptr(pl.mask), s), "cham_step_ints_dev")
        else:
            check(lib.cham_step_ints(ptr(d['ic_rows']), ptr(d['ln_rows']), ptr(pl.pool), ptr(d['ets_rows']), int(d['max_ts']), BT, pmax, ptr(d['seq_len']),
                                     B, ptr(d['mask']), ptr(pl.ids_all), ptr(pl.ref_ts), ptr(pl.seq_len), ptr(pl.mask), s), "cham_step_ints")
        pl.grouped_ev = None
        if self.is_training:      # rows of equal id made contiguous: the embedding-gradient sums of the backward pass (depends on ids only)
            if rt.overlap:        # ~10 launches nobody needs before the end of the backward: on the side lane, behind the id copies above
                e_ids = torch.cuda.Event()
                e_ids.record()
                rt.side_stream.wait_event(e_ids)
                with torch.cuda.stream(rt.side_stream):
                    check(lib.cham_group_rows(ptr(pl.ids_all), RV, rt.item_id_bits, ptr(pl.perm), ptr(pl.seg), ptr(pl.group_ws),
                                              pl.group_ws.numel() * 4, _stream()), "cham_group_rows")
                    pl.grouped_ev = torch.cuda.Event()
                    pl.grouped_ev.record()
            else:
                check(lib.cham_group_rows(ptr(pl.ids_all), RV, rt.item_id_bits, ptr(pl.perm), ptr(pl.seg), ptr(pl.group_ws),
                                          pl.group_ws.numel() * 4, s), "cham_group_rows")
        check(lib.cham_item_dynamic_raw(ptr(pl.ids_all), ptr(pl.ref_ts), RV, ptr(rt.created), ptr(st['pop_norm']),
                                        ptr(pl.rec_raw), ptr(pl.nov_raw), s), "cham_item_dynamic_raw")
        if st['n_last'] > 0 and st.get('device') and dsc:
            check(lib.cham_norm_stats_from_buffer_dev(ptr(st['last']), st['n_last'], ptr(rt.scalars), ptr(rt.created),
                                                      ptr(st['pop_norm']), ptr(pl.stat_scratch), ptr(pl.stats), s),
                  "cham_norm_stats_from_buffer_dev")
        elif st['n_last'] > 0 and st.get('device'):
            check(lib.cham_norm_stats_from_buffer(ptr(st['last']), st['n_last'], d['max_ts'], ptr(rt.created),
                                                  ptr(st['pop_norm']), ptr(pl.stat_scratch), ptr(pl.stats), s),
                  "cham_norm_stats_from_buffer")
        elif st['n_last'] > 0:
            check(lib.cham_norm_stats_from_recent(ptr(st['last']), st['n_last'], d['max_ts'], ptr(rt.created),
                                                  ptr(st['pop_norm']), ptr(pl.stat_scratch), ptr(pl.stats), s),
                  "cham_norm_stats_from_recent")
        else:   # very first batch: population = the call's own non-pad ids (nar_model.py:1078-1084)
            check(lib.cham_row_weights(ptr(pl.ids_all), 2 * BT, ptr(neg_slot), BT * N, pmax, ptr(pl.pool),
                                       ptr(pl.w_rows), pl.w_rows[2 * BT:].data_ptr(), s), "cham_row_weights")
            for g, (a, b) in enumerate([(0, BT), (BT, 2 * BT), (2 * BT, RV)]):
                check(lib.cham_norm_stats_from_rows(pl.rec_raw[a:].data_ptr(), pl.nov_raw[a:].data_ptr(),
                                                    pl.w_rows[a:].data_ptr(), b - a, pl.stats[g].data_ptr(), s),
                      "cham_norm_stats_from_rows")
        check(lib.cham_ctx_assemble(ptr(d['cat']), ptr(d['num']), BT, ptr(rt.ctx_desc), Fc, ptr(rt.flat), ptr(p('gamma_ctx')),
                                    ptr(p('beta_ctx')), ptr(pl.Xc_raw), ptr(pl.Xc_s), s), "cham_ctx_assemble")
        if rt.item_lds:
            check(lib.cham_item_assemble_lds(ptr(pl.ids_all), RV, BT, 2 * BT, ptr(rt.meta_cat), rt.n_items, ptr(rt.ace), L.D,
                                             ptr(pl.rec_raw), ptr(pl.nov_raw), ptr(pl.stats), ptr(rt.item_desc), Fi, ptr(rt.item_segs),
                                             rt.n_item_segs, ptr(rt.item_singles) if rt.n_item_singles else None, rt.n_item_singles,
                                             ptr(rt.flat), ptr(p('gamma_item')), ptr(p('beta_item')), ptr(pl.Xi_raw), ptr(pl.Xi_s), s),
                  "cham_item_assemble_lds")
        else:
            check(lib.cham_item_assemble(ptr(pl.ids_all), RV, BT, 2 * BT, ptr(rt.meta_cat), rt.n_items, ptr(rt.ace), L.D,
                                         ptr(pl.rec_raw), ptr(pl.nov_raw), ptr(pl.stats), ptr(rt.item_desc), Fi, ptr(rt.flat),
                                         ptr(p('gamma_item')), ptr(p('beta_item')), ptr(pl.Xi_raw), ptr(pl.Xi_s), s),
                  "cham_item_assemble")
        if shadows_ev is not None:
            torch.cuda.current_stream().wait_event(shadows_ev)
        else:
            rt.refresh_shadows()
        _roctx.pop(); _roctx.push("K2 PreCAR + CAR, K3 recurrent branch (side lane)")
        drop = self.is_training and self.keep_prob < 1.0
        # (full batches only: with ragged sessions the side lane's recurrent chain IS the critical path and the two launches are better
        # off on the main lane - G1-like lengths 114-115 k vs 111-112 k sessions/s, profiles/r05_notes.md)
        head_split = rt.head_split and rt.overlap and BT * NC > rt.w2_main_rows
        if drop:
            # dropout_keep_prob < 1 (nar_model.py:338, 352, 368): the per-element masks make every CAR row occurrence-specific, so
            # the PreCAR layer runs on the dense [clicked | candidates] x [ctx | item] rows instead of the factorised U + V form
            pl.dropout_buffers(rt)
            keep, Fw, e1, e2 = float(self.keep_prob), Fc + Fi, L.entries['W1c'], L.entries['W1i']
            assert e2.offset == e1.offset + e1.size
            W1 = rt.flat[e1.offset:e2.offset + e2.size].view(Fw, C)
            gW1 = rt.grads[e1.offset:e2.offset + e2.size].view(Fw, C)

            def dropout(x, y, rows, cols, ld, site_first, site_rest, group, posmap, col_split=None, col_shift=0):
                check(lib.cham_dropout(ptr(x), ptr(y), rows, cols, ld, keep, rt.tf_random_seed, step & 0xFFFFFFFF, site_first, site_rest,
                                       group, ptr(posmap), T, d['row_begin'], cols if col_split is None else col_split, col_shift,
                                       _stream()), "cham_dropout")
            self._drop = dict(fn=dropout, W1=W1, gW1=gW1, Fw=Fw)
            check(lib.cham_dense_rows(ptr(pl.Xc_s), Fc, ptr(pl.Xi_s), Fi, BT, N, pmax, ptr(neg_slot), ptr(pl.Xd), s), "cham_dense_rows")
            dropout(pl.Xd, pl.Xd, BT, Fw, Fw, 16, 16, 1, pos, Fc, Fc - L.f_ctx)
            dropout(pl.Xd[BT:], pl.Xd[BT:], Rc, Fw, Fw, 17, 18, NC, pos, Fc, Fc - L.f_ctx)
            rt.gemm(pl.Xd, W1, pl.Z1, BT, C, Fw, Fw, C, C, bias=p('b1'), act=ACT_LEAKY)      # clicked-input rows first ...
        else:
            self._drop = None
            # factorised PreCAR: U (per click) + V (per unique item row), then CAR
            rt.gemm(pl.Xc_s, p('W1c'), pl.U, BT, C, Fc, Fc, C, C, bias=p('b1'))
            rt.gemm(pl.Xi_s, p('W1i'), pl.V, RV, C, Fi, Fi, C, C)
            # PreCAR combine + CAR layer 2 on the clicked-input rows first: they feed the recurrent branch ...
            if not head_split:
                check(lib.cham_combine_fwd(ptr(pl.U), ptr(pl.V), C, BT, N, pmax, ptr(neg_slot), ptr(pl.Z1), 0, BT, s), "cham_combine_fwd")
        if not head_split:
            rt.gemm(pl.Z1, p('W2'), pl.Z2, BT, C, C, C, C, C, bias=p('b2'), act=ACT_TANH)
        rt.fork()
        with rt.side():   # ... which is latency-bound (one workgroup per 32 sessions) and overlaps with ...
            if head_split:      # (round 5) ... at the head of THIS lane: only the recurrent branch reads the clicked rows' CAR output
                if not drop:
                    check(lib.cham_combine_fwd(ptr(pl.U), ptr(pl.V), C, BT, N, pmax, ptr(neg_slot), ptr(pl.Z1), 0, BT, _stream()), "cham_combine_fwd")
                rt.gemm(pl.Z1, p('W2'), pl.Z2, BT, C, C, C, C, C, bias=p('b2'), act=ACT_TANH)
            x, ldx, K = pl.Z2, C, C
            if pos is not None:          # clicked rows back into the [B, T] layout of the recurrent stack (zeros at padded steps)
                pl.Z2f.zero_()
                check(lib.cham_rows_scatter(ptr(pl.Z2), ptr(pos), BT, C, ptr(pl.Z2f), _stream()), "cham_rows_scatter")
                x = pl.Z2f
            for l in range(L.L):
                rt.gemm(x, p('rnn%d/Wx' % l), pl.xproj[l], BTf, NGH, K, ldx, NGH, NGH, bias=p('rnn%d/b' % l))
                if L.rnn_stepwise:      # large hidden size: one GEMM (h W_h) + one gate kernel per time step
                    pl.h_state.zero_()
                    for t in range(T):
                        rt.gemm(pl.h_state, p('rnn%d/Wh' % l), pl.zh, B, 2 * Hp, Hp, Hp, 2 * Hp, 2 * Hp, force_f32=True)
                        check(lib.cham_ugrnn_point_fwd(ptr(pl.xproj[l]), ptr(pl.zh), ptr(pl.seq_len), B, T, t, Hp, ptr(pl.h_state),
                                                       ptr(pl.rnn_out[l]), ptr(pl.hprev[l]), ptr(pl.G[l]), ptr(pl.Cc[l]), _stream()),
                              "cham_ugrnn_point_fwd")
                elif 0 < Rc <= rt.rnn_coop_rows and B <= 1024:
                    ws = rt.rnn_coop_ws(B)
                    check(lib.cham_ugrnn_fwd_coop(ptr(pl.xproj[l]), ptr(p('rnn%d/Wh' % l)), ptr(pl.seq_len), B, T, Hp, ptr(pl.rnn_out[l]),
                                                  ptr(pl.hprev[l]), ptr(pl.G[l]), ptr(pl.Cc[l]), ptr(ws), ws.numel(), _stream()), "cham_ugrnn_fwd_coop")
                else:
                    check(lib.cham_rnn_fwd(cell, ptr(pl.xproj[l]), ptr(p('rnn%d/Wh' % l)), ptr(pl.seq_len), B, T, Hp,
                                           ptr(pl.rnn_out[l]), ptr(pl.hprev[l]), ptr(pl.G[l]), ptr(pl.Cc[l]), ptr(pl.R[l]),
                                           ptr(pl.RH[l]), _stream()), "cham_rnn_fwd")
                x, ldx, K = pl.rnn_out[l], Hp, Hp
                if drop:     # DropoutWrapper(output_keep_prob), nar_model.py:1331: the layer's OUTPUT is dropped, its state is not
                    dropout(pl.rnn_out[l], pl.rnn_drop[l], BTf, Hp, Hp, 20 + l, 20 + l, 1, None)
                    x = pl.rnn_drop[l]
            if pos is not None:
                check(lib.cham_rows_gather(ptr(x), ptr(pos), BT, Hp, ptr(pl.rnn_c), _stream()), "cham_rows_gather")
                x = pl.rnn_c
            rt.gemm(x, p('Wf1'), pl.FC1, BT, 512, Hp, Hp, 512, 512, bias=p('bf1'), act=ACT_LEAKY)
            fc1 = pl.FC1
            if drop:         # nar_model.py:418
                dropout(pl.FC1, pl.FC1d, BT, 512, 512, 19, 19, 1, pos)
                fc1 = pl.FC1d
            rt.gemm(fc1, p('Wf2'), pl.pred, BT, C, 512, 512, C, C, bias=p('bf2'), act=ACT_TANH)
        # ... the candidate rows: PreCAR combine (HBM-bound) + the dominant GEMM, CAR layer 2 on the B*T*(1+N) rows
        if rt.b16:
            # bf16 configuration: candidate-row matrices are bf16 in HBM, weights through their bf16 shadows (csrc/gemm_b16.hip)
            sh = rt.shadow
            if drop:      # dense PreCAR rows (masks differ per occurrence): bf16-rounded operands, fp32 out, stored as the bf16-resident Z1c
                rt.gemm(pl.Xd[BT:], self._drop['W1'], pl.Z1f[BT:], Rc, C, Fc + Fi, Fc + Fi, C, C, bias=p('b1'), act=ACT_LEAKY)
                check(lib.cham_cast_b16(pl.Z1f[BT:].data_ptr(), Rc, C, ptr(pl.Z1c), None, s), "cham_cast_b16")
            else:
                check(lib.cham_combine_fwd_b16(ptr(pl.U), ptr(pl.V), C, BT, N, pmax, ptr(neg_slot), ptr(pl.Z1c), s), "cham_combine_fwd_b16")
            rt.gemm_b16(pl.Z1c, C, 0, sh['W2T'], C, 1, pl.Z2c, C, 0, Rc, C, C, bias=p('b2'), act=ACT_TANH, dma=rt.b16_dma)
            rt.join()
            check(lib.cham_mul_rows_b16(ptr(pl.Z2c), ptr(pl.pred), C, BT, NC, ptr(pl.Mc), s), "cham_mul_rows_b16")
            rt.gemm_b16(pl.Mc, C, 0, sh['Ws1T'], C, 1, pl.S1, 128, 0, Rc, 128, C, bias=p('bs1'), act=ACT_LEAKY)
            rt.gemm_b16(pl.S1, 128, 0, sh['Ws2T'], 128, 1, pl.S2, 64, 0, Rc, 64, 128, bias=p('bs2'), act=ACT_LEAKY)
            rt.gemm_b16(pl.S2, 64, 0, sh['Ws3T'], 64, 1, pl.S3, 32, 0, Rc, 32, 64, bias=p('bs3'), act=ACT_LEAKY)
            softmax_fwd = lib.cham_score_softmax_fwd_b16
        else:
            use_p3 = pl.used_p3 = rt.p3 and not drop and Rc > 0
            if drop:
                rt.gemm(pl.Xd[BT:], self._drop['W1'], pl.Z1[BT:], Rc, C, Fc + Fi, Fc + Fi, C, C, bias=p('b1'), act=ACT_LEAKY)
            elif use_p3 and rt.h2:      # candidate rows straight into two fp16 planes x 2^k, k from the bound max|U| + max|V| (no fp32 copy)
                check(lib.cham_h2_scale_absmax(ptr(pl.U), BT * C, ptr(pl.V), RV * C, ptr(pl.sc_z1), s), "cham_h2_scale_absmax")
                check(lib.cham_combine_fwd_h2b(ptr(pl.U), ptr(pl.V), C, BT, N, pmax, ptr(neg_slot), ptr(pl.Z1p), pl.z1_ps, ptr(pl.sc_z1), 1 if pl.z1_tiles else 0, s),
                      "cham_combine_fwd_h2")
            elif use_p3:      # candidate rows straight into three bf16 planes (no fp32 copy)
                check(lib.cham_combine_fwd_p3(ptr(pl.U), ptr(pl.V), C, BT, N, pmax, ptr(neg_slot), ptr(pl.Z1p), pl.p3_ps, s), "cham_combine_fwd_p3")
            else:
                check(lib.cham_combine_fwd(ptr(pl.U), ptr(pl.V), C, BT, N, pmax, ptr(neg_slot), ptr(pl.Z1), BT, Rc, s), "cham_combine_fwd")
            if use_p3 and rt.h2:
                rt.gemm_h2(pl.Z1p, pl.z1_ps, C, pl.sc_z1, rt.w2tp, C * C, C, rt.sc_w2, 0, pl.Z2[BT:], C, Rc, C, C, bias=p('b2'), act=ACT_TANH, a_tiles=pl.z1_tiles)
            elif use_p3:
                rt.gemm_p3(pl.Z1p, pl.p3_ps, C, rt.w2tp, C * C, C, 0, pl.Z2[BT:], C, Rc, C, C, bias=p('b2'), act=ACT_TANH)
            else:
                rt.gemm(pl.Z1[BT:], p('W2'), pl.Z2[BT:], Rc, C, C, C, C, C, bias=p('b2'), act=ACT_TANH)
            rt.join()
            # scorer: (cand (.) pred) -> 128 -> 64 -> 32 -> 1, softmax(/tau), masked NLL
            _roctx.pop(); _roctx.push("K5 scorer, softmax, loss")
            Z2c = pl.Z2[BT:Rall]
            rt.gemm(Z2c, p('Ws1'), pl.S1, Rc, 128, C, C, 128, 128, bias=p('bs1'), act=ACT_LEAKY, rowscale=pl.pred, ldrs=C, rs_div=NC,
                    h2scales=(rt.sc_unit, rt.sc_ws1n) if rt.s1_h2_fwd else None)
            rt.gemm(pl.S1, p('Ws2'), pl.S2, Rc, 64, 128, 128, 64, 64, bias=p('bs2'), act=ACT_LEAKY)
            rt.gemm(pl.S2, p('Ws3'), pl.S3, Rc, 32, 64, 64, 32, 32, bias=p('bs3'), act=ACT_LEAKY)
            softmax_fwd = lib.cham_score_softmax_fwd
        check(softmax_fwd(ptr(pl.S3), 32, ptr(p('Ws4')), ptr(p('bs4')), BT, N, float(self.softmax_temperature),
                          ptr(pl.mask), ptr(pl.logits), ptr(pl.probs), ptr(pl.nll), self.novelty_reg_factor,
                          ptr(neg_ids), ptr(st['pop_norm']), ptr(pl.nov_aux), s), "cham_score_softmax_fwd")
        def loss_kernels(s):
            check(lib.cham_sumsq_partial(ptr(rt.flat), L.n_reg, ptr(rt.sumsq), s), "cham_sumsq_partial")
            if dsc:
                check(lib.cham_loss_finalize_dev(ptr(pl.nll), BT, ptr(rt.scalars), ptr(rt.sumsq), float(self.reg_weight_decay), ptr(pl.loss), s),
                      "cham_loss_finalize_dev")
            else:
                check(lib.cham_loss_finalize(ptr(pl.nll), BT, d['sum_mask'], ptr(rt.sumsq), float(self.reg_weight_decay), ptr(pl.loss), s),
                      "cham_loss_finalize")
        if getattr(self, '_defer_loss', False) and rt.overlap and rt.loss_side:
            # inside a training step (train_step: forward -> backward -> Adam) the L2 term and the loss record leave the main lane - nothing in
            # the backward pass reads them, they sat between the softmax and its backward (round 6); the backward's final join covers them
            e_nll = torch.cuda.Event(); e_nll.record()
            rt.side_stream.wait_event(e_nll)
            with torch.cuda.stream(rt.side_stream):
                loss_kernels(_stream())
        else:
            loss_kernels(s)
        _roctx.pop()
        self.total_loss = pl.loss            # device [total, xe, reg]; xe is this rank's share under data parallel
        if not self.is_training and st.get('device'):
            self.articles_recent_pop_norm.note_consumed(d['aci'])      # last read of the state in an EVAL step
        return pl

    # ------------------------------------------------------------------ backward (hand-derived; nar_model.py:718)
    def backward(self):
        """Hand-derived backward in two stream lanes.  MAIN carries the critical dgrad chain (softmax -> scorer -> CAR layer 2 ->
        PreCAR combine -> features); SIDE carries everything that only produces weight gradients - wgrad GEMMs,
        bias column sums - plus the session-FC / recurrent chain and the clicked-row CAR dgrad, so the HBM-bound elementwise
        kernels of one lane run beside the MFMA-bound GEMMs of the other (DESIGN.md "Step schedule").  Every cross-lane
        dependency is an explicit event; with overlap off the same program order runs on one stream."""
        rt, lib, L = self.rt, self.rt.lib, self.rt.layout
        pl, d = self._plan, self._d
        # the log bases are host-side values of the library read when a kernel is launched: set again here, another model of this process
        # (different bases) may have run its forward pass since ours
        check(lib.cham_set_log_bases(self.elapsed_days_smooth_log_base, self.popularity_smooth_log_base), "cham_set_log_bases")
        s = _stream()
        B, T, N = pl.B, pl.T, pl.N
        pos, BT, BTf, NC, pmax = pl.pos, pl.P, pl.BT, pl.NC, pl.pmax          # see forward(): BT = valid positions
        Rc = BT * NC; Rall = BT + Rc; RV = 2 * BT + pmax + 1
        neg_ids, neg_slot = pl.cur_neg_ids, pl.cur_neg_slot
        C, Hp, Fc, Fi = L.C, L.Hp, L.Fc, L.Fi
        cell, NGH = (1 if L.cell == 'gru' else 0), L.NG * L.Hp
        p, g = rt.p, rt.g
        main_stream, on = torch.cuda.current_stream(), rt.overlap
        b16 = rt.b16
        sh = rt.shadow
        drop = self._drop
        dropout = drop['fn'] if drop else None
        use_p3 = getattr(pl, 'used_p3', False)

        def mark():                      # event on the current stream
            if not on:
                return None
            ev = torch.cuda.Event(); ev.record()
            return ev

        import contextlib

        @contextlib.contextmanager
        def side(*events):               # run the block on the side lane after `events`
            if not on:
                yield
                return
            for ev in events:
                rt.side_stream.wait_event(ev)
            with torch.cuda.stream(rt.side_stream):
                yield

        def main_wait(ev):
            if on:
                main_stream.wait_event(ev)

        @contextlib.contextmanager
        def aux(*events):                # run the block on the third lane after `events`
            for ev in events:
                rt.aux_stream.wait_event(ev)
            with torch.cuda.stream(rt.aux_stream):
                yield
        e_auxdone = None

        e_start = mark()                 # side lane must not run ahead of the previous step's tail
        if on and rt.loss_side:          # (round 6) the zero fill of the embedding gradients on the side lane: their writers - the last kernels
            with side(e_start):          # of the main / third lane - come behind e_dZ1in, an event of this lane
                rt.grads[:L.emb_end].zero_()
        else:
            rt.grads[:L.emb_end].zero_()
            e_start = mark()
        if rt.dev_scalars:      # sum(mask) from the step-scalar record (written by this step's forward)
            check((lib.cham_score_softmax_bwd_b16_dev if b16 else lib.cham_score_softmax_bwd_dev)(
                ptr(pl.S3), 32, ptr(p('Ws4')), ptr(pl.probs), ptr(pl.mask), BT, N, float(self.softmax_temperature), ptr(rt.scalars),
                ptr(pl.ds), ptr(pl.dS3), self.novelty_reg_factor, ptr(neg_ids), ptr(self._dev_state['pop_norm']), ptr(pl.logits),
                ptr(pl.nov_aux), s), "cham_score_softmax_bwd_dev")
        else:
            check((lib.cham_score_softmax_bwd_b16 if b16 else lib.cham_score_softmax_bwd)(
                ptr(pl.S3), 32, ptr(p('Ws4')), ptr(pl.probs), ptr(pl.mask), BT, N, float(self.softmax_temperature), d['sum_mask'],
                ptr(pl.ds), ptr(pl.dS3), self.novelty_reg_factor, ptr(neg_ids), ptr(self._dev_state['pop_norm']), ptr(pl.logits),
                ptr(pl.nov_aux), s), "cham_score_softmax_bwd")
        if self._dev_state.get('device'):
            self.articles_recent_pop_norm.note_consumed(d['aci'])      # last read of the state in a TRAIN step
        # scorer dgrad chain on this lane (three short GEMMs); the side lane takes the layer-1 weight gradient FIRST - 65 GFLOP of
        # matrix work that then runs beside the HBM-bound k_mulpred_bwd instead of beside the MFMA-bound CAR dgrad - and the small
        # (HBM-bound, split-K) weight / bias gradients of layers 2-4 after it
        if b16:
            rt.gemm_b16(pl.dS3, 32, 0, sh['Ws3'], 32, 1, pl.dS2, 64, 0, Rc, 64, 32, dref=pl.S2, ldr=64, dact=ACT_LEAKY)
            rt.gemm_b16(pl.dS2, 64, 0, sh['Ws2'], 64, 1, pl.dS1, 128, 0, Rc, 128, 64, dref=pl.S1, ldr=128, dact=ACT_LEAKY)
            Z2c, dZ2c = pl.Z2c[:Rc], pl.dZ2c[:Rc]
        else:
            rt.gemm(pl.dS3, p('Ws3'), pl.dS2, Rc, 64, 32, 32, 32, 64, transB=1, dref=pl.S2, ldr=64, dact=ACT_LEAKY)
            rt.gemm(pl.dS2, p('Ws2'), pl.dS1, Rc, 128, 64, 64, 64, 128, transB=1, dref=pl.S1, ldr=128, dact=ACT_LEAKY)
            if not (use_p3 and rt.dm_fused and 32 <= NC <= 256):
                pl.ensure_rows('dZ2')          # the scorer layer-1 dgrad goes through HBM
            Z2c, dZ2c = pl.Z2[BT:Rall], pl.dZ2[BT:Rall]
        h2 = use_p3 and rt.h2
        if h2:      # scale of the gradient at the CAR tanh from the Cauchy-Schwarz bound of dS1 Ws1^T: max row norm of dS1 x max row norm of Ws1
            check(lib.cham_h2_scale_rownorm2(ptr(pl.dS1), Rc, pl.dS1.shape[1], pl.dS1.shape[1], rt.sc_ws1n.data_ptr() + 8, ptr(pl.sc_dz2),
                                             ptr(pl.sc_ds1), s), "cham_h2_scale_rownorm2")
        e_dS1 = mark()     # (starting the side lane only after the next GEMM, to pair its MFMA work with k_mulpred_bwd's HBM work,
        #                      measured 0.26 ms slower: 17.28-17.33 vs 17.01-17.07 ms, A/B in one gpurun call)
        # scorer layer-1 dgrad + cand (.) pred backward in one kernel (csrc/dm_fused.hip): dM never reaches HBM
        dm_fused = use_p3 and rt.dm_fused and 32 <= NC <= 256
        dm_fused_b16 = b16 and rt.dm_fused_b16 and 32 <= NC <= 256
        if dm_fused_b16:
            pass
        elif b16:
            rt.gemm_b16(pl.dS1, 128, 0, sh['Ws1'], 128, 1, dZ2c, C, 0, Rc, C, 128)
        elif not dm_fused:
            rt.gemm(pl.dS1, p('Ws1'), dZ2c, Rc, C, 128, 128, 128, C, transB=1)
        def scorer_small_wgrads():       # layers 2-4: short split-K GEMMs + column sums, nothing but Adam (and the early DP bucket) waits for them
            if b16:
                rt.gemm_b16(pl.S1, 128, 1, pl.dS2, 64, 0, g('Ws2'), 64, 1, 128, 64, Rc, splits=0)
                rt.colsum(pl.dS2, 64, Rc, 64, g('bs2'), b16=True)
                rt.gemm_b16(pl.S2, 64, 1, pl.dS3, 32, 0, g('Ws3'), 32, 1, 64, 32, Rc, splits=0)
                rt.colsum(pl.dS3, 32, Rc, 32, g('bs3'), b16=True)
                rt.colsum(pl.S3, 32, Rc, 32, g('Ws4'), w=pl.ds, b16=True)
            else:
                rt.gemm(pl.S1, pl.dS2, g('Ws2'), 128, 64, Rc, 128, 64, 64, transA=1, splits=0)
                rt.colsum(pl.dS2, 64, Rc, 64, g('bs2'))
                rt.gemm(pl.S2, pl.dS3, g('Ws3'), 64, 32, Rc, 64, 32, 32, transA=1, splits=0)
                rt.colsum(pl.dS3, 32, Rc, 32, g('bs3'))
                rt.colsum(pl.S3, 32, Rc, 32, g('Ws4'), w=pl.ds)
            rt.colsum(pl.ds, 1, Rc, 1, g('bs4'))
        # Side lane, "critical chain first": with short sessions the CAR GEMMs shrink and the
        # main lane ends up waiting for the clicked-row gradient that comes out of the side lane's FC -> recurrent -> CAR chain (0.4 ms
        # of a 3.4 ms step in the kernel trace of the G1-like bench leg).  So that chain is enqueued ahead of every weight / bias
        # gradient it does not need; those follow behind it.
        # (bf16 configuration with full-length sessions: 6.26 vs 6.17 ms - the deferred gradients then land beside the W2 wgrad - so there it
        # is used for compacted, i.e. ragged, batches only: 2.23 vs 2.35 ms; fp32: 13.0 = 13.0 ms full-length, 2.95 vs 3.11 ms G1-like lengths)
        crit_first = on and (not b16 or pos is not None)
        w2_main = bool(on and use_p3 and 0 < Rc <= rt.w2_main_rows)
        # third lane for the small weight gradients (full batches of the default / six-plane fp32 arithmetic, one recurrent layer of the
        # UGRNN kind: with more layers the per-layer gradients read a buffer the next layer's backward overwrites)
        use_aux = bool(on and rt.wgrad_aux and use_p3 and not w2_main and not b16 and L.L == 1 and cell == 0 and not L.rnn_stepwise and not drop)
        def ws1_wgrad():
            if b16:      # weight gradients: activations^T x gradients, both bf16 [rows, *] (TN through the LDS transpose read)
                rt.gemm_b16(pl.Mc, C, 1, pl.dS1, 128, 0, g('Ws1'), 128, 1, C, 128, Rc, splits=0)
                rt.colsum(pl.dS1, 128, Rc, 128, g('bs1'), b16=True)
            else:
                rt.gemm(Z2c, pl.dS1, g('Ws1'), C, 128, Rc, C, 128, 128, transA=1, rowscale=pl.pred, ldrs=C, rs_div=NC, splits=0,
                        h2scales=(rt.sc_unit, pl.sc_ds1) if (rt.s1_h2 and use_p3) else None)
                rt.colsum(pl.dS1, 128, Rc, 128, g('bs1'))
        with side(e_start, e_dS1):
            ws1_wgrad()
            if not crit_first:
                scorer_small_wgrads()
        if dm_fused:
            prof = rt.profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if h2 and pl.dz2_tiles:      # ... with the planes of dZ2 written tile-blocked (ds1 / w scale records: the kernel's own products on two fp16 planes too)
                f16 = rt.dm_f16
                check(lib.cham_dm_mulpred_h2_blk(ptr(pl.dS1), 128, 128, ptr(rt.ws1h if f16 else rt.ws1p), C * 128, ptr(pl.sc_ds1) if f16 else None,
                                                 ptr(rt.sc_ws1n) if f16 else None, ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(pl.dZ2p), pl.dz2_ps, ptr(pl.sc_dz2),
                                                 ptr(pl.dpred), ptr(pl.b2part), s), "cham_dm_mulpred_h2_blk")
            elif h2:
                if rt.dm_f16:     # the kernel's own products on two fp16 planes too
                    check(lib.cham_dm_mulpred_h2h(ptr(pl.dS1), 128, 128, ptr(rt.ws1h), C * 128, ptr(pl.sc_ds1), ptr(rt.sc_ws1n), ptr(Z2c), ptr(pl.pred),
                                                  C, BT, N, ptr(pl.dZ2p), pl.dz2_ps, ptr(pl.sc_dz2), ptr(pl.dpred), ptr(pl.b2part), s), "cham_dm_mulpred_h2h")
                else:
                    check(lib.cham_dm_mulpred_h2(ptr(pl.dS1), 128, 128, ptr(rt.ws1p), C * 128, ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(pl.dZ2p), pl.dz2_ps,
                                                 ptr(pl.sc_dz2), ptr(pl.dpred), ptr(pl.b2part), s), "cham_dm_mulpred_h2")
            else:
                check(lib.cham_dm_mulpred_p3(ptr(pl.dS1), 128, 128, ptr(rt.ws1p), C * 128, ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(pl.dZ2p), pl.p3_ps,
                                             ptr(pl.dpred), ptr(pl.b2part), s), "cham_dm_mulpred_p3")
            if prof is not None:
                e1.record()
                prof.append(dict(M=Rc, N=C, K=128, transA=0, transB=1, splits=1, act=0, dref=False, dact=0, bias=False, rowscale=False, bf16=False,
                                 dmf=True, h2out=bool(h2), f16p=bool(h2 and rt.dm_f16), tile=0, epi=0, ev=(e0, e1)))
        elif dm_fused_b16:      # bf16 configuration: the same fusion over single bf16 matrices (dM rounded to bf16 where the pair stores it)
            prof = rt.profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.cham_dm_mulpred_b16(ptr(pl.dS1), 128, 128, ptr(sh['Ws1']), ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(dZ2c), ptr(pl.dpred),
                                          ptr(pl.b2part), s), "cham_dm_mulpred_b16")
            if prof is not None:
                e1.record()
                prof.append(dict(M=Rc, N=C, K=128, transA=0, transB=1, splits=1, act=0, dref=False, dact=0, bias=False, rowscale=False, bf16=True,
                                 dmf=True, tile=0, epi=0, ev=(e0, e1)))
        elif h2:          # gradient at the CAR tanh straight into two fp16 planes + this position's share of the b2 gradient
            check(lib.cham_mulpred_bwd_h2(ptr(dZ2c), ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(pl.dpred), ptr(pl.dZ2p), pl.dz2_ps, ptr(pl.b2part),
                                          ptr(pl.sc_dz2), s), "cham_mulpred_bwd_h2")
        elif use_p3:      # gradient at the CAR tanh straight into three bf16 planes + this position's share of the b2 gradient
            check(lib.cham_mulpred_bwd_p3(ptr(dZ2c), ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(pl.dpred), ptr(pl.dZ2p), pl.p3_ps, ptr(pl.b2part), s),
                  "cham_mulpred_bwd_p3")
        else:             # k_mulpred_bwd: HBM-bound (3 GB), between two MFMA-bound GEMMs
            check((lib.cham_mulpred_bwd_b16 if b16 else lib.cham_mulpred_bwd)(ptr(dZ2c), ptr(Z2c), ptr(pl.pred), C, BT, N, ptr(pl.dpred), s),
                  "cham_mulpred_bwd")
        e_dZ2c = mark()              # dZ2c final + dpred
        # The candidate-row dgrad of CAR layer 2 on the main lane (needs dZ2c only): the largest GEMM of the backward, ENQUEUED BEFORE the
        # side lane's long launch sequence below - in the bf16 configuration the GPU keeps up with the host, and the ~25 launches of the
        # side block (0.5-0.8 ms of host time) left this lane idle for that long (kernel-trace timeline, profiles/r02_notes.md);
        # NT layout on the 256x256 tile.
        if b16:
            rt.gemm_b16(dZ2c, C, 0, sh['W2'], C, 1, pl.dZ1c, C, 0, Rc, C, C, dref=pl.Z1c, ldr=C, dact=ACT_LEAKY, dma=rt.b16_dma)
        def w2_wgrad_planes(splits):      # candidate rows' share of the W2 weight gradient from the planes (TN, split-K)
            if h2:
                rt.gemm_h2(pl.Z1p, pl.z1_ps, C, pl.sc_z1, pl.dZ2p, pl.dz2_ps, C, pl.sc_dz2, 1, g('W2'), C, C, C, Rc, splits=splits,
                           a_tiles=pl.z1_tiles, b_tiles=pl.dz2_tiles)
            else:
                rt.gemm_p3(pl.Z1p, pl.p3_ps, C, pl.dZ2p, pl.p3_ps, C, 1, g('W2'), C, C, C, Rc, splits=splits)
        if h2:          # planes of dZ2 x planes of W2 as stored; leaky' from the sign of Z1's h plane
            gsum = getattr(pl, 'gsum', None) if not drop else None
            rt.gemm_h2(pl.dZ2p, pl.dz2_ps, C, pl.sc_dz2, rt.w2p, C * C, C, rt.sc_w2, 0, pl.dZ1[BT:], C, Rc, C, C, dref_h=pl.Z1p, ldr=C, dact=ACT_LEAKY,
                       a_tiles=pl.dz2_tiles, dref_blocked=1 if pl.z1_tiles else 0, group_rows=NC, groupsum=gsum)
        elif use_p3:
            rt.gemm_p3(pl.dZ2p, pl.p3_ps, C, rt.w2p, C * C, C, 0, pl.dZ1[BT:], C, Rc, C, C, dref_h=pl.Z1p, ldr=C, dact=ACT_LEAKY)
        elif not b16 and Rc > 0:
            rt.gemm(pl.dZ2[BT:Rall], p('W2'), pl.dZ1[BT:Rall], Rc, C, C, C, C, C, transB=1, dref=pl.Z1[BT:Rall], ldr=C, dact=ACT_LEAKY)
        e_cdgrad = mark()
        e_w2main = None
        if w2_main:      # candidate rows' share of the W2 weight gradient here, in the main lane's wait for the side lane's recurrent chain
            w2_wgrad_planes(0)
            e_w2main = mark()
        # session FCs + recurrent layers (latency-bound: one workgroup per 32 sessions) ...
        last = L.L - 1
        deferred_w2 = None
        with side(e_dZ2c):
            ss = _stream()
            rnn_y = (lambda l: pl.rnn_drop[l]) if drop else (lambda l: pl.rnn_out[l])     # what the next layer / FC1 consumed

            def fc_wgrads():
                rt.gemm(pl.FC1d if drop else pl.FC1, pl.dpred, g('Wf2'), 512, C, BT, 512, C, C, transA=1, splits=0)
                rt.colsum(pl.dpred, C, BT, C, g('bf2'))
                rt.gemm(pl.rnn_c if pos is not None else rnn_y(last), pl.dFC1, g('Wf1'), Hp, 512, BT, Hp, 512, 512, transA=1, splits=0)
                rt.colsum(pl.dFC1, 512, BT, 512, g('bf1'))
                if crit_first:
                    scorer_small_wgrads()
                if rt.dp_early_bucket is not None and not self._accumulating:
                    rt.dp_early_bucket(rt.grads)     # data parallel: [Wf1 .. Ws4] gradients are final - their all-reduce starts now
            rt.gemm(pl.dpred, p('Wf2'), pl.dFC1, BT, 512, C, C, C, 512, transB=1, dref=pl.FC1, ldr=512, dact=ACT_LEAKY)
            if drop:                     # (mask and leaky' are both element-wise factors: the order does not matter)
                dropout(pl.dFC1, pl.dFC1, BT, 512, 512, 19, 19, 1, pos)
            if not crit_first:
                fc_wgrads()
            if pos is not None:          # d rnn_out back into the [B, T] layout (zero at padded steps)
                rt.gemm(pl.dFC1, p('Wf1'), pl.drnn_c, BT, Hp, 512, 512, 512, Hp, transB=1)
                pl.drnn.zero_()
                check(lib.cham_rows_scatter(ptr(pl.drnn_c), ptr(pos), BT, Hp, ptr(pl.drnn), ss), "cham_rows_scatter")
            else:
                rt.gemm(pl.dFC1, p('Wf1'), pl.drnn, BT, Hp, 512, 512, 512, Hp, transB=1)
            if drop:
                dropout(pl.drnn, pl.drnn, BTf, Hp, Hp, 20 + last, 20 + last, 1, None)
            for l in range(last, -1, -1):
                if L.rnn_stepwise:
                    pl.carry.zero_()
                    for t in range(T - 1, -1, -1):
                        check(lib.cham_ugrnn_point_bwd(ptr(pl.drnn), ptr(pl.carry), ptr(pl.seq_len), B, T, t, Hp, ptr(pl.hprev[l]),
                                                       ptr(pl.G[l]), ptr(pl.Cc[l]), ptr(pl.dxproj), ptr(pl.dzs), ptr(pl.direct), ss),
                              "cham_ugrnn_point_bwd")
                        # carry = direct + dzs W_h^T  (rows beyond their length: dzs = 0, direct = carry -> unchanged)
                        pl.carry.copy_(pl.direct)
                        rt.gemm(pl.dzs, p('rnn%d/Wh' % l), pl.carry, B, Hp, 2 * Hp, 2 * Hp, 2 * Hp, Hp, transB=1, accumulate=1, force_f32=True)
                elif 0 < Rc <= rt.rnn_coop_rows and B <= 1024:
                    ws = rt.rnn_coop_ws(B)
                    check(lib.cham_ugrnn_bwd_coop(ptr(pl.drnn), ptr(p('rnn%d/Wh' % l)), ptr(pl.seq_len), B, T, Hp, ptr(pl.hprev[l]), ptr(pl.G[l]),
                                                  ptr(pl.Cc[l]), ptr(pl.dxproj), ptr(ws), ws.numel(), ss), "cham_ugrnn_bwd_coop")
                else:
                    check(lib.cham_transpose_f32(ptr(p('rnn%d/Wh' % l)), Hp, 2 * Hp, ptr(pl.WhT), ss), "cham_transpose_f32")
                    if cell == 1:
                        check(lib.cham_transpose_f32(ptr(p('rnn%d/Wch' % l)), Hp, Hp, pl.WhT[2 * Hp:].data_ptr(), ss), "cham_transpose_f32")
                    check(lib.cham_rnn_bwd(cell, ptr(pl.drnn), ptr(pl.WhT), ptr(pl.seq_len), B, T, Hp, ptr(pl.hprev[l]), ptr(pl.G[l]),
                                           ptr(pl.Cc[l]), ptr(pl.R[l]), ptr(pl.dxproj), ss), "cham_rnn_bwd")
                if l == 0:   # -> gradient w.r.t. the CAR tanh pre-activation of the clicked-input rows (main lane waits for it)
                    dxp = pl.dxproj
                    if pos is not None:
                        dxp = pl.dxproj_c
                        check(lib.cham_rows_gather(ptr(pl.dxproj), ptr(pos), BT, NGH, ptr(dxp), ss), "cham_rows_gather")
                    rt.gemm(dxp, p('rnn0/Wx'), pl.dZ2, BT, C, NGH, NGH, NGH, C, transB=1, dref=pl.Z2, ldr=C, dact=ACT_TANH)
                    # CAR layer-2 dgrad of the clicked rows right here (0.1 ms; behind the W2 wgrad it would starve: that kernel's
                    # 256 workgroups fill every CU's register file for 4 ms) - the main lane's PreCAR backward waits for it ...
                    rt.gemm(pl.dZ2, p('W2'), pl.dZ1, BT, C, C, C, C, C, transB=1, dref=pl.Z1, ldr=C, dact=ACT_LEAKY)
                    e_dZ1in = mark()
                    if crit_first and not use_aux:
                        fc_wgrads()
                    if b16:
                        # ... and the CAR layer-2 weight gradient: the candidate rows (bf16, TN) + the clicked-input rows (fp32), runs beside it
                        rt.gemm_b16(pl.Z1c, C, 1, dZ2c, C, 0, g('W2'), C, 1, C, C, Rc, splits=0, dma=rt.b16_dma)
                        rt.gemm(pl.Z1, pl.dZ2, g('W2'), C, C, BT, C, C, C, transA=1, splits=0, accumulate=1)
                        if dm_fused_b16:      # the per-position sums of the stored gradient rows, written by the fused kernel
                            rt.colsum(pl.b2part, C, BT, C, g('b2'))
                        else:
                            rt.colsum(dZ2c, C, Rc, C, g('b2'), b16=True)
                        rt.colsum(pl.dZ2, C, BT, C, g('b2'), accumulate=1)
                    elif use_p3 and use_aux:
                        # (round 5) the small weight / bias gradients leave this lane for the third one - every producer they read is
                        # final at e_dZ1in - and the W2 weight gradient (candidate rows from the planes, TN split-K, + the clicked rows)
                        # starts the moment the main lane's CAR dgrad has finished, not behind ~1 ms of small launches
                        with aux(e_dZ1in):
                            if crit_first:
                                fc_wgrads()
                            rt.colsum(pl.b2part, C, BT, C, g('b2'))
                            rt.colsum(pl.dZ2, C, BT, C, g('b2'), accumulate=1)
                            rt.gemm(pl.Z2, dxp, g('rnn0/Wx'), C, NGH, BT, C, NGH, NGH, transA=1, splits=0)
                            rt.gemm(pl.hprev[0], pl.dxproj, g('rnn0/Wh'), Hp, 2 * Hp, BTf, Hp, NGH, 2 * Hp, transA=1, splits=0, force_f32=True)
                            rt.colsum(pl.dxproj, NGH, BTf, NGH, g('rnn0/b'))
                            e_auxdone = mark()
                        rt.side_stream.wait_event(e_cdgrad)
                        w2_wgrad_planes(rt.p3_w2_splits)
                        rt.gemm(pl.Z1, pl.dZ2, g('W2'), C, C, BT, C, C, C, transA=1, splits=0, accumulate=1)
                        continue
                    elif use_p3:
                        # ... and the CAR layer-2 weight gradient: the candidate rows from their planes (TN, split-K), the clicked-input
                        # rows (fp32) added by the on-the-fly kernel; b2 from the per-position partial sums of k_mulpred_bwd_p3
                        def w2_grads():
                            if w2_main:       # (the candidate rows' share was written by the main lane: behind e_cdgrad it is complete)
                                rt.side_stream.wait_event(e_w2main)
                            else:
                                if on:
                                    rt.side_stream.wait_event(e_cdgrad)
                                w2_wgrad_planes(rt.p3_w2_splits)
                            rt.gemm(pl.Z1, pl.dZ2, g('W2'), C, C, BT, C, C, C, transA=1, splits=0, accumulate=1)
                        rt.colsum(pl.b2part, C, BT, C, g('b2'))
                        rt.colsum(pl.dZ2, C, BT, C, g('b2'), accumulate=1)
                        if on and not w2_main:
                            # the plane weight gradient waits for the main lane's CAR dgrad (two one-workgroup-per-CU matrix kernels only
                            # time-slice the chip): everything this lane still has to do that does NOT wait - bias sums, the recurrent and
                            # input-projection weight gradients - goes first and runs under that dgrad; the big GEMM is the lane's last work
                            deferred_w2 = w2_grads
                        else:
                            w2_grads()
                    else:
                        # ... and the CAR layer-2 weight gradient over ALL rows, the second-largest GEMM of the step, runs beside it
                        if on:
                            rt.side_stream.wait_event(e_cdgrad)
                        rt.gemm(pl.Z1, pl.dZ2, g('W2'), C, C, Rall, C, C, C, transA=1, splits=0)
                        rt.colsum(pl.dZ2, C, Rall, C, g('b2'))
                    rt.gemm(pl.Z2, dxp, g('rnn0/Wx'), C, NGH, BT, C, NGH, NGH, transA=1, splits=0)
                else:
                    rt.gemm(pl.dxproj, p('rnn%d/Wx' % l), pl.drnn, BTf, Hp, NGH, NGH, NGH, Hp, transB=1)
                    rt.gemm(rnn_y(l - 1), pl.dxproj, g('rnn%d/Wx' % l), Hp, NGH, BTf, Hp, NGH, NGH, transA=1, splits=0)
                    if drop:
                        dropout(pl.drnn, pl.drnn, BTf, Hp, Hp, 20 + l - 1, 20 + l - 1, 1, None)
                # recurrent weights: their forward product runs in the fp32 time-step kernel -> fp32 wgrad in every mode
                rt.gemm(pl.hprev[l], pl.dxproj, g('rnn%d/Wh' % l), Hp, 2 * Hp, BTf, Hp, NGH, 2 * Hp, transA=1, splits=0, force_f32=True)
                if cell == 1:   # candidate kernel: (r * h_prev)^T dz_c
                    rt.gemm(pl.RH[l], pl.dxproj[:, 2 * Hp:], g('rnn%d/Wch' % l), Hp, Hp, BTf, Hp, NGH, Hp, transA=1, splits=0, force_f32=True)
                rt.colsum(pl.dxproj, NGH, BTf, NGH, g('rnn%d/b' % l))
            if deferred_w2 is not None:
                deferred_w2()
        def precar_backward(ws):
            """PreCAR combine scatter, W1 weight gradients, feature / embedding backward (on whatever lane is current)."""
            st = _stream()
            if drop:        # dense PreCAR backward: one weight gradient over all CAR rows, d(input rows) masked, then summed per
                #             position / per item row by the same deterministic scatter as the factorised path (width Fc + Fi)
                Fw = drop['Fw']
                dZ1 = pl.dZ1
                if b16:      # fp32 image of [clicked rows (fp32) ; candidate rows (bf16-resident)] for the dense PreCAR backward
                    dZ1 = pl.dZ1f
                    dZ1[:BT].copy_(pl.dZ1[:BT])
                    check(lib.cham_upcast_b16(ptr(pl.dZ1c), Rc * C, dZ1[BT:].data_ptr(), st), "cham_upcast_b16")
                rt.gemm(pl.Xd, dZ1, drop['gW1'], Fw, C, Rall, Fw, C, C, transA=1, splits=0)
                rt.colsum(dZ1, C, Rall, C, g('b1'))
                rt.gemm(dZ1, drop['W1'], pl.dXd, Rall, Fw, C, C, C, Fw, transB=1)
                dropout(pl.dXd, pl.dXd, BT, Fw, Fw, 16, 16, 1, pos, Fc, Fc - L.f_ctx)
                dropout(pl.dXd[BT:], pl.dXd[BT:], Rc, Fw, Fw, 17, 18, NC, pos, Fc, Fc - L.f_ctx)
                check(lib.cham_combine_bwd(ptr(pl.dXd), Fw, BT, N, pmax, ptr(neg_slot), ptr(pl.dUx), ptr(pl.dVx), ptr(pl.drop_ws),
                                           pl.drop_ws.numel() * 4, st), "cham_combine_bwd")
                pl.dXc[:BT].copy_(pl.dUx[:BT, :Fc]); pl.dXi[:RV].copy_(pl.dVx[:RV, Fc:])
            elif b16:
                check(lib.cham_combine_bwd_b16(ptr(pl.dZ1), ptr(pl.dZ1c), C, BT, N, pmax, ptr(neg_slot), ptr(pl.dU), ptr(pl.dV), ptr(ws),
                                               ws.numel() * 4, st), "cham_combine_bwd_b16")
            elif h2 and getattr(pl, 'gsum', None) is not None:      # dU from the group sums of the CAR dgrad's epilogue
                check(lib.cham_combine_bwd_gs(ptr(pl.dZ1), C, BT, N, pmax, ptr(neg_slot), ptr(pl.dU), ptr(pl.dV), ptr(ws), ws.numel() * 4,
                                              ptr(pl.gsum), pl.gsum.numel() * 4, st), "cham_combine_bwd_gs")
            else:
                check(lib.cham_combine_bwd(ptr(pl.dZ1), C, BT, N, pmax, ptr(neg_slot), ptr(pl.dU), ptr(pl.dV), ptr(ws), ws.numel() * 4, st),
                      "cham_combine_bwd")
            n_rows_cat = d['cat'].shape[1]

            def feature_bwd(dX, Xraw, R, F, gname, bname):
                # dgamma / dbeta column sums: the coalesced two-launch form through this lane's workspace (CHAM_FEATURE_BWD_WS=0: one workgroup per column)
                if rt.feature_bwd_ws:
                    wsl = rt._lane_ws('gemm_ws')
                    check(lib.cham_feature_bwd_ws(ptr(dX), ptr(Xraw), R, F, ptr(g(gname)), ptr(g(bname)), ptr(wsl), wsl.numel() * 4, _stream()),
                          "cham_feature_bwd_ws")
                else:
                    check(lib.cham_feature_bwd(ptr(dX), ptr(Xraw), R, F, ptr(g(gname)), ptr(g(bname)), _stream()), "cham_feature_bwd")

            def ctx_chain():      # user-context half of the PreCAR input: weight gradient, bias, d(features), scale / center, embedding tables
                if not drop:
                    rt.gemm(pl.Xc_s, pl.dU, g('W1c'), Fc, C, BT, Fc, C, C, transA=1, splits=0)
                    rt.colsum(pl.dU, C, BT, C, g('b1'))
                    rt.gemm(pl.dU, p('W1c'), pl.dXc, BT, Fc, C, C, C, Fc, transB=1)
                feature_bwd(pl.dXc, pl.Xc_raw, BT, Fc, 'gamma_ctx', 'beta_ctx')
                for kind, feat, c0, dim, card, off in rt.ctx_emb_groups:
                    check(lib.cham_emb_grad_scan(ptr(pl.dXc), BT, Fc, c0, dim, ptr(p('gamma_ctx')), d['cat'].data_ptr() + 8 * feat * n_rows_cat,
                                                 None, card, rt.grads.data_ptr() + 4 * off, _stream()), "cham_emb_grad_scan")

            def item_chain():     # item half
                if not drop:
                    rt.gemm(pl.Xi_s, pl.dV, g('W1i'), Fi, C, RV, Fi, C, C, transA=1, splits=0)
                    rt.gemm(pl.dV, p('W1i'), pl.dXi, RV, Fi, C, C, C, Fi, transB=1)
                feature_bwd(pl.dXi, pl.Xi_raw, RV, Fi, 'gamma_item', 'beta_item')
                for kind, feat, c0, dim, card, off in rt.item_emb_groups:
                    if kind == COL_ITEMEMB:
                        if getattr(pl, 'grouped_ev', None) is not None:
                            torch.cuda.current_stream().wait_event(pl.grouped_ev)
                        check(lib.cham_emb_grad_grouped(ptr(pl.dXi), RV, Fi, c0, dim, ptr(p('gamma_item')), ptr(pl.ids_all), ptr(pl.perm),
                                                        ptr(pl.seg), rt.grads.data_ptr() + 4 * off, _stream()), "cham_emb_grad_grouped")
                    else:
                        check(lib.cham_emb_grad_scan(ptr(pl.dXi), RV, Fi, c0, dim, ptr(p('gamma_item')),
                                                     rt.meta_cat.data_ptr() + 8 * feat * rt.n_items, ptr(pl.ids_all), card,
                                                     rt.grads.data_ptr() + 4 * off, _stream()), "cham_emb_grad_scan")

            # Round 6: the two halves are independent once dU / dV exist, and they are the SERIAL tail of the step - ~25 latency-bound launches
            # behind the W2 weight gradient with the chip otherwise idle.  The user-context half goes to the third lane, the item half
            # stays here (CHAM_TAIL_SPLIT=0: both on this lane, the order of rounds 1-5).  Same kernels, same arguments: same results.
            if on and rt.tail_split and not drop and st == main_stream.cuda_stream:
                e_duv = mark()
                with aux(e_duv):
                    ctx_chain()
                    e_ctx = mark()
                item_chain()
                main_stream.wait_event(e_ctx)
            else:
                ctx_chain()
                item_chain()

        if on:
            main_wait(e_dZ1in)
        precar_backward(rt.gemm_ws)      # beside the W2 wgrad of the side lane
        rt.join()
        if e_auxdone is not None:
            main_stream.wait_event(e_auxdone)

    def adam_lr_t(self, t):
        """tf.train.AdamOptimizer's bias-corrected learning rate of optimizer step t (>= 1)."""
        return self.lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)

    def apply_gradients(self):
        """tf.train.AdamOptimizer(lr, 0.9, 0.999, 1e-8).apply_gradients (nar_model.py:708-722) + the dense L2 term."""
        rt, L = self.rt, self.rt.layout
        rt.global_step += 1           # (also invalidates the bf16 weight shadows: NARRuntime.refresh_shadows keys on it)
        if not rt.capturing and getattr(self, '_plan', None) is not None:
            self._plan.eager_steps = getattr(self._plan, 'eager_steps', 0) + 1      # (GraphedTrainStep captures warm shapes only)
        t = rt.global_step
        lr_t = self.adam_lr_t(t)
        if rt.dev_scalars:
            rt.set_step_scalars(lr_t=lr_t, fields=4)

        def adam(a, b, grads, g_off):       # Adam on the parameter range [a, b); grads[g_off + i] pairs with flat[a + i]
            n_reg = min(max(L.n_reg - a, 0), b - a)
            if rt.dev_scalars:      # lr_t from the step-scalar record
                check(rt.lib.cham_adam_tf_dev(rt.flat.data_ptr() + 4 * a, grads.data_ptr() + 4 * g_off, rt.m.data_ptr() + 4 * a,
                                              rt.v.data_ptr() + 4 * a, b - a, n_reg, float(self.reg_weight_decay), ptr(rt.scalars),
                                              0.9, 0.999, 1e-8, _stream()), "cham_adam_tf_dev")
                return
            check(rt.lib.cham_adam_tf(rt.flat.data_ptr() + 4 * a, grads.data_ptr() + 4 * g_off, rt.m.data_ptr() + 4 * a,
                                      rt.v.data_ptr() + 4 * a, b - a, n_reg, float(self.reg_weight_decay), float(lr_t),
                                      0.9, 0.999, 1e-8, _stream()), "cham_adam_tf")

        if rt.dp_sharded is not None:       # reduce-scatter -> Adam on this rank's slice -> all-gather (parallel.py)
            rt.dp_sharded(rt.grads, rt.flat, adam)
            return
        if rt.dp_allreduce is not None:
            rt.dp_allreduce(rt.grads)
        adam(0, L.total, rt.grads, 0)

    def train_step_microbatched(self, features, labels, micro_sessions, global_features=None, global_labels=None, row_begin=0):
        """One optimizer step over the batch processed as row shards of ``micro_sessions`` sessions (activations of the
        B*T*(1+N) candidate rows bounded by the shard; BASELINE config 5 = 4096 sessions x 200 negatives).  Every shard sees
        the GLOBAL ids (pool, max timestamp, sum(mask), sampler keyed by the global row), gradients accumulate, ONE Adam.
        Same result as train_step on the whole batch up to fp32 summation order (tests: shard-sum / micro-batch parity)."""
        from .parallel import slice_batch
        rt = self.rt
        gf = features if global_features is None else global_features
        gl = labels if global_labels is None else global_labels
        n = np.asarray(features['item_clicked']).shape[0]
        if getattr(rt, 'grads_acc', None) is None:
            rt.grads_acc = torch.empty_like(rt.grads)
            rt.loss_acc = torch.zeros(3, dtype=torch.float32, device=rt.device)
        self._accumulating = True
        for k, b in enumerate(range(0, n, micro_sessions)):
            e = min(n, b + micro_sessions)
            f, l = slice_batch(features, labels, b, e)
            self.forward(self.upload_batch(f, l, gf, gl, row_begin=row_begin + b))
            self.backward()
            check(rt.lib.cham_accumulate(ptr(rt.grads_acc), ptr(rt.grads), rt.layout.total, int(k == 0), _stream()), "cham_accumulate")
            check(rt.lib.cham_loss_accumulate(ptr(rt.loss_acc), ptr(self._plan.loss), int(k == 0), _stream()), "cham_loss_accumulate")
        self._accumulating = False
        rt.grads, rt.grads_acc = rt.grads_acc, rt.grads          # Adam (and the data-parallel all-reduce) read rt.grads
        self.apply_gradients()
        self.total_loss = rt.loss_acc
        return self.total_loss

    def stage_next(self, dataset):
        """Training-loop hook (estimator.Estimator.train): upload the batch AFTER the current one and draw its negatives now
        (presample), so that neither the H2D copies nor the sampler sit at the head of the next step."""
        self._staged = None
        state = self.articles_recent_pop_norm
        if not (self.rt.presample and self.is_training and getattr(state, 'is_device', False)):
            return
        nxt = dataset.peek()
        if nxt is None:
            return
        d = self.upload_batch(nxt[0], nxt[1])
        self.presample(d)
        self._staged = (nxt[0]['item_clicked'], d)

    def train_step(self, device_batch=None):
        """One optimizer step on the current batch (the reference's ``session.run(model.train)``)."""
        d = device_batch
        if d is None:
            staged, self._staged = getattr(self, '_staged', None), None
            if staged is not None and staged[0] is self.inputs['item_clicked']:        # the very arrays stage_next() uploaded
                d = staged[1]
            else:
                d = self.upload_batch(self.inputs, self.labels)
        with _roctx.range_("NAR step: forward (K0 sampler .. loss)"):        # (CHAM_ROCTX=1: roctx ranges for rocprofv3 --marker-trace)
            self._defer_loss = True          # the loss record may finish on the side lane: backward() joins it
            try:
                pl = self.forward(d)
            finally:
                self._defer_loss = False
        if self.eval_cold_start:       # nar_model.py:520: the ranked candidates are also needed while TRAINING for the cold-start analysis
            self._rank_items(pl, d)
        with _roctx.range_("NAR step: backward"):
            self.backward()
        with _roctx.range_("NAR step: exchange + L2 + TF-Adam"):
            self.apply_gradients()
        return self.total_loss

    @staticmethod
    def eval_step_key(global_step, eval_iter):
        """Sampler key of the eval_iter-th batch of an evaluate() call (distinct from every training step's key)."""
        return (global_step + 1000003 * (eval_iter + 1)) & 0xFFFFFFFF

    def _rank_items(self, pl, d):
        """rank_items_by_predicted_prob (nar_model.py:777-794) of the last forward pass -> self._eval (what the hook fetches)."""
        rt, dev = self.rt, self.rt.device
        ev = self._eval
        if ev is None or ev['pred_ids'].shape != (pl.B, pl.T, pl.NC):
            ev = self._eval = dict(pred_ids=torch.zeros(pl.B, pl.T, pl.NC, dtype=torch.int64, device=dev),
                                   pred_probs=torch.zeros(pl.B, pl.T, pl.NC, dtype=torch.float32, device=dev),
                                   label_rank=torch.zeros(pl.B, pl.T, dtype=torch.int32, device=dev))
        if pl.pos is None:
            check(rt.lib.cham_rank_items(ptr(pl.probs), ptr(d['label_next']), ptr(pl.neg_ids), ptr(pl.mask), pl.BT, pl.N,
                                         ptr(ev['pred_ids']), ptr(ev['pred_probs']), ptr(ev['label_rank']), _stream()),
                  "cham_rank_items")
        else:       # rank the valid positions, then back into the [B, T, 1+N] layout (padded positions: ids 0, probs 0, rank -1)
            P, NC = pl.P, pl.NC
            c = ev.get('compact')
            if c is None or c[0].shape[0] < P:
                c = ev['compact'] = (torch.zeros(pl.BT, NC, dtype=torch.int64, device=dev), torch.zeros(pl.BT, NC, dtype=torch.float32, device=dev),
                                     torch.zeros(pl.BT, dtype=torch.int32, device=dev))
            check(rt.lib.cham_rank_items(ptr(pl.probs), ptr(d['ln_rows']), ptr(pl.cur_neg_ids), ptr(pl.mask), P, pl.N,
                                         ptr(c[0]), ptr(c[1]), ptr(c[2]), _stream()), "cham_rank_items")
            ev['pred_ids'].zero_(); ev['pred_probs'].zero_(); ev['label_rank'].fill_(-1)
            for src, dst, words in ((c[0], ev['pred_ids'], 2 * NC), (c[1], ev['pred_probs'], NC), (c[2], ev['label_rank'], 1)):
                check(rt.lib.cham_rows_scatter(ptr(src), ptr(pl.pos), P, words, ptr(dst), _stream()), "cham_rows_scatter")

    def evaluate_step(self, device_batch=None):
        """EVAL-mode ``session.run``: forward with the eval negative-sample counts (nar_trainer_gcom.py:240-242) +
        rank_items_by_predicted_prob (nar_model.py:777-794)."""
        d = device_batch if device_batch is not None else self.upload_batch(self.inputs, self.labels)
        pl = self.forward(d)
        self._rank_items(pl, d)
        self._eval_iter += 1
        return self.total_loss

    def run_step(self):
        """What ``session.run(train_op | eval fetches)`` does for the current value of the input handles."""
        return self.train_step() if self.is_training else self.evaluate_step()

    # convenience for tests / hooks -------------------------------------------------------------------
    def outputs_numpy(self):
        pl = self._plan
        torch.cuda.synchronize()
        return dict(loss=pl.loss.cpu().numpy(), logits=pl.full_rows(pl.logits).view(pl.B, pl.T, pl.NC).cpu().numpy(),
                    probs=pl.full_rows(pl.probs).view(pl.B, pl.T, pl.NC).cpu().numpy(), neg_items=pl.neg_ids.cpu().numpy(),
                    neg_slot=pl.neg_slot.cpu().numpy(), pool=pl.pool.cpu().numpy(), meta=pl.meta.cpu().numpy())


class GraphedTrainStep:
    """One training step of a NARModuleModel - forward, backward, TF-Adam, the recent-clicks state update and the NEXT batch's negative
    sampling - captured ONCE in a hipGraph (torch.cuda.CUDAGraph: stream capture of the step's lanes) and replayed per step.  The reference
    runs a step as ONE session.run (nar_model.py:1434-1470); eagerly this runtime submits ~135 launches through ctypes (0.9-1.0 ms of host
    time per step: what bounds a 32-session data-parallel shard or short sessions); a replay submits three: [the batch -> the step's input
    slot (one device-to-device copy of the packed block), cham_step_scalars_set, graph launch].

    What makes the step capturable: its launches carry no per-step host value - sampler key, max time stamp, sum(mask) and Adam's lr_t are
    read from the device record (csrc/common.h ChamStepScalars) - and its inputs sit at fixed addresses (the slot).  Limits (checked by
    supports()): one batch shape per object (B, T, global batch, row offset; NOT compacted - i.e. every position valid, or CHAM_COMPACT=0),
    device-resident ClickedItemsState, keep_prob 1, no data-parallel exchange inside the step, pinned upload ring (the packed block).
    Results are BIT-IDENTICAL to the eager step (tests/test_graph_step_gpu.py): same kernels, same arguments, same order per lane.

    Sampler hand-over: the graph's forward reads sampler-output set A; behind the state update the graph draws the NEXT batch's negatives
    (key .step_next) into set B from the `next` input slot and, as its last node, copies B -> A.  The first replay is preceded by an eager
    draw of the first batch's negatives into A."""

    def __init__(self, model, state):
        self.model, self.state, self.rt = model, state, model.rt
        self.graph, self.key, self.slots, self._expect = None, None, None, None
        self.replays = 0

    @staticmethod
    def shape_key(d):
        return (d['B'], d['T'], d['Bg'], d['row_begin'], d['P'], d.get('_total'), d.get('_layout'))

    def supports(self, d):
        """None when batch d can go through the captured step, else the reason (str)."""
        m, rt = self.model, self.rt
        if not rt.dev_scalars:
            return "CHAM_DEV_SCALARS=0"
        if d.get('pos') is not None or d['P'] != d['B'] * d['T']:
            return "compacted batch (the valid-position count varies from batch to batch)"
        if d.get('_block') is None:
            return "no packed device block (pinned upload ring off)"
        if not getattr(self.state, 'is_device', False) or self.state.stream is None:
            return "host-side ClickedItemsState"
        if self.state.n_updates < 1:
            return "empty recent-clicks state (the very first batch takes its normalisation statistics from its own rows: another launch sequence)"
        if getattr(rt, 'dp_active', False):
            return "data-parallel exchange inside the step"
        if m.keep_prob < 1.0 or m.eval_cold_start or not m.is_training or not rt.presample:
            return "dropout / cold-start analysis / presampling off"
        if self.key is not None and self.shape_key(d) != self.key:
            return "another batch shape than the captured one"
        if self.graph is None:      # capture needs a warm shape: buffers, workspaces, the step-scalar record and the kernels' dynamic-LDS
            # attributes are set up by the first eager step - none of that belongs inside a stream capture
            pl = rt._plans.get((d['B'], d['T'], m.negative_samples, m.negative_sample_from_buffer, d['Bg'], rt.p3, rt.h2))
            if pl is None or getattr(pl, 'eager_steps', 0) < 1 or rt.scalars is None:
                return "no eager training step has run on this batch shape yet (run one train_step first)"
        return None

    def _make_slot(self, d):
        blk = torch.zeros(max(d['_total'], 256), dtype=torch.uint8, device=self.rt.device)
        slot = dict(B=d['B'], T=d['T'], Bg=d['Bg'], row_begin=d['row_begin'], sum_mask=None, max_ts=None, P=d['P'], pos=None, uploaded=None,
                    _block=blk, _total=d['_total'], _layout=d['_layout'])
        for k, off, dt, shape in d['_layout']:
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            slot[k] = blk[off:off + n].view(_TORCH_DTYPE[dt]).view(shape)
        slot.update(ic_rows=slot['item_clicked'].view(-1), ln_rows=slot['label_next'].view(-1), ets_rows=slot['event_ts'].view(-1))
        return slot

    def _load(self, slot, d):
        torch.cuda.current_stream().wait_event(d['uploaded'])
        slot['_block'][:d['_total']].copy_(d['_block'][:d['_total']], non_blocking=True)

    def _capture(self, d, d_next):
        m, rt, state = self.model, self.rt, self.state
        self.key = self.shape_key(d)
        cur, nxt = self._make_slot(d), self._make_slot(d_next)
        self.slots = (cur, nxt)
        pl = rt.plan(d['B'], d['T'], m.negative_samples, m.negative_sample_from_buffer, d['Bg'])
        self._load(cur, d); self._load(nxt, d_next)
        m.feed_state(state, state)
        # the first batch's negatives, eagerly, into the set the captured forward reads (key by value: this is step rt.global_step)
        a = pl._samp_cur
        cur_scalars = dict(cur, max_ts=d['max_ts'], sum_mask=d['sum_mask'])
        m._neg_sample(pl, cur_scalars, rt.global_step, a, _stream())
        torch.cuda.synchronize()
        state.updated_event = state.consumed_event = None          # (everything has finished: no event of the eager past is waited for inside the capture)
        pl.grouped_ev = None
        g = torch.cuda.CUDAGraph()
        rt.capturing = True
        step0, upd0 = rt.global_step, state.n_updates
        try:
            with torch.cuda.graph(g):
                cur['_presampled'] = (pl, a, rt.global_step, None)
                m.forward(cur)
                m.backward()
                m.apply_gradients()                     # (increments rt.global_step: undone below - nothing has executed)
                state.update_from_device_batch(cur['aci'], cur['g_event_ts'])
                ok = m.presample(nxt)                   # -> set 1 - a, key .step_next
                assert ok, "presample() declined inside the capture"
                ev = nxt.pop('_presampled')[3]
                main = torch.cuda.current_stream()
                main.wait_event(ev)
                if state.updated_event is not None:
                    main.wait_event(state.updated_event)
                for name in ('neg_ids', 'neg_slot', 'pool', 'canon', 'meta'):
                    pl._samp[a][name].copy_(pl._samp[1 - a][name], non_blocking=True)
                # ... and the next batch becomes the current one: the host loads ONE slot per step
                cur['_block'].copy_(nxt['_block'], non_blocking=True)
        finally:      # (also when the capture fails: nothing of it has executed, the eager step remains usable)
            rt.capturing = False
            rt.global_step, state.n_updates = step0, upd0
            state.updated_event = state.consumed_event = None
            pl.grouped_ev = None
            pl.use_sampler_set(a)
            cur.pop('_presampled', None); nxt.pop('_presampled', None)
        self.graph, self.plan = g, pl
        return g

    def step(self, d, d_next):
        """One optimizer step on batch d; d_next = the batch of the following step (its negatives are drawn behind this step).  Returns the
        device loss tensor [total, xe, reg] (as train_step)."""
        why = self.supports(d) or self.supports(d_next)
        if why is not None:
            raise RuntimeError("GraphedTrainStep: %s" % why)
        m, rt, state = self.model, self.rt, self.state
        if self.graph is None:
            self._capture(d, d_next)
        else:
            if d is not self._expect:
                # not the batch the previous replay moved into the current slot and drew negatives for: load it and draw them now,
                # eagerly, with this step's key by value (what the eager forward does for a batch that was not presampled)
                self._load(self.slots[0], d)
                m._neg_sample(self.plan, d, rt.global_step, self.plan._samp_cur, _stream())
            self._load(self.slots[1], d_next)
        self._expect = d_next
        t = rt.global_step + 1
        check(rt.lib.cham_step_scalars_set(ptr(rt.scalars), rt.global_step & 0xFFFFFFFF, (rt.global_step + 1) & 0xFFFFFFFF, int(d['max_ts']),
                                           float(d['sum_mask']), float(m.adam_lr_t(t)), 7, _stream()), "cham_step_scalars_set")
        self.graph.replay()
        rt.global_step = t
        state.n_updates += 1
        state.updated_event = state.consumed_event = None
        self.replays += 1
        m._plan, m._d, m.total_loss = self.plan, self.slots[0], self.plan.loss
        return m.total_loss


class ItemsStateUpdaterHook:
    """SessionRunHook around every step (nar_model.py:1369-1700), NAR-path subset:
      * before_run (:1434-1470): feeds ``articles_recent_pop_norm`` / ``pop_recent_items_buffer`` (+ the ACE matrix and
        metadata placeholders, which are already resident in HBM here) and names the fetches;
      * after_run (:1504-1650): in EVAL, HitRate@n / MRR@n of the ranked candidates (the reference's numpy streaming
        metrics, metrics.py, plus the TF streaming twins nar_model.py:826-835, 859-885); always: the recent-clicks state
        update from the batch (:1635-1649);
      * begin / end (:1410-1431, :1669-1695): state snapshot around evaluation, metrics appended to
        ``eval_sessions_metrics_log``;
      * ``eval_metrics_by_session_position`` (HitRate@n per click position, :1718) and ``eval_cold_start`` (steps between an
        item's first click and its first top-n recommendation, :1480-1494, 1621-1625, 1662-1666; runs in TRAIN and EVAL).
    Out of scope: the baseline recommenders (``eval_benchmark_classifiers`` must be empty), the
    co-occurrence matrix (:1650), novelty / diversity / coverage metrics."""

    def __init__(self, mode, model, eval_metrics_top_n, clicked_items_state, eval_sessions_metrics_log,
                 sessions_negative_items_log=None, sessions_chameleon_recommendations_log=None,
                 content_article_embeddings_matrix=None, articles_metadata=None, eval_negative_sample_relevance=None,
                 eval_benchmark_classifiers=[], eval_metrics_by_session_position=False, eval_cold_start=False,
                 eval_metric_ops=None):
        if eval_benchmark_classifiers:
            raise NotImplementedError("baseline recommenders (nar/benchmarks) are out of scope: pass --disable_eval_benchmarks")
        self.eval_cold_start = eval_cold_start
        self.eval_metrics_by_session_position = eval_metrics_by_session_position
        self.mode, self.model = mode, model
        self.eval_metrics_top_n = eval_metrics_top_n
        self.clicked_items_state = clicked_items_state
        self.eval_sessions_metrics_log = eval_sessions_metrics_log
        self.sessions_negative_items_log = sessions_negative_items_log
        self.sessions_chameleon_recommendations_log = sessions_chameleon_recommendations_log
        self.content_article_embeddings_matrix = content_article_embeddings_matrix
        self.articles_metadata = articles_metadata or {}
        self.eval_metric_ops = eval_metric_ops or {}

    def begin(self):
        if self.mode == ModeKeys.EVAL:
            from .metrics import HitRate, HitRateBySessionPosition, MRR
            self.clicked_items_state.save_state_checkpoint()                    # nar_model.py:1415
            self.eval_streaming_metrics_last = {}
            self.streaming_metrics = [HitRate(self.eval_metrics_top_n), MRR(self.eval_metrics_top_n)]       # create_eval_metrics, :1696-1721
            if self.eval_metrics_by_session_position:
                self.streaming_metrics.append(HitRateBySessionPosition(self.eval_metrics_top_n))
            self.stats_logs = []

    def after_create_session(self, session=None, coord=None):
        pass

    def before_run(self, run_context):
        from .estimator import SessionRunArgs
        m = self.model
        fetches = {'clicked_items': m.item_clicked, 'clicked_timestamps': m.event_timestamp,
                   'next_item_labels': m.next_item_label, 'last_item_label': m.label_last_item,
                   'session_id': m.session_id, 'user_id': m.user_id}
        if self.eval_cold_start or self.mode == ModeKeys.EVAL:                 # nar_model.py:1444
            fetches.update(predicted_item_ids=m.predicted_item_ids, eval_batch_negative_items=m.batch_negative_items,
                           batch_items_count=m.batch_items_count, batch_unique_items_count=m.batch_unique_items_count,
                           predicted_item_probs=m.predicted_item_probs, label_rank=m.label_rank)
        dev = getattr(self.clicked_items_state, 'is_device', False)
        feed_dict = {m.ph_articles_recent_pop_norm: self.clicked_items_state if dev else self.clicked_items_state.get_articles_recent_pop_norm(),
                     m.ph_pop_recent_items_buffer: self.clicked_items_state if dev else self.clicked_items_state.get_recent_clicks_buffer(),
                     m.ph_content_article_embeddings_matrix: self.content_article_embeddings_matrix}
        for name in self.articles_metadata:
            if name in m.ph_articles_metadata:
                feed_dict[m.ph_articles_metadata[name]] = self.articles_metadata[name]
        return SessionRunArgs(fetches=fetches, feed_dict=feed_dict)

    def after_run(self, run_context, run_values):
        r = run_values.results
        clicked_items = r['clicked_items']
        clicked_timestamps = np.squeeze(r['clicked_timestamps'], axis=-1)
        next_item_labels, last_item_label = r['next_item_labels'], r['last_item_label']
        sessions_ids = r['session_id']
        if self.mode == ModeKeys.EVAL:
            predicted_item_ids = r['predicted_item_ids']
            # TF streaming twins (sparse_recall_at_top_k, define_mrr_metric) from the device-side rank of the positive
            rank = r['label_rank']
            valid = rank >= 0
            hit = valid & (rank < self.eval_metrics_top_n)
            if 'hitrate_at_n' in self.eval_metric_ops:
                self.eval_metric_ops['hitrate_at_n'].update(hit.sum(), valid.sum())
            if 'mrr_at_n' in self.eval_metric_ops:
                self.eval_metric_ops['mrr_at_n'].update((1.0 / (1.0 + rank[hit])).sum(), valid.sum())
            self.eval_streaming_metrics_last = {k: v.result() for k, v in self.eval_metric_ops.items()}
            if self.sessions_negative_items_log is not None:                   # :1530-1541
                for session_id, labels, neg_items in zip(sessions_ids, next_item_labels, r['eval_batch_negative_items']):
                    self.sessions_negative_items_log.append(
                        {'session_id': str(session_id),
                         'negative_items': [n for l, n in zip(labels.tolist(), neg_items.tolist()) if l != 0]})
            if self.sessions_chameleon_recommendations_log is not None:        # :1544-1580
                probs = r['predicted_item_probs'].round(decimals=7)
                pop = self.clicked_items_state.get_articles_recent_pop_norm()
                for session_id, labels, ids, pp in zip(sessions_ids, next_item_labels, predicted_item_ids, probs):
                    keep = labels != 0
                    self.sessions_chameleon_recommendations_log.append(
                        {'session_id': str(session_id), 'next_click_labels': labels[keep].tolist(),
                         'predicted_item_ids': ids[keep].tolist(), 'predicted_item_probs': pp[keep].tolist(),
                         'predicted_item_norm_pop': pop[ids[keep]].round(decimals=7).tolist()})
            self.stats_logs.append({'batch_items_count': r['batch_items_count'],
                                    'batch_unique_items_count': r['batch_unique_items_count'],
                                    'batch_sessions_count': len(sessions_ids)})
            from .evaluation import compute_metrics_results, update_metrics
            labels_norm_pop = preds_norm_pop = None
            if self.eval_metrics_by_session_position:                          # nar_model.py:1590-1593
                labels_norm_pop = self.clicked_items_state.get_articles_recent_pop_norm()[next_item_labels]
            update_metrics(predicted_item_ids, next_item_labels, labels_norm_pop, preds_norm_pop, clicked_items,
                           self.streaming_metrics, recommender='chameleon')
            self.eval_streaming_metrics_last.update(compute_metrics_results(self.streaming_metrics, recommender='chameleon'))
        if self.eval_cold_start:                                               # nar_model.py:1621-1625, both modes
            self.update_items_cold_start_state(r['user_id'], clicked_items, next_item_labels, r['eval_batch_negative_items'],
                                               r['predicted_item_ids'])
        # state update, nar_model.py:1635-1649
        if getattr(self.clicked_items_state, 'is_device', False):     # straight from the batch tensors already in HBM
            d = self.model._d
            self.clicked_items_state.update_from_device_batch(d['aci'], d['g_event_ts'])
            return
        from .clicked_items_state import batch_clicks_for_state
        ids, ts = batch_clicks_for_state(clicked_items, last_item_label, clicked_timestamps)
        self.clicked_items_state.update_items_state(ids, ts)

    def update_items_cold_start_state(self, users_ids, clicked_items, next_item_labels, eval_batch_negative_items, predicted_item_ids):
        """nar_model.py:1480-1494 (the benchmark recommenders' part, :1496-1501, is out of scope)."""
        st = self.clicked_items_state
        clicked_items_nonzero = set(np.asarray(clicked_items).reshape(-1).tolist()) | set(np.asarray(next_item_labels).reshape(-1).tolist())
        clicked_items_nonzero.discard(0)
        st.increment_current_step()
        st.update_items_first_click_step(clicked_items_nonzero)
        predicted_top_item_ids = np.asarray(predicted_item_ids)[:, :, :self.eval_metrics_top_n]
        st.get_cold_start_state().update_items_num_steps_before_first_rec(predicted_top_item_ids, st.items_first_click_step,
                                                                          st.get_current_step())

    def end(self, session=None):
        if self.mode == ModeKeys.EVAL:                                         # :1669-1695
            self.eval_streaming_metrics_last['clicks_count'] = int(np.sum([x['batch_items_count'] for x in self.stats_logs]))
            self.eval_streaming_metrics_last['sessions_count'] = int(np.sum([x['batch_sessions_count'] for x in self.stats_logs]))
            if self.eval_cold_start:                                           # add_cold_start_stats, :1662-1666
                self.eval_streaming_metrics_last['coldstart_chameleon'] = self.clicked_items_state.get_cold_start_state().get_statistics()
            self.eval_sessions_metrics_log.append(self.eval_streaming_metrics_last)
            self.clicked_items_state.restore_state_checkpoint()
