"""ctypes binding of libchameleon_nar.so (include/chameleon_nar.h).  No CPU fallback: if the library is
missing or a kernel call fails this raises - the product path never silently degrades."""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int32, c_int64, c_long, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libchameleon_nar.so")

P = c_void_p

_SIGNATURES = {
    "cham_neg_sample_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cham_neg_sample": (c_int, [P, c_int, c_int, P, c_int, c_uint32, c_uint32, c_int, c_int, c_int, c_int, P, P, P, P, P,
                                P, c_size_t, P]),
    "cham_ctx_assemble": (c_int, [P, P, c_int, P, c_int, P, P, P, P, P, P]),
    "cham_set_log_bases": (c_int, [c_float, c_float]),
    "cham_step_ints": (c_int, [P, P, P, P, c_int64, c_int, c_int, P, c_int, P, P, P, P, P, P]),
    "cham_item_dynamic_raw": (c_int, [P, P, c_int, P, P, P, P, P]),
    "cham_norm_stats_from_recent": (c_int, [P, c_int, c_int64, P, P, P, P, P]),
    "cham_norm_stats_from_buffer": (c_int, [P, c_int, c_int64, P, P, P, P, P]),
    "cham_norm_stats_from_rows": (c_int, [P, P, P, c_int, P, P]),
    "cham_row_weights": (c_int, [P, c_int, P, c_size_t, c_int, P, P, P, P]),
    "cham_item_assemble": (c_int, [P, c_int, c_int, c_int, P, c_int, P, c_int, P, P, P, P, c_int, P, P, P, P, P, P]),
    "cham_item_assemble_lds": (c_int, [P, c_int, c_int, c_int, P, c_int, P, c_int, P, P, P, P, c_int, P, c_int, P, c_int, P, P, P, P, P, P]),
    "cham_feature_bwd": (c_int, [P, P, c_int, c_int, P, P, P]),
    "cham_feature_bwd_workspace_bytes": (c_size_t, [c_int]),
    "cham_feature_bwd_ws": (c_int, [P, P, c_int, c_int, P, P, P, c_size_t, P]),
    "cham_emb_grad_scan": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P]),
    "cham_group_rows_workspace_bytes": (c_size_t, [c_int]),
    "cham_group_rows_segments_len": (c_size_t, [c_int]),
    "cham_group_rows": (c_int, [P, c_int, c_int, P, P, P, c_size_t, P]),
    "cham_emb_grad_grouped": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "cham_dropout": (c_int, [P, P, c_long, c_int, c_int, c_float, c_uint32, c_uint32, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, P]),
    "cham_dense_rows": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, P, P, P]),
    "cham_gemm_f32": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int,
                              P, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    "cham_gemm_bf16": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int,
                               P, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    "cham_gemm_f32x3": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int,
                                P, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    "cham_gemm_f32x2h": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int,
                                 P, c_int, c_int, c_int, P, c_size_t, c_int, P, P, P]),
    "cham_gemm_f32x3_set_variant": (None, [c_int]),
    "cham_gemm_f32x3_launch_counts": (None, [P, c_int]),
    "cham_gemm_p3": (c_int, [P, c_int64, c_int, P, c_int64, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int,
                             P, c_size_t, c_int, P]),
    "cham_gemm_b16_dma": (c_int, [P, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    "cham_gemm_p3_launch_counts": (None, [P, c_int]),
    "cham_gemm_b16_dma_set_nt_wide": (c_int, [c_int]),
    "cham_split3": (c_int, [P, c_int, c_int, c_int, P, c_int64, c_int, P, c_int64, c_int, P]),
    "cham_combine_fwd_p3": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_int64, P]),
    "cham_mulpred_bwd_p3": (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_int64, P, P]),
    "cham_dm_mulpred_p3": (c_int, [P, c_int, c_int, P, c_int64, P, P, c_int, c_int, c_int, P, c_int64, P, P, P]),
    "cham_h2_scale_absmax": (c_int, [P, c_size_t, P, c_size_t, P, P]),
    "cham_h2_scale_rownorm": (c_int, [P, c_long, c_int, c_int, P, P, P]),
    "cham_h2_scale_rownorm2": (c_int, [P, c_long, c_int, c_int, P, P, P, P]),
    "cham_split2h": (c_int, [P, c_int, c_int, c_int, P, c_int64, c_int, P, c_int64, c_int, P, c_int, P]),
    "cham_gemm_h2": (c_int, [P, c_int64, c_int, P, P, c_int64, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int,
                             P, c_size_t, c_int, P]),
    "cham_h2b_block_elements": (c_int, []),
    "cham_gemm_h2b": (c_int, [P, c_int64, c_int, P, P, c_int64, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int,
                              P, c_size_t, c_int, c_int, c_int, c_int, P]),
    "cham_combine_fwd_h2b": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_int64, P, c_int, P]),
    "cham_dm_mulpred_h2_blk": (c_int, [P, c_int, c_int, P, c_int64, P, P, P, P, c_int, c_int, c_int, P, c_int64, P, P, P, P]),
    "cham_gemm_h2_launch_counts": (None, [P, c_int]),
    "cham_gemm_h2_set_nt_wide": (c_int, [c_int]),
    "cham_combine_fwd_h2": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_int64, P, P]),
    "cham_mulpred_bwd_h2": (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_int64, P, P, P]),
    "cham_dm_mulpred_h2": (c_int, [P, c_int, c_int, P, c_int64, P, P, c_int, c_int, c_int, P, c_int64, P, P, P, P]),
    "cham_dm_mulpred_h2h": (c_int, [P, c_int, c_int, P, c_int64, P, P, P, P, c_int, c_int, c_int, P, c_int64, P, P, P, P]),
    "cham_dm_mulpred_b16": (c_int, [P, c_int, c_int, P, P, P, c_int, c_int, c_int, P, P, P, P]),
    "cham_gemm_b16": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int,
                              P, c_size_t, c_int, P]),
    "cham_gemm_b16_set_variant": (None, [c_int]),
    "cham_gemm_b16_launch_counts": (None, [P, c_int]),
    "cham_gemm_set_variant": (None, [c_int]),
    "cham_gemm_launch_counts": (None, [P, c_int]),
    "cham_combine_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_long, c_long, P]),
    "cham_combine_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cham_combine_bwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "cham_combine_bwd_gs": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P, c_size_t, P]),
    "cham_gemm_h2_groupsum_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cham_gemm_h2_dgrad_gs": (c_int, [P, c_int64, c_int, P, P, c_int64, c_int, P, P, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, c_int, P,
                                      c_size_t, P]),
    "cham_combine_fwd_b16": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "cham_combine_bwd_b16": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "cham_mulpred_bwd_b16": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "cham_mul_rows_b16": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "cham_score_softmax_fwd_b16": (c_int, [P, c_int, P, P, c_int, c_int, c_float, P, P, P, P, c_float, P, P, P, P]),
    "cham_score_softmax_bwd_b16": (c_int, [P, c_int, P, P, P, c_int, c_int, c_float, c_float, P, P, c_float, P, P, P, P, P]),
    "cham_colsum_b16": (c_int, [P, c_int, c_int, c_int, P, P, c_int, P, c_size_t, P]),
    "cham_cast_b16": (c_int, [P, c_int, c_int, P, P, P]),
    "cham_upcast_b16": (c_int, [P, c_size_t, P, P]),
    "cham_rnn_fwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "cham_rnn_bwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P]),
    "cham_rnn_coop_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cham_ugrnn_fwd_coop": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "cham_ugrnn_bwd_coop": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "cham_rnn_coop_timeouts": (c_int, [P, c_int, c_int, P]),
    "cham_ugrnn_point_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "cham_ugrnn_point_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "cham_transpose_f32": (c_int, [P, c_int, c_int, P, P]),
    "cham_rows_gather": (c_int, [P, P, c_long, c_int, P, P]),
    "cham_rows_scatter": (c_int, [P, P, c_long, c_int, P, P]),
    "cham_mulpred_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "cham_score_softmax_fwd": (c_int, [P, c_int, P, P, c_int, c_int, c_float, P, P, P, P, c_float, P, P, P, P]),
    "cham_score_softmax_bwd": (c_int, [P, c_int, P, P, P, c_int, c_int, c_float, c_float, P, P, c_float, P, P, P, P, P]),
    "cham_rank_items": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P]),
    "cham_state_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cham_state_update": (c_int, [P, P, c_int, c_int, c_double, P, P, c_int, P, P, P, c_int, c_int, P, P, c_size_t, P]),
    "cham_sumsq_partial": (c_int, [P, c_size_t, P, P]),
    "cham_loss_finalize": (c_int, [P, c_int, c_float, P, c_float, P, P]),
    "cham_adam_tf": (c_int, [P, P, P, P, c_size_t, c_size_t, c_float, c_float, c_float, c_float, c_float, P]),
    "cham_step_scalars_bytes": (c_int, []),
    "cham_step_scalars_set": (c_int, [P, c_uint32, c_uint32, c_int64, c_float, c_float, c_int, P]),
    "cham_neg_sample_dev": (c_int, [P, c_int, c_int, P, c_int, c_uint32, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_size_t, P]),
    "cham_step_ints_dev": (c_int, [P, P, P, P, P, c_int, c_int, P, c_int, P, P, P, P, P, P]),
    "cham_norm_stats_from_buffer_dev": (c_int, [P, c_int, P, P, P, P, P, P]),
    "cham_score_softmax_bwd_dev": (c_int, [P, c_int, P, P, P, c_int, c_int, c_float, P, P, P, c_float, P, P, P, P, P]),
    "cham_score_softmax_bwd_b16_dev": (c_int, [P, c_int, P, P, P, c_int, c_int, c_float, P, P, P, c_float, P, P, P, P, P]),
    "cham_loss_finalize_dev": (c_int, [P, c_int, P, P, c_float, P, P]),
    "cham_adam_tf_dev": (c_int, [P, P, P, P, c_size_t, c_size_t, c_float, P, c_float, c_float, c_float, P]),
    "cham_accumulate": (c_int, [P, P, c_size_t, c_int, P]),
    "cham_loss_accumulate": (c_int, [P, P, c_int, P]),
    "cham_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cham_colsum": (c_int, [P, c_int, c_int, c_int, P, P, c_int, P, c_size_t, P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


class ChameleonLibError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the HIP library; raises ChameleonLibError when it has not been built (python -m chameleon_recsys_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ChameleonLibError(
            "HIP extension %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    # library-wide kernel switches, set ONCE per process at load (not per runtime: a second NARRuntime must not flip the kernels of the
    # first - ADVICE r05).  CHAM_H2_NT_WIDE=0: the 32-byte-piece NT kernels of round 4 (bit-identical results; A/B arm)
    lib.cham_gemm_h2_set_nt_wide(1 if os.environ.get("CHAM_H2_NT_WIDE", "1") == "1" else 0)
    # CHAM_B16_NT_WIDE=0: the bf16 configuration's NT CAR GEMMs on gemm_b1_kernel (32-byte pieces) instead of gemm_b1w_kernel (round 6; A/B arm)
    lib.cham_gemm_b16_dma_set_nt_wide(1 if os.environ.get("CHAM_B16_NT_WIDE", "1") == "1" else 0)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise ChameleonLibError("%s failed with code %d" % (what, rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
