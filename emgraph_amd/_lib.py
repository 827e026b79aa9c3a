"""ctypes binding of libemgraph_hip.so (include/emgraph_hip.h).

The HIP library IS the product's compute path.  There is no CPU fallback: if the shared object is
missing or a call fails this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMGRAPH_HIP_LIB") or os.path.join(_HERE, "lib", "libemgraph_hip.so")

# ---- constants mirrored from include/emgraph_hip.h -------------------------------------------
ABI_VERSION = 8
TRANSE_L1, TRANSE_L2, DISTMULT, COMPLEX, HOLE, TRANSE_P = range(6)
SIDE_S, SIDE_O, SIDE_SO = range(3)
LOSS_PAIRWISE, LOSS_NLL, LOSS_ABSOLUTE_MARGIN, LOSS_SELF_ADVERSARIAL, LOSS_MULTICLASS_NLL = range(5)
OPT_SGD, OPT_MOMENTUM, OPT_ADAGRAD, OPT_ADAM, OPT_ADAM_LAZY = range(5)
SCORE_FINAL, SCORE_PARTIAL = 0, 1
EVAL_S, EVAL_O, EVAL_SPO, EVAL_S_O = range(4)

LOSS_IDS = {"pairwise": LOSS_PAIRWISE, "nll": LOSS_NLL, "absolute_margin": LOSS_ABSOLUTE_MARGIN,
            "self_adversarial": LOSS_SELF_ADVERSARIAL, "multiclass_nll": LOSS_MULTICLASS_NLL}
OPT_IDS = {"sgd": OPT_SGD, "momentum": OPT_MOMENTUM, "adagrad": OPT_ADAGRAD, "adam": OPT_ADAM,
           "adam_lazy": OPT_ADAM_LAZY}
SIDE_IDS = {"s": SIDE_S, "o": SIDE_O, "s+o": SIDE_SO, "s,o": SIDE_SO}
EVAL_SIDE_IDS = {"s": EVAL_S, "o": EVAL_O, "s+o": EVAL_SPO, "s,o": EVAL_S_O}


class EmgError(RuntimeError):
    """A libemgraph_hip call returned a non-zero code."""


_p = C.c_void_p
_i32, _i64, _u64, _f32 = C.c_int32, C.c_int64, C.c_uint64, C.c_float
_int = C.c_int

# name -> (restype, argtypes); every symbol include/emgraph_hip.h declares
SIGNATURES = {
    "emg_version": (_int, []),
    "emg_last_error": (C.c_char_p, []),
    "emg_target": (C.c_char_p, []),
    "emg_source_hash": (C.c_char_p, []),
    "emg_score_triples": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _i32, _f32, _p, _i64, _i32, _p, _p]),
    "emg_finalize_scores": (_int, [_int, _f32, _p, _i64, _p]),
    "emg_corrupt_codes": (_int, [_i64, _i32, _int, _i64, _p, _u64, _u64, _p, _p, _p, _p]),
    "emg_corrupt_expand": (_int, [_p, _i64, _i32, _p, _p, _p]),
    "emg_train_forward": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _i32, _f32, _p, _i64, _i32, _p, _i32, _p, _p, _p]),
    "emg_loss": (_int, [_int, _p, _p, _i64, _i32, _i32, _f32, _f32, _p, _p, _p, _p]),
    "emg_train_backward": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _i32, _f32, _p, _i64, _i32, _p, _p, _p,
                                  _p, _p, _i64, _p, _p, _p]),
    "emg_apply_workspace_bytes": (_i64, [_i64, _i64]),
    "emg_apply_workspace_bytes_ex": (_i64, [_i64, _i64, _i32]),
    "emg_apply_rows": (_int, [_int, _p, _i64, _i64, _i32, _p, _p, _p, _i32, _p, _i64, _p, _i64,
                              C.POINTER(_f32), _p, _i64, _p]),
    "emg_lp_regularizer": (_int, [_p, _i64, _i64, _i32, _f32, _i32, _f32, _p, _p]),
    "emg_lp_grad_rows": (_int, [_p, _i64, _i64, _i32, _f32, _i32, _p, _i64, _p, _p, _p]),
    "emg_clip_rows": (_int, [_p, _i64, _i64, _i32, _f32, _p]),
    "emg_scatter_rows": (_int, [_p, _i64, _i64, _i32, _p, _i64, _p, _i64, _p]),
    "emg_eval_build_queries": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _i32, _f32, _p, _i64, _int, _p, _i64,
                                      _p, _p]),
    "emg_eval_count": (_int, [_int, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _i32, _f32, _int, _p, _i64, _p, _p, _p]),
    "emg_eval_filter_count": (_int, [_int, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i32, _f32, _int, _p, _p,
                                     _p, _p, _p]),
    "emg_eval_scores_dense": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _p, _i32, _f32, _int, _p, _i64, _p,
                                     _i64, _p]),
    "emg_to_bf16": (_int, [_p, _i64, _i64, _i32, _p, _i64, _p]),
    "emg_to_f16": (_int, [_p, _i64, _i64, _i32, _p, _i64, _p]),
}

class BackwardArgs(C.Structure):
    """mirror of `emg_backward_args` (include/emgraph_hip.h)"""
    _fields_ = [
        ("model", _i32), ("k_int", _i32), ("scale", _f32), ("eta", _i32),
        ("ent", _p), ("n_ent", _i64), ("ld_ent", _i64),
        ("rel", _p), ("n_rel", _i64), ("ld_rel", _i64),
        ("pos", _p), ("B", _i64), ("codes", _p),
        ("fused_loss", _i32), ("margin", _f32), ("loss_accum", _p),
        ("g_pos", _p), ("g_neg", _p),
        ("bw_scores_pos", _p), ("bw_scores_neg", _p),
        ("scores_pos_out", _p), ("scores_neg_out", _p),
        ("contrib_ent", _p), ("contrib_rel", _p), ("ldc", _i64),
        ("single_ent", _p), ("opt", _i32), ("step", _i32), ("hyper", _f32 * 8),
        ("ent_state0", _p), ("ent_state1", _p), ("tag_ent", _p),
        ("fac_ws_ent", _p), ("fac_ws_ent_bytes", _i64),
        ("layout_B", _i64), ("ctl", _p),
        ("lp_accum", _p),
        ("lr_hist", _p),
        ("inplace_window", _i32), ("loss_slots", _i32),
    ]


SIGNATURES.update({
    "emg_build_dest": (_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    "emg_train_backward_ex": (_int, [C.POINTER(BackwardArgs), _p]),
    "emg_group_dest": (_int, [_p, _i64, _i64, _p, _i64, _p, _p]),
    "emg_group_dest_keyed": (_int, [_p, _p, _i64, _i64, _p, _i64, _p]),
    "emg_apply_grouped": (_int, [_int, _p, _i64, _i64, _i32, _p, _p, _p, _i32, _p, _i64, _i64, _i32,
                                 C.POINTER(_f32), _p, _p, _i64, _p]),
    "emg_apply_grouped_factored": (_int, [_int, _p, _i64, _i64, _i32, _p, _p, _p, _i32, _p, _i64, _i64, _i32,
                                          C.POINTER(_f32), _p, _p, _i64, _p]),
})

class PrepareArgs(C.Structure):
    """mirror of `emg_prepare_args` (include/emgraph_hip.h)"""
    _fields_ = [
        ("pos", _p), ("B", _i64), ("eta", _i32), ("n_sides", _i32), ("sides", _i32 * 4),
        ("n_choices", _i64), ("entities_list", _p), ("seed", _u64), ("draw_counter0", _u64),
        ("inj_mask", _p), ("inj_repl", _p),
        ("codes", _p),
        ("dest_ent", _p), ("n_extra_ent", _i64), ("n_ent", _i64),
        ("dest_rel", _p), ("n_extra_rel", _i64), ("n_rel", _i64),
        ("ws_ent", _p), ("ws_ent_bytes", _i64), ("ws_rel", _p), ("ws_rel_bytes", _i64),
        ("single_flags", _p),
        ("B_global", _i64), ("row_offset", _i64),
        ("factored", _i32), ("ws_clean", _i32),
        ("layout_B", _i64), ("ctl", _p),
    ]


SIGNATURES.update({"emg_prepare_batch": (_int, [C.POINTER(PrepareArgs), _p])})

SIGNATURES.update({
    "emg_eval_pos_int_bf16": (_int, [_int, _p, _i64, _i32, _f32, _p, _i64, _int, _p, _i64, _p, _p, _p]),
    "emg_eval_count_bf16": (_int, [_int, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _p, _i64, _i32, _f32, _p, _p, _i32, _p]),
    "emg_eval_filter_count_bf16": (_int, [_int, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i32, _f32, _p, _p,
                                          _p, _p, _p]),
    "emg_eval_scores_dense_bf16": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _p, _i32, _f32, _p, _i64, _p]),
    "emg_eval_prefilter_bounds": (_int, [_p, _i64, _i64, _p, _i64, _i32, _p, _p]),
    "emg_eval_prefilter_band": (_int, [_p, _i64, _i64, _p, _i64, _i32, _p, _p, _p]),
    "emg_eval_prefilter_f16": (_int, [_int, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i32, _f32, _p, _p, _p, _i64, _p]),
    "emg_eval_prefilter_f16_ties": (_int, [_int, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i32, _f32, _p, _p, _p, _p, _i64, _p]),
    "emg_eval_prefilter_segments": (_i64, [_i64, _i64]),
    "emg_eval_prefilter_segments_k": (_i64, [_i64, _i64, _i32]),
    "emg_eval_prefilter_waves": (_i32, [_i32]),
    "emg_eval_prefilter_max_cols": (_i32, []),
    "emg_eval_rescore_pairs_rows": (_int, [_int, _p, _i64, _p, _p, _i64, _i64, _i32, _f32, _p, _i64, _p, _i64, _i32, _i32, _p, _p, _p]),
    "emg_eval_prefilter_ld": (_i64, [_i32]),
    "emg_eval_rescore_tiles_ws_bytes": (_i64, [_i64]),
    "emg_eval_rescore_pairs_tiles": (_int, [_int, _p, _i64, _p, _p, _i64, _i64, _i64, _i32, _f32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p]),
    "emg_eval_rescore_pairs": (_int, [_int, _p, _i64, _p, _p, _i64, _i64, _i32, _f32, _p, _i64, _p, _i64, _p, _p, _p]),
    "emg_eval_rescore_pairs_ex": (_int, [_int, _p, _i64, _p, _p, _i64, _i64, _i32, _f32, _p, _i64, _p, _i64, _i32, _p, _p, _p]),
    "emg_to_f16_l2": (_int, [_p, _i64, _i64, _i32, _int, _p, _i64, _p, _p, _p]),
    "emg_eval_l2_thresholds": (_int, [_p, _i64, _i64, _p, _p, _p, _i32, _p, _p]),
    "emg_eval_prefilter_f16_thr": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i32, _p, _p, _p, _i64, _p]),
    "emg_eval_sad_range": (_int, [_p, _i64, _i64, _p, _i64, _i64, _i32, _p, _p]),
    "emg_eval_sad_ld": (_i64, [_i32]),
    "emg_eval_sad_quantize": (_int, [_p, _i64, _i64, _i32, _p, _p, _i64, _p]),
    "emg_eval_sad_thresholds": (_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    "emg_eval_sad_segments": (_i64, [_i64, _i64]),
    "emg_eval_prefilter_sad": (_int, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i32, _p, _p, _p, _i64, _p]),
})

class StepArgs(C.Structure):
    """mirror of `emg_step_args` (include/emgraph_hip.h)"""
    _fields_ = [
        ("model", _i32), ("k_int", _i32), ("scale", _f32), ("eta", _i32), ("n_sides", _i32), ("sides", _i32 * 4),
        ("ent", _p), ("n_ent", _i64), ("ld_ent", _i64), ("rel", _p), ("n_rel", _i64), ("ld_rel", _i64),
        ("ent_state0", _p), ("ent_state1", _p), ("rel_state0", _p), ("rel_state1", _p),
        ("tag_ent", _p), ("tag_rel", _p),
        ("opt", _i32), ("step", _i32), ("hyper", _f32 * 8),
        ("pos", _p), ("B", _i64),
        ("n_choices", _i64), ("entities_list", _p), ("seed", _u64), ("draw_counter0", _u64),
        ("inj_mask", _p), ("inj_repl", _p),
        ("loss", _i32), ("margin", _f32), ("alpha", _f32), ("loss_accum", _p),
        ("inplace", _i32),
        ("workspace", _p), ("workspace_bytes", _i64),
    ]


SIGNATURES.update({
    "emg_corrupt_fit": (_int, [_p, _i64, _i32, _int, _i64, _p, _i64, _u64, _u64, _p, _p]),
    "emg_rank_1vsall": (_int, [_int, _p, _i64, _i64, _p, _i64, _i64, _i32, _f32, _p, _i64, _int, _p, _i64, _p, _p, _int,
                               _int, _p, _p]),
    "emg_train_step_workspace_bytes": (_i64, [_i64, _i32, _i32, _i64, _i64]),
    "emg_train_step": (_int, [C.POINTER(StepArgs), _p]),
})

class ApplyArgs(C.Structure):
    """mirror of `emg_apply_args`"""
    _fields_ = [
        ("opt", _i32), ("k_int", _i32), ("table", _p), ("n_rows", _i64), ("ld", _i64),
        ("state0", _p), ("state1", _p), ("tag", _p), ("step", _i32), ("skip_single", _i32),
        ("contrib", _p), ("ldc", _i64), ("n_contrib", _i64),
        ("hyper", _f32 * 8), ("lp_accum", _p), ("workspace", _p), ("workspace_bytes", _i64),
        ("factored", _i32), ("table_index", _i32),
        ("layout_n", _i64), ("ctl", _p),
        ("deferred_dense", _i32), ("reserved1", _i32),
    ]


SIGNATURES.update({
    "emg_apply_grouped_ex": (_int, [C.POINTER(ApplyArgs), _p]),
    "emg_apply_grouped_pair": (_int, [C.POINTER(ApplyArgs), C.POINTER(ApplyArgs), _p]),
    "emg_deferred_catchup": (_int, [_int, _p, _i64, _i64, _i32, _p, _p, _p, C.POINTER(_f32), _p, _i32, _p, _p, _i64, _i64, _i32, _i64, _p]),
    "emg_deferred_materialize": (_int, [_int, _p, _i64, _i64, _i32, _p, _p, _p, C.POINTER(_f32), _p, _i32, _p, _p]),
})


class PlanSlot(C.Structure):
    """mirror of `emg_plan_slot`"""
    _fields_ = [("codes", _p), ("dest_ent", _p), ("dest_rel", _p), ("single", _p),
                ("ws_ent", _p), ("ws_ent_bytes", _i64), ("ws_rel", _p), ("ws_rel_bytes", _i64)]


class PlanConfig(C.Structure):
    """mirror of `emg_plan_config`"""
    _fields_ = [
        ("model", _i32), ("k_int", _i32), ("scale", _f32), ("eta", _i32), ("n_sides", _i32), ("sides", _i32 * 4),
        ("ent", _p), ("n_ent", _i64), ("ld_ent", _i64), ("rel", _p), ("n_rel", _i64), ("ld_rel", _i64),
        ("ent_state0", _p), ("ent_state1", _p), ("rel_state0", _p), ("rel_state1", _p), ("tag_ent", _p), ("tag_rel", _p),
        ("opt", _i32), ("loss", _i32), ("margin", _f32), ("alpha", _f32),
        ("seed", _u64), ("batches_count", _i64),
        ("X", _p), ("n_triples", _i64),
        ("cap_B", _i64),
        ("scores", _p), ("g", _p), ("contrib_ent", _p), ("contrib_rel", _p), ("ldc", _i64),
        ("loss_accum", _p), ("lp_sum", _p),
        ("factored", _i32), ("loss_slots", _i32),
        ("lp_lambda_ent", _f32), ("lp_lambda_rel", _f32), ("lp_p", _i32),
        ("fused", _i32), ("inplace", _i32), ("normalize", _i32),
        ("n_slots", _i32), ("slots", PlanSlot * 4),
        ("aux_min_rows", _i64),
        ("ctl_buf", _p), ("ctl_bytes", _i64),
        ("lr_t_hist", _p),
    ]


class PlanBatch(C.Structure):
    """mirror of `emg_plan_batch`"""
    _fields_ = [("start", _i64), ("B", _i64), ("epoch", _i32), ("batch", _i32),
                ("n_choices", _i64), ("entities_list", _p), ("inj_mask", _p), ("inj_repl", _p)]


SIGNATURES.update({
    "emg_plan_create": (_int, [C.POINTER(PlanConfig), C.POINTER(_p)]),
    "emg_plan_step": (_int, [_p, C.POINTER(PlanBatch), _i32, C.POINTER(_f32), C.POINTER(PlanBatch), _i32, _p]),
    "emg_plan_graph_ok": (_int, [_p]),
    "emg_plan_deferred_ok": (_int, [_i64, _i32, _i64, _i64]),
    "emg_plan_run": (_int, [_p, C.POINTER(PlanBatch), _i32, _i32, C.POINTER(_f32), _p]),
    "emg_plan_timing": (_int, [_p, _i32]),
    "emg_plan_stage_ms": (_int, [_p, C.POINTER(_f32), C.POINTER(_i32)]),
    "emg_plan_destroy": (_int, [_p]),
    "emg_init_table": (_int, [_int, _p, _i64, _i64, _i32, _f32, _f32, _u64, _u64, _p]),
})

_lib = None


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmgError(
            "libemgraph_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or emgraph_amd/csrc/build.sh.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.emg_version()
    if ver != ABI_VERSION:
        raise EmgError("libemgraph_hip ABI version %d != expected %d" % (ver, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().emg_last_error().decode("utf-8", "replace")
        raise EmgError("%s failed (code %d): %s" % (what or "libemgraph_hip call", rc, msg))
