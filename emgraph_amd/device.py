"""Thin typed wrappers: PyTorch-ROCm tensors (device-memory holders + stream provider) -> C-ABI calls.

No arithmetic happens here; every function forwards pointers and sizes to libemgraph_hip.so on the
current torch stream.  Tables are 2-D float32 tensors with unit column stride; their row stride is
passed as ``ld``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def require_gpu():
    if not torch.cuda.is_available():
        raise L.EmgError("emgraph_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                         "there is no CPU fallback.")
    L.load()


_forced = None  # (torch stream object, c_void_p handle) while inside `on_stream`


def _stream():
    if _forced is not None:
        return _forced[1]
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class on_stream:
    """Route the library calls made inside the block to ``stream`` WITHOUT touching torch's current stream:
    `torch.cuda.current_stream()` / `with torch.cuda.stream(...)` cost ~8 us each and a training step makes a
    dozen of them.  Only for calls whose outputs are preallocated (torch allocations still follow torch's stream)."""

    def __init__(self, stream):
        self._new = (stream, C.c_void_p(stream.cuda_stream))

    def __enter__(self):
        global _forced
        self._prev, _forced = _forced, self._new
        return self._new[0]

    def __exit__(self, *exc):
        global _forced
        _forced = self._prev
        return False


def active_stream():
    """the stream library calls currently go to (torch stream object)"""
    return _forced[0] if _forced is not None else torch.cuda.current_stream()


def _chk_table(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError("%s must be a 2-D float32 CUDA tensor with unit column stride" % name)
    return t.data_ptr(), t.shape[0], t.stride(0)


def _hyper8(hyper):
    """{lr, momentum, beta1, beta2, eps, lr_t, lp_lambda, lp_p}: six-value tuples mean 'no folded LP regulariser'"""
    h = [float(x) for x in hyper] + [0.0] * (8 - len(hyper))
    return (C.c_float * 8)(*h)


def _chk_vec(t, dtype, name, n=None):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError("%s must be a contiguous %s CUDA tensor" % (name, dtype))
    if n is not None and t.numel() != n:
        raise ValueError("%s has %d elements, expected %d" % (name, t.numel(), n))
    return t.data_ptr()


def score_triples(model_id, ent, rel, k_int, scale, spo, flags=L.SCORE_FINAL, out=None):
    lib = L.load()
    pe, ne, lde = _chk_table(ent, "ent")
    pr, nr, ldr = _chk_table(rel, "rel")
    n = spo.shape[0]
    ps = _chk_vec(spo, torch.int32, "spo", 3 * n)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=ent.device)
    L.check(lib.emg_score_triples(model_id, pe, ne, lde, pr, nr, ldr, k_int, scale, ps, n, flags,
                                  _chk_vec(out, torch.float32, "out", n), _stream()), "emg_score_triples")
    return out


def finalize_scores(model_id, scale, scores):
    lib = L.load()
    L.check(lib.emg_finalize_scores(model_id, scale, _chk_vec(scores, torch.float32, "scores"), scores.numel(),
                                    _stream()), "emg_finalize_scores")
    return scores


def corrupt_codes(B, eta, side, n_choices, device, entities_list=None, seed=0, counter=0, inj_mask=None,
                  inj_repl=None, out=None):
    lib = L.load()
    n = B * eta
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=device)
    L.check(lib.emg_corrupt_codes(B, eta, side, n_choices, _chk_vec(entities_list, torch.int32, "entities_list"),
                                  seed & 0xFFFFFFFFFFFFFFFF, counter & 0xFFFFFFFFFFFFFFFF,
                                  _chk_vec(inj_mask, torch.int32, "inj_mask", n if inj_mask is not None else None),
                                  _chk_vec(inj_repl, torch.int32, "inj_repl", n if inj_repl is not None else None),
                                  _chk_vec(out, torch.int32, "codes", n), _stream()), "emg_corrupt_codes")
    return out


def corrupt_expand(pos, eta, codes):
    lib = L.load()
    B = pos.shape[0]
    out = torch.empty((B * eta, 3), dtype=torch.int32, device=pos.device)
    L.check(lib.emg_corrupt_expand(_chk_vec(pos, torch.int32, "pos", 3 * B), B, eta,
                                   _chk_vec(codes, torch.int32, "codes", B * eta), out.data_ptr(), _stream()),
            "emg_corrupt_expand")
    return out


def train_forward(model_id, ent, rel, k_int, scale, pos, eta, codes, flags=L.SCORE_FINAL, scores_pos=None,
                  scores_neg=None):
    lib = L.load()
    pe, ne, lde = _chk_table(ent, "ent")
    pr, nr, ldr = _chk_table(rel, "rel")
    B = pos.shape[0]
    if scores_pos is None:
        scores_pos = torch.empty(B, dtype=torch.float32, device=ent.device)
    if scores_neg is None:
        scores_neg = torch.empty(B * eta, dtype=torch.float32, device=ent.device)
    L.check(lib.emg_train_forward(model_id, pe, ne, lde, pr, nr, ldr, k_int, scale,
                                  _chk_vec(pos, torch.int32, "pos", 3 * B), B, eta,
                                  _chk_vec(codes, torch.int32, "codes", B * eta) if eta else None, flags,
                                  _chk_vec(scores_pos, torch.float32, "scores_pos", B),
                                  _chk_vec(scores_neg, torch.float32, "scores_neg", B * eta) if eta else None,
                                  _stream()), "emg_train_forward")
    return scores_pos, scores_neg


def loss(loss_id, scores_pos, scores_neg, B, eta, n_sides, margin, alpha, loss_accum, g_pos=None, g_neg=None):
    lib = L.load()
    if g_pos is None:
        g_pos = torch.empty(B, dtype=torch.float32, device=scores_pos.device)
    if g_neg is None:
        g_neg = torch.empty(B * eta * n_sides, dtype=torch.float32, device=scores_pos.device)
    L.check(lib.emg_loss(loss_id, _chk_vec(scores_pos, torch.float32, "scores_pos", B),
                         _chk_vec(scores_neg, torch.float32, "scores_neg", B * eta * n_sides), B, eta, n_sides,
                         margin, alpha, _chk_vec(loss_accum, torch.float64, "loss_accum", 1),
                         _chk_vec(g_pos, torch.float32, "g_pos", B),
                         _chk_vec(g_neg, torch.float32, "g_neg", B * eta * n_sides), _stream()), "emg_loss")
    return g_pos, g_neg


def train_backward(model_id, ent, rel, k_int, scale, pos, eta, codes, g_pos, g_neg, contrib_ent, contrib_rel,
                   dest_ent, dest_rel):
    lib = L.load()
    pe, ne, lde = _chk_table(ent, "ent")
    pr, nr, ldr = _chk_table(rel, "rel")
    B = pos.shape[0]
    pce, nce, ldc = _chk_table(contrib_ent, "contrib_ent")
    pcr, ncr, ldc2 = _chk_table(contrib_rel, "contrib_rel")
    if ldc != ldc2 or nce < (2 + eta) * B or ncr < B:
        raise ValueError("contribution buffers have the wrong shape")
    L.check(lib.emg_train_backward(model_id, pe, ne, lde, pr, nr, ldr, k_int, scale,
                                   _chk_vec(pos, torch.int32, "pos", 3 * B), B, eta,
                                   _chk_vec(codes, torch.int32, "codes", B * eta) if eta else None,
                                   _chk_vec(g_pos, torch.float32, "g_pos", B),
                                   _chk_vec(g_neg, torch.float32, "g_neg", B * eta) if eta else None,
                                   pce, pcr, ldc, _chk_vec(dest_ent, torch.int32, "dest_ent", (2 + eta) * B),
                                   _chk_vec(dest_rel, torch.int32, "dest_rel", B), _stream()), "emg_train_backward")


def build_dest(pos, eta, codes, dest_ent, dest_rel):
    lib = L.load()
    B = pos.shape[0]
    L.check(lib.emg_build_dest(_chk_vec(pos, torch.int32, "pos", 3 * B), B, eta,
                               _chk_vec(codes, torch.int32, "codes", B * eta) if eta else None,
                               _chk_vec(dest_ent, torch.int32, "dest_ent", (2 + eta) * B),
                               _chk_vec(dest_rel, torch.int32, "dest_rel", B), _stream()), "emg_build_dest")


def train_backward_ex(model_id, ent, rel, k_int, scale, pos, eta, codes, contrib_ent, contrib_rel, fused_loss=-1,
                      margin=1.0, loss_accum=None, g_pos=None, g_neg=None, bw_scores_pos=None, bw_scores_neg=None,
                      scores_pos_out=None, scores_neg_out=None, single_ent=None, opt_id=0, step=0, hyper=None,
                      ent_state0=None, ent_state1=None, tag_ent=None, fac_ws_ent=None, lp_accum=None, loss_slots=0):
    """emg_train_backward_ex: fused (fused_loss>=0) or external-gradient backward, optional in-place
    singleton updates (single_ent flags from group_dest).  ``fac_ws_ent`` (bilinear models): FACTORED entity
    contributions — the entity workspace of ``prepare_batch(..., factored=True)`` for this batch; ``contrib_ent`` then
    holds 4*B rows (see include/emgraph_hip.h) and the entity apply is ``apply_grouped(..., factored=True)``.
    ``hyper`` of 8 values + ``lp_accum`` (plain SGD only): the LP regulariser folded into the in-place updates.
    ``loss_slots`` (a power of two > 1): ``loss_accum`` holds that many doubles, the loss is their sum."""
    lib = L.load()
    B = pos.shape[0]
    a = L.BackwardArgs()
    a.model, a.k_int, a.scale, a.eta = model_id, k_int, scale, eta
    a.ent, a.n_ent, a.ld_ent = _chk_table(ent, "ent")
    a.rel, a.n_rel, a.ld_rel = _chk_table(rel, "rel")
    a.pos, a.B = _chk_vec(pos, torch.int32, "pos", 3 * B), B
    a.codes = _chk_vec(codes, torch.int32, "codes", B * eta) if eta else None
    a.fused_loss, a.margin = fused_loss, margin
    a.loss_accum = _chk_vec(loss_accum, torch.float64, "loss_accum", max(1, int(loss_slots))) if loss_accum is not None else None
    a.loss_slots = int(loss_slots)
    a.g_pos = _chk_vec(g_pos, torch.float32, "g_pos", B)
    a.g_neg = _chk_vec(g_neg, torch.float32, "g_neg", B * eta) if eta else None
    a.bw_scores_pos = _chk_vec(bw_scores_pos, torch.float32, "bw_scores_pos", B)
    a.bw_scores_neg = _chk_vec(bw_scores_neg, torch.float32, "bw_scores_neg", B * eta if bw_scores_neg is not None else None)
    a.scores_pos_out = _chk_vec(scores_pos_out, torch.float32, "scores_pos_out", B)
    a.scores_neg_out = _chk_vec(scores_neg_out, torch.float32, "scores_neg_out", B * eta if scores_neg_out is not None else None)
    pce, nce, ldc = _chk_table(contrib_ent, "contrib_ent")
    pcr, ncr, ldc2 = _chk_table(contrib_rel, "contrib_rel")
    if ldc != ldc2 or nce < (4 if fac_ws_ent is not None else 2 + eta) * B or ncr < B:
        raise ValueError("contribution buffers have the wrong shape")
    a.contrib_ent, a.contrib_rel, a.ldc = pce, pcr, ldc
    a.single_ent = _chk_vec(single_ent, torch.uint8, "single_ent", (2 + eta) * B if single_ent is not None else None)
    a.opt, a.step = opt_id, step
    if hyper is not None:
        for i, v in enumerate(hyper):
            a.hyper[i] = float(v)
    if ent_state0 is not None:
        a.ent_state0 = _chk_table(ent_state0, "ent_state0")[0]
    if ent_state1 is not None:
        a.ent_state1 = _chk_table(ent_state1, "ent_state1")[0]
    a.tag_ent = _chk_vec(tag_ent, torch.int32, "tag_ent")
    if fac_ws_ent is not None:
        a.fac_ws_ent, a.fac_ws_ent_bytes = fac_ws_ent.data_ptr(), fac_ws_ent.numel() * fac_ws_ent.element_size()
    a.lp_accum = _chk_vec(lp_accum, torch.float64, "lp_accum", 1) if lp_accum is not None else None
    L.check(lib.emg_train_backward_ex(C.byref(a), _stream()), "emg_train_backward_ex")


def init_table(table, k_int, kind, a, b, seed, stream_id):
    """fill ``table[:, :k_int]`` in place: kind 'uniform' = U[a, b), 'normal' = N(mean a, std b) (emg_init_table)"""
    lib = L.load()
    pt, n, ld = _chk_table(table, "table")
    L.check(lib.emg_init_table({"uniform": 0, "normal": 1}[kind], pt, n, ld, k_int, float(a), float(b),
                               int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id), _stream()), "emg_init_table")
    return table


def group_dest(dest, n, n_rows, workspace, single_flags=None):
    lib = L.load()
    L.check(lib.emg_group_dest(_chk_vec(dest, torch.int32, "dest"), n, n_rows, workspace.data_ptr(),
                               workspace.numel() * workspace.element_size(),
                               _chk_vec(single_flags, torch.uint8, "single_flags"), _stream()), "emg_group_dest")


def group_dest_keyed(dest, order_key, n, n_rows, workspace):
    """stable grouping of ``dest`` with a destination's contributions in ascending ``order_key`` (int32 >= 0, distinct within a
    destination); apply with ``apply_grouped(..., factored=True)`` on the buffer the ids index"""
    lib = L.load()
    L.check(lib.emg_group_dest_keyed(_chk_vec(dest, torch.int32, "dest"), _chk_vec(order_key, torch.int32, "order_key"), n, n_rows,
                                     workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()), "emg_group_dest_keyed")


def prepare_batch(pos, eta, sides, n_choices, codes, dest_ent, dest_rel, n_ent, n_rel, ws_ent, ws_rel,
                  entities_list=None, seed=0, counter0=0, inj_mask=None, inj_repl=None, n_extra_ent=0, n_extra_rel=0,
                  single_flags=None, B_global=0, row_offset=0, factored=False):
    """codes of all corruption sides + destination ids + stable grouping (+ singleton flags) in ONE library call.
    ``B_global`` / ``row_offset``: ``pos`` is rows [row_offset, row_offset + B) of a larger (multi-GPU) batch and
    draws the negatives that batch would draw for those rows."""
    lib = L.load()
    B = pos.shape[0]
    n_neg = B * eta * len(sides)
    a = L.PrepareArgs()
    a.pos = _chk_vec(pos, torch.int32, "pos", 3 * B)
    a.B, a.eta, a.n_sides = B, eta, len(sides)
    for i, sd in enumerate(sides):
        a.sides[i] = sd
    a.n_choices = n_choices
    a.entities_list = _chk_vec(entities_list, torch.int32, "entities_list")
    a.seed, a.draw_counter0 = seed & 0xFFFFFFFFFFFFFFFF, counter0 & 0xFFFFFFFFFFFFFFFF
    a.inj_mask = _chk_vec(inj_mask, torch.int32, "inj_mask", n_neg if inj_mask is not None else None)
    a.inj_repl = _chk_vec(inj_repl, torch.int32, "inj_repl", n_neg if inj_repl is not None else None)
    a.codes = _chk_vec(codes, torch.int32, "codes", n_neg)
    a.dest_ent = _chk_vec(dest_ent, torch.int32, "dest_ent", n_extra_ent + 2 * B + n_neg)
    a.dest_rel = _chk_vec(dest_rel, torch.int32, "dest_rel", n_extra_rel + B)
    a.n_extra_ent, a.n_ent, a.n_extra_rel, a.n_rel = n_extra_ent, n_ent, n_extra_rel, n_rel
    a.ws_ent, a.ws_ent_bytes = ws_ent.data_ptr(), ws_ent.numel() * ws_ent.element_size()
    a.ws_rel, a.ws_rel_bytes = ws_rel.data_ptr(), ws_rel.numel() * ws_rel.element_size()
    a.single_flags = _chk_vec(single_flags, torch.uint8, "single_flags")
    a.B_global, a.row_offset = int(B_global), int(row_offset)
    a.factored = 1 if factored else 0
    L.check(lib.emg_prepare_batch(C.byref(a), _stream()), "emg_prepare_batch")


def apply_grouped(opt_id, table, k_int, state0, state1, tag, step, contrib, n_contrib, skip_single, hyper, workspace,
                  lp_accum=None, factored=False):
    """``hyper`` = (lr, momentum, beta1, beta2, eps, lr_t[, lp_lambda, lp_p]); with lp_lambda != 0 the LP regulariser's
    gradient is folded into every row's update and ``lp_accum`` (device double) receives sum |w|^p.
    ``factored``: the contributions were written by ``train_backward_ex(..., fac_ws_ent=workspace)``."""
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    p0 = _chk_table(state0, "state0")[0] if state0 is not None else None
    p1 = _chk_table(state1, "state1")[0] if state1 is not None else None
    pc, _, ldc = _chk_table(contrib, "contrib")
    h = _hyper8(hyper)
    lp = _chk_vec(lp_accum, torch.float64, "lp_accum", 1) if lp_accum is not None else None
    fn = lib.emg_apply_grouped_factored if factored else lib.emg_apply_grouped
    L.check(fn(opt_id, pt, nrows, ld, k_int, p0, p1, _chk_vec(tag, torch.int32, "tag"), step, pc, ldc, n_contrib,
               int(skip_single), h, lp, workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()),
            "emg_apply_grouped_factored" if factored else "emg_apply_grouped")


def _apply_args(opt_id, table, k_int, state0, state1, tag, step, contrib, n_contrib, skip_single, hyper, workspace,
                lp_accum=None, factored=False):
    a = L.ApplyArgs()
    a.opt, a.k_int = opt_id, k_int
    a.table, a.n_rows, a.ld = _chk_table(table, "table")
    a.state0 = _chk_table(state0, "state0")[0] if state0 is not None else None
    a.state1 = _chk_table(state1, "state1")[0] if state1 is not None else None
    a.tag = _chk_vec(tag, torch.int32, "tag")
    a.step, a.skip_single = step, int(skip_single)
    pc, _, ldc = _chk_table(contrib, "contrib")
    a.contrib, a.ldc, a.n_contrib = pc, ldc, n_contrib
    for i, v in enumerate(_hyper8(hyper)):
        a.hyper[i] = v
    a.lp_accum = _chk_vec(lp_accum, torch.float64, "lp_accum", 1) if lp_accum is not None else None
    a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    a.factored = 1 if factored else 0
    return a


def apply_grouped_pair(first, second):
    """two tables' applies (each a dict of apply_grouped's arguments) through shared launches (emg_apply_grouped_pair)"""
    a, b = _apply_args(**first), _apply_args(**second)
    b.table_index = 1
    L.check(L.load().emg_apply_grouped_pair(C.byref(a), C.byref(b), _stream()), "emg_apply_grouped_pair")


def apply_workspace_views(workspace, n_contrib):
    """(sorted destination ids, contribution indices) int32 views of a grouping workspace filled by group_dest /
    prepare_batch (layout of emg_group.hip::layout_impl: keys at byte 0, values at align256(4 n))"""
    kb = (4 * n_contrib + 255) // 256 * 256
    w32 = workspace.view(torch.int32)
    return w32[:n_contrib], w32[kb // 4:kb // 4 + n_contrib]


def apply_workspace_bytes(n_contrib, n_rows, k_int=0):
    """grouping workspace; with ``k_int`` it also holds the scratch of the long-segment reduction (emg_apply_long)"""
    lib = L.load()
    n = lib.emg_apply_workspace_bytes_ex(n_contrib, n_rows, k_int) if k_int else lib.emg_apply_workspace_bytes(n_contrib, n_rows)
    if n < 0:
        L.check(-1, "emg_apply_workspace_bytes")
    return int(n)


def apply_rows(opt_id, table, k_int, state0, state1, tag, step, contrib, dest, n_contrib, hyper, workspace):
    """hyper = (lr, momentum, beta1, beta2, eps, lr_t) python floats."""
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    p0 = _chk_table(state0, "state0")[0] if state0 is not None else None
    p1 = _chk_table(state1, "state1")[0] if state1 is not None else None
    pc, _, ldc = _chk_table(contrib, "contrib")
    h = _hyper8(hyper)
    L.check(lib.emg_apply_rows(opt_id, pt, nrows, ld, k_int, p0, p1, _chk_vec(tag, torch.int32, "tag"), step, pc, ldc,
                               _chk_vec(dest, torch.int32, "dest"), n_contrib, h, workspace.data_ptr(),
                               workspace.numel() * workspace.element_size(), _stream()), "emg_apply_rows")


def lp_regularizer(table, k_int, lam, p, grad_scale_lr, loss_accum):
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    L.check(lib.emg_lp_regularizer(pt, nrows, ld, k_int, lam, p, grad_scale_lr,
                                   _chk_vec(loss_accum, torch.float64, "loss_accum", 1) if loss_accum is not None else None,
                                   _stream()), "emg_lp_regularizer")


def lp_grad_rows(table, k_int, lam, p, contrib, dest, loss_accum):
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    pc, nc, ldc = _chk_table(contrib, "contrib")
    if nc < nrows:
        raise ValueError("contrib has fewer rows than the table")
    L.check(lib.emg_lp_grad_rows(pt, nrows, ld, k_int, lam, p, pc, ldc, _chk_vec(dest, torch.int32, "dest", nrows),
                                 _chk_vec(loss_accum, torch.float64, "loss_accum", 1), _stream()), "emg_lp_grad_rows")


def scatter_rows(table, k_int, rows, ids):
    """rows[j] -> table[ids[j]] for the ids inside the table (distinct; others skipped)"""
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    pr, n, ldr = _chk_table(rows, "rows")
    L.check(lib.emg_scatter_rows(pt, nrows, ld, k_int, pr, ldr, _chk_vec(ids, torch.int32, "ids", n), n, _stream()), "emg_scatter_rows")


def clip_rows(table, k_int, max_norm=1.0):
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    L.check(lib.emg_clip_rows(pt, nrows, ld, k_int, max_norm, _stream()), "emg_clip_rows")


def eval_build_queries(model_id, ent, rel, k_int, scale, test_spo, side_mode, ldq=None):
    lib = L.load()
    pe, ne, lde = _chk_table(ent, "ent")
    pr, nr, ldr = _chk_table(rel, "rel")
    n_q = test_spo.shape[0]
    n_rows = 2 * n_q if side_mode >= L.EVAL_SPO else n_q
    ldq = ldq or ((k_int + 3) // 4) * 4
    Q = torch.zeros((n_rows, ldq), dtype=torch.float32, device=ent.device)
    pos_int = torch.empty(n_rows, dtype=torch.int32, device=ent.device)
    L.check(lib.emg_eval_build_queries(model_id, pe, ne, lde, pr, nr, ldr, k_int, scale,
                                       _chk_vec(test_spo, torch.int32, "test_spo", 3 * n_q), n_q, side_mode,
                                       Q.data_ptr(), ldq, pos_int.data_ptr(), _stream()), "emg_eval_build_queries")
    return Q, pos_int


def eval_count(model_id, Q, pos_int, ent, k_int, scale, cnt_gt, cnt_eq, cand=None, n_cand=None, precision=0,
               ent_bf16=None):
    lib = L.load()
    pq, n_rows, ldq = _chk_table(Q, "Q")
    pe, ne, lde = _chk_table(ent, "ent")
    if cand is not None:
        n_cand = cand.numel()
    elif n_cand is None:
        n_cand = ne
    pb, ldb = (None, 0)
    if ent_bf16 is not None:
        pb, ldb = ent_bf16.data_ptr(), ent_bf16.stride(0)
    L.check(lib.emg_eval_count(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows), n_rows, pe,
                               n_cand, lde, _chk_vec(cand, torch.int32, "cand"), k_int, scale, precision, pb, ldb,
                               _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                               _chk_vec(cnt_eq, torch.int32, "cnt_eq", n_rows), _stream()), "emg_eval_count")


def eval_filter_count(model_id, Q, pos_int, ent, ent_offset, k_int, scale, filt_ptr, filt_idx, fcnt_gt, fcnt_eq,
                      precision=0):
    lib = L.load()
    pq, n_rows, ldq = _chk_table(Q, "Q")
    pe, ne, lde = _chk_table(ent, "ent")
    L.check(lib.emg_eval_filter_count(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows), n_rows, pe,
                                      ne, lde, ent_offset, k_int, scale, precision,
                                      _chk_vec(filt_ptr, torch.int64, "filt_ptr", n_rows + 1),
                                      _chk_vec(filt_idx, torch.int32, "filt_idx"),
                                      _chk_vec(fcnt_gt, torch.int32, "fcnt_gt", n_rows),
                                      _chk_vec(fcnt_eq, torch.int32, "fcnt_eq", n_rows), _stream()),
            "emg_eval_filter_count")


def eval_scores_dense(model_id, Q, ent, k_int, scale, cand=None, n_cand=None, precision=0):
    lib = L.load()
    pq, n_rows, ldq = _chk_table(Q, "Q")
    pe, ne, lde = _chk_table(ent, "ent")
    if cand is not None:
        n_cand = cand.numel()
    elif n_cand is None:
        n_cand = ne
    S = torch.full((n_rows, n_cand), float("nan"), dtype=torch.float32, device=ent.device)
    L.check(lib.emg_eval_scores_dense(model_id, pq, ldq, n_rows, pe, n_cand, lde,
                                      _chk_vec(cand, torch.int32, "cand"), k_int, scale, precision, None, 0,
                                      S.data_ptr(), S.stride(0), _stream()), "emg_eval_scores_dense")
    return S


def to_bf16(table, k_int, ld_dst=None):
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    ld_dst = ld_dst or ((k_int + 7) // 8) * 8
    out = torch.empty((nrows, ld_dst), dtype=torch.bfloat16, device=table.device)
    L.check(lib.emg_to_bf16(pt, nrows, ld, k_int, out.data_ptr(), ld_dst, _stream()), "emg_to_bf16")
    return out


def to_f16(table, k_int, ld_dst=None):
    """IEEE-half copy (RNE) of a table / query matrix, rows zero-padded to ``ld_dst`` (precision mode 2)"""
    lib = L.load()
    pt, nrows, ld = _chk_table(table, "table")
    ld_dst = ld_dst or bf16_ld(k_int)
    out = torch.empty((nrows, ld_dst), dtype=torch.float16, device=table.device)
    L.check(lib.emg_to_f16(pt, nrows, ld, k_int, out.data_ptr(), ld_dst, _stream()), "emg_to_f16")
    return out


# ---- bf16 MFMA evaluation path (throughput mode) -------------------------------------------------
def bf16_pad(k_int):
    """contraction length handed to the bf16 kernels: k_int rounded up to a whole MFMA k-step (16)"""
    return ((k_int + 15) // 16) * 16


def bf16_ld(k_int):
    """row stride (elements) of the bf16 operands: zero-padded to the kernel's 64-element k-slice"""
    return ((k_int + 63) // 64) * 64


def _chk_bf16(t, name):
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError("%s must be a 2-D bfloat16 CUDA tensor with unit column stride" % name)
    return t.data_ptr(), t.shape[0], t.stride(0)


def eval_pos_int_bf16(model_id, ent_bf16, k_int, scale, test_spo, side_mode, q_bf16):
    lib = L.load()
    pe, ne, lde = _chk_bf16(ent_bf16, "ent_bf16")
    pq, n_rows, ldq = _chk_bf16(q_bf16, "q_bf16")
    n_q = test_spo.shape[0]
    pos_int = torch.empty(n_rows, dtype=torch.int32, device=q_bf16.device)
    self_ent = torch.empty(n_rows, dtype=torch.int32, device=q_bf16.device)
    L.check(lib.emg_eval_pos_int_bf16(model_id, pe, lde, k_int, scale, _chk_vec(test_spo, torch.int32, "test_spo", 3 * n_q),
                                      n_q, side_mode, pq, ldq, pos_int.data_ptr(), self_ent.data_ptr(), _stream()),
            "emg_eval_pos_int_bf16")
    return pos_int, self_ent


def eval_count_bf16(model_id, q_bf16, pos_int, self_ent, ent_bf16, k_int, scale, cnt_gt, cnt_eq, cand=None, ent_offset=0,
                    need=0):
    """need: 0 both counters; 1 cnt_gt += #(>=) only ('worst'); 2 cnt_gt += #(>) only ('best'); cnt_eq unspecified then"""
    lib = L.load()
    pq, n_rows, ldq = _chk_bf16(q_bf16, "q_bf16")
    pe, ne, lde = _chk_bf16(ent_bf16, "ent_bf16")
    n_cand = cand.numel() if cand is not None else ne
    L.check(lib.emg_eval_count_bf16(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows),
                                    _chk_vec(self_ent, torch.int32, "self_ent", n_rows), n_rows, pe, n_cand, lde,
                                    _chk_vec(cand, torch.int32, "cand"), ent_offset, bf16_pad(k_int), scale,
                                    _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                    _chk_vec(cnt_eq, torch.int32, "cnt_eq", n_rows), int(need), _stream()),
            "emg_eval_count_bf16")


def eval_filter_count_bf16(model_id, q_bf16, pos_int, self_ent, ent_bf16, ent_offset, k_int, scale, filt_ptr, filt_idx,
                           fcnt_gt, fcnt_eq):
    lib = L.load()
    pq, n_rows, ldq = _chk_bf16(q_bf16, "q_bf16")
    pe, ne, lde = _chk_bf16(ent_bf16, "ent_bf16")
    L.check(lib.emg_eval_filter_count_bf16(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows),
                                           _chk_vec(self_ent, torch.int32, "self_ent", n_rows), n_rows, pe, ne, lde,
                                           ent_offset, k_int, scale, _chk_vec(filt_ptr, torch.int64, "filt_ptr", n_rows + 1),
                                           _chk_vec(filt_idx, torch.int32, "filt_idx"),
                                           _chk_vec(fcnt_gt, torch.int32, "fcnt_gt", n_rows),
                                           _chk_vec(fcnt_eq, torch.int32, "fcnt_eq", n_rows), _stream()),
            "emg_eval_filter_count_bf16")


def _chk_f16(t, name):
    if not (t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError("%s must be a 2-D float16 CUDA tensor with unit column stride" % name)
    return t.data_ptr(), t.shape[0], t.stride(0)


def eval_prefilter_bounds(ent, ent_f16, k_int):
    """(max ||e||, max ||e~||, max ||e~ - e||) over the rows of an f32 table (slab) and its half copy: 3 float64 ON THE
    DEVICE (no host round trip), what ``eval_prefilter_band`` needs of the table"""
    lib = L.load()
    pe, n, ld = _chk_table(ent, "ent")
    ph, nh, ldh = _chk_f16(ent_f16, "ent_f16")
    if nh != n:
        raise ValueError("ent_f16 must have the rows of ent")
    out = torch.empty(3, dtype=torch.float64, device=ent.device)
    L.check(lib.emg_eval_prefilter_bounds(pe, n, ld, ph, ldh, k_int, out.data_ptr(), _stream()), "emg_eval_prefilter_bounds")
    return out


def eval_prefilter_band(Q, Q_f16, k_int, bounds):
    """per query row the rigorous bound on |half-precision MFMA accumulator - exact f32 chain| (include/emgraph_hip.h)"""
    lib = L.load()
    pq, n, ldq = _chk_table(Q, "Q")
    ph, nh, ldh = _chk_f16(Q_f16, "Q_f16")
    if nh != n:
        raise ValueError("Q_f16 must have the rows of Q")
    band = torch.empty(n, dtype=torch.float32, device=Q.device)
    L.check(lib.emg_eval_prefilter_band(pq, n, ldq, ph, ldh, k_int, _chk_vec(bounds, torch.float64, "bounds", 3),
                                        band.data_ptr(), _stream()), "emg_eval_prefilter_band")
    return band


def prefilter_ld(k_cols):
    """row stride (elements) of the half-precision prefilter's operands for a contraction over k_cols columns"""
    return int(L.load().emg_eval_prefilter_ld(k_cols))


def eval_prefilter_segments(n_rows, n_cand, k_cols=400):
    """number of segments (= waves of the prefilter kernel) the pair buffer is cut into at a contraction width of k_cols;
    pair_count has one more entry"""
    return int(L.load().emg_eval_prefilter_segments_k(n_rows, n_cand, bf16_pad(k_cols)))


def prefilter_waves(k_cols):
    """segments each workgroup of the prefilter writes at this width (8, or 4 above 400 columns): the re-scoring's
    segments_per_block"""
    return int(L.load().emg_eval_prefilter_waves(bf16_pad(k_cols)))


def prefilter_max_cols():
    return int(L.load().emg_eval_prefilter_max_cols())


def eval_prefilter_f16(model_id, q_f16, pos_int, band, ent_f16, ent_offset, k_int, scale, cnt_gt, pairs, pair_count, cnt_eq=None):
    """half-precision MFMA prefilter of precision mode 2: definite `>` counts into cnt_gt, undecided (row, entity)
    pairs into ``pairs`` (int64 [capacity]), per-segment counts + overflow flag into ``pair_count`` (int32
    [segments + 1]).  Raises EmgError(EMG_ENOSUP) for shapes the register-stationary kernel does not cover.
    ``cnt_eq``: the form that also PROVES TIES (emg_eval_prefilter_f16_ties): candidates inside the positive's integer cell by more
    than the band are counted there, only the two bands around the cell's ends become pairs."""
    lib = L.load()
    pq, n_rows, ldq = _chk_f16(q_f16, "q_f16")
    pe, ne, lde = _chk_f16(ent_f16, "ent_f16")
    n_seg = eval_prefilter_segments(n_rows, ne, k_int)
    if cnt_eq is not None:
        L.check(lib.emg_eval_prefilter_f16_ties(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows),
                                                _chk_vec(band, torch.float32, "band", n_rows), n_rows, pe, ne, lde, ent_offset,
                                                bf16_pad(k_int), scale, _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                                _chk_vec(cnt_eq, torch.int32, "cnt_eq", n_rows),
                                                _chk_vec(pairs, torch.int64, "pairs"), _chk_vec(pair_count, torch.int32, "pair_count", n_seg + 1),
                                                pairs.numel(), _stream()), "emg_eval_prefilter_f16_ties")
        return n_seg
    L.check(lib.emg_eval_prefilter_f16(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows),
                                       _chk_vec(band, torch.float32, "band", n_rows), n_rows, pe, ne, lde, ent_offset,
                                       bf16_pad(k_int), scale, _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                       _chk_vec(pairs, torch.int64, "pairs"), _chk_vec(pair_count, torch.int32, "pair_count", n_seg + 1),
                                       pairs.numel(), _stream()), "emg_eval_prefilter_f16")
    return n_seg


def eval_rescore_pairs(model_id, Q, pos_int, ent, ent_offset, k_int, scale, pairs, pair_count, n_seg, cnt_gt, cnt_eq,
                       segments_per_block=4, rows_per_segment=0):
    """exact re-scoring of the prefilter's undecided pairs (counts read on the device); ``segments_per_block``: waves per
    workgroup of the prefilter that wrote them (prefilter_waves(width); 4 after the fixed-point one) — the XCD affinity of the
    re-scoring; ``rows_per_segment``: 32 after the half-precision MFMA prefilter (long segments are then re-scored with
    their query rows in LDS), 0 otherwise (include/emgraph_hip.h)"""
    lib = L.load()
    pq, n_rows, ldq = _chk_table(Q, "Q")
    pe, ne, lde = _chk_table(ent, "ent")
    L.check(lib.emg_eval_rescore_pairs_rows(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows), pe, lde, ent_offset,
                                            k_int, scale, _chk_vec(pairs, torch.int64, "pairs"), pairs.numel(),
                                            _chk_vec(pair_count, torch.int32, "pair_count", n_seg + 1), n_seg, segments_per_block,
                                            rows_per_segment, _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                            _chk_vec(cnt_eq, torch.int32, "cnt_eq", n_rows), _stream()), "emg_eval_rescore_pairs_rows")


def eval_rescore_pairs_tiles(model_id, Q, pos_int, ent, ent_offset, k_int, scale, pairs, pair_count, n_seg, sorted_pairs, tile_ws,
                             cnt_gt, cnt_eq):
    """entity-tile-major re-scoring of the prefilter's undecided pairs (include/emgraph_hip.h): pairs bucketed by tile of 32
    entity rows, the tile's rows in LDS, query rows streamed; ``sorted_pairs``: int64 scratch of pairs.numel() entries,
    ``tile_ws``: uint8 scratch of rescore_tiles_ws_bytes(rows of ent), zero before its first use"""
    lib = L.load()
    pq, n_rows, ldq = _chk_table(Q, "Q")
    pe, ne, lde = _chk_table(ent, "ent")
    L.check(lib.emg_eval_rescore_pairs_tiles(model_id, pq, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n_rows), pe, lde, ent_offset, ne,
                                             k_int, scale, _chk_vec(pairs, torch.int64, "pairs"), pairs.numel(),
                                             _chk_vec(pair_count, torch.int32, "pair_count", n_seg + 1), n_seg,
                                             _chk_vec(sorted_pairs, torch.int64, "sorted_pairs"), sorted_pairs.numel(),
                                             tile_ws.data_ptr(), tile_ws.numel(), _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                             _chk_vec(cnt_eq, torch.int32, "cnt_eq", n_rows), _stream()), "emg_eval_rescore_pairs_tiles")


def rescore_tiles_ws_bytes(n_local):
    return int(L.load().emg_eval_rescore_tiles_ws_bytes(n_local))


def to_f16_l2(src, k_int, is_query, ld_dst=None):
    """half rows of the TransE-L2 contraction (include/emgraph_hip.h): entity rows [e | n_hi | n_lo] -> (rows, residual
    max as 1 float64 on the device); query rows [2q | -1 | -1] -> (rows, the f32 rows 2q)"""
    lib = L.load()
    ps, n, ld = _chk_table(src, "src")
    ldd = ld_dst or prefilter_ld(k_int + 2)
    out = torch.empty((n, ldd), dtype=torch.float16, device=src.device)
    if is_query:
        dbl = torch.zeros_like(src)
        L.check(lib.emg_to_f16_l2(ps, n, ld, k_int, 1, out.data_ptr(), ldd, dbl.data_ptr(), None, _stream()), "emg_to_f16_l2")
        return out, dbl
    res = torch.zeros(1, dtype=torch.float64, device=src.device)
    L.check(lib.emg_to_f16_l2(ps, n, ld, k_int, 0, out.data_ptr(), ldd, None, res.data_ptr(), _stream()), "emg_to_f16_l2")
    return out, res


def eval_l2_thresholds(Q, pos_int, band, bounds4, k_int):
    lib = L.load()
    pq, n, ldq = _chk_table(Q, "Q")
    thr = torch.empty(2 * n, dtype=torch.float32, device=Q.device)
    L.check(lib.emg_eval_l2_thresholds(pq, n, ldq, _chk_vec(pos_int, torch.int32, "pos_int", n),
                                       _chk_vec(band, torch.float32, "band", n), _chk_vec(bounds4, torch.float64, "bounds4", 4),
                                       k_int, thr.data_ptr(), _stream()), "emg_eval_l2_thresholds")
    return thr


def eval_prefilter_f16_thr(q_f16, thr, ent_f16, ent_offset, k_cols, cnt_gt, pairs, pair_count):
    """the half-precision MFMA prefilter with given accumulator thresholds (TransE-L2); k_cols = contraction width
    (k_int + 2).  Raises EmgError(EMG_ENOSUP) for shapes the register-stationary kernel does not cover."""
    lib = L.load()
    pq, n_rows, ldq = _chk_f16(q_f16, "q_f16")
    pe, ne, lde = _chk_f16(ent_f16, "ent_f16")
    n_seg = eval_prefilter_segments(n_rows, ne, k_cols)
    L.check(lib.emg_eval_prefilter_f16_thr(pq, ldq, _chk_vec(thr, torch.float32, "thr", 2 * n_rows), n_rows, pe, ne, lde, ent_offset,
                                           bf16_pad(k_cols), _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                           _chk_vec(pairs, torch.int64, "pairs"),
                                           _chk_vec(pair_count, torch.int32, "pair_count", n_seg + 1), pairs.numel(), _stream()),
            "emg_eval_prefilter_f16_thr")
    return n_seg


def _chk_u16(t, name):
    if not (t.is_cuda and t.dtype == torch.int16 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError("%s must be a 2-D int16 CUDA tensor (u16 image rows) with unit column stride" % name)
    return t.data_ptr(), t.shape[0], t.stride(0)


def eval_sad_range(ent, rel, k_int):
    """(max|ent|, max|rel|) as 2 float64 ON THE DEVICE: the range of the 16-bit fixed-point images of the TransE-L1
    prefilter (csrc/emg_rank_sad.hip)"""
    lib = L.load()
    pe, ne, lde = _chk_table(ent, "ent")
    pr, nr, ldr = _chk_table(rel, "rel")
    out = torch.empty(2, dtype=torch.float64, device=ent.device)
    L.check(lib.emg_eval_sad_range(pe, ne, lde, pr, nr, ldr, k_int, out.data_ptr(), _stream()), "emg_eval_sad_range")
    return out


def eval_sad_ld(k_int):
    return int(L.load().emg_eval_sad_ld(k_int))


def eval_sad_quantize(src, k_int, rng):
    """u16 image (stored in an int16 tensor) of the f32 rows over the range ``rng`` (eval_sad_range)"""
    lib = L.load()
    ps, n, ld = _chk_table(src, "src")
    ldd = eval_sad_ld(k_int)
    out = torch.empty((n, ldd), dtype=torch.int16, device=src.device)
    L.check(lib.emg_eval_sad_quantize(ps, n, ld, k_int, _chk_vec(rng, torch.float64, "range", 2), out.data_ptr(), ldd,
                                      _stream()), "emg_eval_sad_quantize")
    return out


def eval_sad_thresholds(pos_int, k_int, rng):
    """int32 [2, n_rows] holding the uint32 sums (lo, hi) between which a candidate is undecided"""
    lib = L.load()
    n = pos_int.numel()
    out = torch.empty((2, n), dtype=torch.int32, device=pos_int.device)
    L.check(lib.emg_eval_sad_thresholds(_chk_vec(pos_int, torch.int32, "pos_int", n), n, k_int,
                                        _chk_vec(rng, torch.float64, "range", 2), out[0].data_ptr(), out[1].data_ptr(),
                                        _stream()), "emg_eval_sad_thresholds")
    return out


def eval_sad_segments(n_rows, n_cand):
    return int(L.load().emg_eval_sad_segments(n_rows, n_cand))


def eval_prefilter_sad(q_u16, thresholds, ent_u16, ent_offset, k_int, cnt_gt, pairs, pair_count):
    """fixed-point prefilter of the TransE-L1 exact-fast mode: definite `>` counts into cnt_gt, undecided (row, entity)
    pairs into ``pairs`` in the layout eval_rescore_pairs reads"""
    lib = L.load()
    pq, n_rows, ldq = _chk_u16(q_u16, "q_u16")
    pe, ne, lde = _chk_u16(ent_u16, "ent_u16")
    n_seg = eval_sad_segments(n_rows, ne)
    if not (thresholds.is_cuda and thresholds.dtype == torch.int32 and tuple(thresholds.shape) == (2, n_rows)
            and thresholds.is_contiguous()):
        raise ValueError("thresholds must be the int32 [2, n_rows] tensor of eval_sad_thresholds")
    L.check(lib.emg_eval_prefilter_sad(pq, ldq, thresholds[0].data_ptr(), thresholds[1].data_ptr(), n_rows, pe, ne, lde,
                                       ent_offset, k_int, _chk_vec(cnt_gt, torch.int32, "cnt_gt", n_rows),
                                       _chk_vec(pairs, torch.int64, "pairs"),
                                       _chk_vec(pair_count, torch.int32, "pair_count", n_seg + 1), pairs.numel(), _stream()),
            "emg_eval_prefilter_sad")
    return n_seg


def eval_scores_dense_bf16(model_id, q_bf16, ent_bf16, k_int, scale, cand=None):
    lib = L.load()
    pq, n_rows, ldq = _chk_bf16(q_bf16, "q_bf16")
    pe, ne, lde = _chk_bf16(ent_bf16, "ent_bf16")
    n_cand = cand.numel() if cand is not None else ne
    S = torch.full((n_rows, n_cand), float("nan"), dtype=torch.float32, device=q_bf16.device)
    L.check(lib.emg_eval_scores_dense_bf16(model_id, pq, ldq, n_rows, pe, n_cand, lde,
                                           _chk_vec(cand, torch.int32, "cand"), bf16_pad(k_int), scale, S.data_ptr(),
                                           S.stride(0), _stream()), "emg_eval_scores_dense_bf16")
    return S


# ---- one-call forms (emg_api.hip): the C-ABI a maintainer binds from the reference; this package's own training /
# ---- evaluation loops use the fine-grained calls above so that they can overlap stages on several streams
def corrupt_fit(pos, eta, side, entities_size=0, entities_list=None, seed=0, counter=0):
    lib = L.load()
    B = pos.shape[0]
    out = torch.empty((B * eta, 3), dtype=torch.int32, device=pos.device)
    n_list = entities_list.numel() if entities_list is not None else 0
    L.check(lib.emg_corrupt_fit(_chk_vec(pos, torch.int32, "pos", 3 * B), B, eta, side, entities_size,
                                _chk_vec(entities_list, torch.int32, "entities_list"), n_list,
                                seed & 0xFFFFFFFFFFFFFFFF, counter & 0xFFFFFFFFFFFFFFFF, out.data_ptr(), _stream()),
            "emg_corrupt_fit")
    return out


def rank_1vsall(model_id, ent, rel, k_int, scale, test_spo, side_mode, strategy=0, cand=None, filt_ptr=None,
                filt_idx=None, precision_mode=0):
    """ranks int32 [n_q] (or [n_q, 2] = [subject, object] for side_mode 's,o'), everything on the current stream"""
    lib = L.load()
    pe, ne, lde = _chk_table(ent, "ent")
    pr, nr, ldr = _chk_table(rel, "rel")
    n_q = test_spo.shape[0]
    out = torch.empty((n_q, 2) if side_mode == L.EVAL_S_O else (n_q,), dtype=torch.int32, device=ent.device)
    n_rows = 2 * n_q if side_mode >= L.EVAL_SPO else n_q
    L.check(lib.emg_rank_1vsall(model_id, pe, ne, lde, pr, nr, ldr, k_int, scale,
                                _chk_vec(test_spo, torch.int32, "test_spo", 3 * n_q), n_q, side_mode,
                                _chk_vec(cand, torch.int32, "cand"), cand.numel() if cand is not None else 0,
                                _chk_vec(filt_ptr, torch.int64, "filt_ptr", n_rows + 1 if filt_ptr is not None else None),
                                _chk_vec(filt_idx, torch.int32, "filt_idx"), strategy, precision_mode, out.data_ptr(),
                                _stream()), "emg_rank_1vsall")
    return out


def train_step_workspace_bytes(B, eta_total, k_int, n_ent, n_rel):
    n = L.load().emg_train_step_workspace_bytes(B, eta_total, k_int, n_ent, n_rel)
    if n < 0:
        L.check(-1, "emg_train_step_workspace_bytes")
    return int(n)


def train_step(model_id, ent, rel, k_int, scale, pos, eta, sides, loss_id, loss_accum, opt_id, step, hyper, workspace,
               margin=1.0, alpha=0.5, states=(None, None, None, None), tags=(None, None), n_choices=0,
               entities_list=None, seed=0, counter0=0, inj_mask=None, inj_repl=None, inplace=True):
    lib = L.load()
    a = L.StepArgs()
    a.model, a.k_int, a.scale, a.eta, a.n_sides = model_id, k_int, scale, eta, len(sides)
    for i, sd in enumerate(sides):
        a.sides[i] = sd
    a.ent, a.n_ent, a.ld_ent = _chk_table(ent, "ent")
    a.rel, a.n_rel, a.ld_rel = _chk_table(rel, "rel")
    a.ent_state0, a.ent_state1, a.rel_state0, a.rel_state1 = [(_chk_table(t, "state")[0] if t is not None else None)
                                                              for t in states]
    a.tag_ent, a.tag_rel = [_chk_vec(t, torch.int32, "tag") for t in tags]
    a.opt, a.step = opt_id, step
    for i, h in enumerate(hyper):
        a.hyper[i] = float(h)
    B = pos.shape[0]
    a.pos, a.B = _chk_vec(pos, torch.int32, "pos", 3 * B), B
    a.n_choices = n_choices
    a.entities_list = _chk_vec(entities_list, torch.int32, "entities_list")
    a.seed, a.draw_counter0 = seed & 0xFFFFFFFFFFFFFFFF, counter0 & 0xFFFFFFFFFFFFFFFF
    a.inj_mask = _chk_vec(inj_mask, torch.int32, "inj_mask")
    a.inj_repl = _chk_vec(inj_repl, torch.int32, "inj_repl")
    a.loss, a.margin, a.alpha = loss_id, margin, alpha
    a.loss_accum = _chk_vec(loss_accum, torch.float64, "loss_accum", 1)
    a.inplace = 1 if inplace else 0
    a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    L.check(lib.emg_train_step(C.byref(a), _stream()), "emg_train_step")
