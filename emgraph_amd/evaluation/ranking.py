"""Host side of the filtered 1-vs-all ranking path (mirrors emgraph/evaluation/protocol.py and the
eval half of emgraph/models/EmbeddingModel.py).

What stays on the host (numpy, vectorised — no SQLite, no per-triple Python):
  * the filter index: a one-off CSR of known positives per (query, side), replacing
    SQLiteAdapter.get_participating_entities (sqlite_adapter.py:449-508: two SQL queries + a connect per
    test triple);
  * turning the device's (gt, eq) counters into ranks (perform_comparision's three strategies,
    EmbeddingModel.py:2010-2033, and the rank assembly :1966-1986).
Everything numeric runs in libemgraph_hip.so.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib as L
from .. import device as D


def _expand_ranges(lo, hi):
    """indices of concatenated ranges [lo_i, hi_i) + the owning range id of each index"""
    lens = (hi - lo).astype(np.int64)
    total = int(lens.sum())
    owner = np.repeat(np.arange(len(lo), dtype=np.int64), lens)
    if total == 0:
        return np.zeros(0, np.int64), owner
    starts = np.cumsum(lens) - lens
    idx = np.arange(total, dtype=np.int64) - np.repeat(starts, lens) + np.repeat(lo.astype(np.int64), lens)
    return idx, owner


def _stable_argsort(key):
    """stable argsort of non-negative int64 keys.  Large key sets that fit 31 bits go through the library's stable radix
    sort on the GPU (emg_group_dest: the grouping the training step uses; 1M keys in well under a millisecond + two 4 MB
    copies, against ~75 ms for numpy's merge sort); anything else — small sets, wide keys, a GPU-less host — through numpy.
    Both orders are THE stable order, so the index is the same either way."""
    n = len(key)
    if n >= 100_000 and torch.cuda.is_available() and int(key.max()) < (1 << 31) - 1 and int(key.min()) >= 0:
        dev = torch.device("cuda", torch.cuda.current_device())
        # on its own stream: the index is built lazily, i.e. usually while the big count kernel of the first query tile
        # runs on the caller's stream — the sort must not queue behind it
        side = _sort_streams.get(dev.index)
        if side is None:
            side = _sort_streams[dev.index] = torch.cuda.Stream(device=dev, priority=-1)
        with torch.cuda.stream(side):
            kd = torch.from_numpy(key.astype(np.int32)).to(dev)
            ws = torch.empty(D.apply_workspace_bytes(n, int(key.max()) + 1), dtype=torch.uint8, device=dev)
            D.group_dest(kd, n, int(key.max()) + 1, ws)
            order = D.apply_workspace_views(ws, n)[1].cpu().numpy().astype(np.int64)   # (synchronises the side stream only)
        return order
    return np.argsort(key, kind="stable")


_sort_streams = {}


class FilterIndex:
    """One-off index of the filter triples: (s,p)-sorted objects and (o,p)-sorted subjects.  Built once
    per evaluate_performance call (two argsorts); each query chunk then costs two searchsorted calls.
    Replaces the SQLite side-table + its (subject,predicate) / (predicate,object) indexes
    (sqlite_adapter.py:54-98,234-262) and the two SQL queries + connect per test triple (:449-508)."""

    def __init__(self, filter_triples):
        F = np.asarray(filter_triples, dtype=np.int64).reshape(-1, 3)
        self.n_rel = int(F[:, 1].max()) + 1 if len(F) else 1
        self.max_entity = int(max(F[:, 0].max(), F[:, 2].max())) if len(F) else -1
        self._F = F
        self._Fd = None
        self._cols = None
        self._sides = {}     # built on first use (_side): rank_triples_device asks for the CSR of a query tile AFTER it has
                             # launched the tile's count kernel, so the two sorts run underneath that kernel

    def _side(self, name):
        """(sorted keys, values in that order) of one side: 'obj' = objects by (subject, relation), 'sub' = subjects by
        (object, relation)"""
        if name not in self._sides:
            kcol, vcol = {"obj": (0, 2), "sub": (2, 0)}[name]
            n, kmax = len(self._F), (self.max_entity + 1) * self.n_rel
            if n >= 100_000 and torch.cuda.is_available() and 0 <= kmax < (1 << 31) - 1 and int(self._F.min()) >= 0:
                # large filter sets on a GPU host: keys, the stable sort (the library's grouping) and the two gathers on the
                # device, on a stream of their own (usually underneath the count kernel of the first query tile)
                dev = torch.device("cuda", torch.cuda.current_device())
                side = _sort_streams.get(dev.index)
                if side is None:
                    side = _sort_streams[dev.index] = torch.cuda.Stream(device=dev, priority=-1)
                with torch.cuda.stream(side):
                    if self._Fd is None:
                        self._Fd = torch.from_numpy(self._F).to(dev)
                    key = self._Fd[:, kcol] * self.n_rel + self._Fd[:, 1]
                    ws = torch.empty(D.apply_workspace_bytes(n, kmax + 1), dtype=torch.uint8, device=dev)
                    D.group_dest(key.to(torch.int32), n, kmax + 1, ws)
                    order = D.apply_workspace_views(ws, n)[1].long()
                    self._sides[name] = (key.index_select(0, order).cpu().numpy(),
                                         self._Fd[:, vcol].index_select(0, order).cpu().numpy())
                if len(self._sides) == 2:
                    self._Fd = None
                return self._sides[name]
            if self._cols is None:   # contiguous columns: the strided [n, 3] views cost 3-4x in every pass below
                self._cols = [np.ascontiguousarray(self._F[:, c]) for c in range(3)]
            key = self._cols[kcol] * self.n_rel + self._cols[1]
            order = _stable_argsort(key)
            self._sides[name] = (key[order], self._cols[vcol][order])
        return self._sides[name]

    def _pairs(self, name, q_ent, q_rel, q_self):
        """(row, entity) pairs: filter entities matching the row's (entity, relation) key, plus the row's
        own entity ('select <id> union select distinct ...', sqlite_adapter.py:472-495)."""
        skey, sval = self._side(name)
        qk = np.where(q_rel < self.n_rel, q_ent * self.n_rel + q_rel, -1)
        lo = np.searchsorted(skey, qk, side="left")
        hi = np.searchsorted(skey, qk, side="right")
        idx, owner = _expand_ranges(lo, hi)
        rows = np.concatenate([owner, np.arange(len(qk), dtype=np.int64)])
        ents = np.concatenate([sval[idx], q_self.astype(np.int64)])
        return rows, ents

    def csr(self, test_triples, side_mode, n_ent, entities_subset=None):
        """CSR (filt_ptr int64[n_rows+1], filt_idx int32) of known positives per query ROW, in the row
        order of emg_eval_build_queries (object-side rows first for 's+o'/'s,o').

        object-side row of (s,p,o):  {o} U {o' : (s,p,o') in F};  subject-side: {s} U {s' : (s',p,o) in F}.
        With ``entities_subset`` only members of the subset are kept (EmbeddingModel.py:1898-1940 intent)."""
        T = np.asarray(test_triples, dtype=np.int64).reshape(-1, 3)
        n_q = T.shape[0]
        rows_all, ents_all = [], []
        row_base = 0
        if side_mode in (L.EVAL_O, L.EVAL_SPO, L.EVAL_S_O):
            r, e = self._pairs("obj", T[:, 0], T[:, 1], T[:, 2])
            rows_all.append(r + row_base)
            ents_all.append(e)
            row_base += n_q
        if side_mode in (L.EVAL_S, L.EVAL_SPO, L.EVAL_S_O):
            r, e = self._pairs("sub", T[:, 2], T[:, 1], T[:, 0])
            rows_all.append(r + row_base)
            ents_all.append(e)
            row_base += n_q
        n_rows = row_base
        rows = np.concatenate(rows_all) if rows_all else np.zeros(0, np.int64)
        ents = np.concatenate(ents_all) if ents_all else np.zeros(0, np.int64)
        if entities_subset is not None:
            keep = np.isin(ents, np.asarray(entities_subset, dtype=np.int64))
            rows, ents = rows[keep], ents[keep]
        comb = np.unique(rows * np.int64(n_ent) + ents)  # de-duplicate (SQL UNION / DISTINCT), sorted by row
        rows_u = comb // n_ent
        ents_u = (comb - rows_u * n_ent).astype(np.int32)
        ptr = np.zeros(n_rows + 1, np.int64)
        np.cumsum(np.bincount(rows_u, minlength=n_rows), out=ptr[1:])
        return ptr, ents_u


def build_filter_csr(filter_triples, test_triples, side_mode, n_ent, entities_subset=None):
    """Convenience: FilterIndex(filter_triples).csr(...)."""
    return FilterIndex(filter_triples).csr(test_triples, side_mode, n_ent, entities_subset)


def _cmp(gt, eq, strategy):
    """perform_comparision (EmbeddingModel.py:2018-2033) from the (>, ==) counters."""
    if strategy == "worst":
        return gt + eq
    if strategy == "best":
        return gt
    if strategy == "middle":
        return gt + np.ceil(eq / 2).astype(np.int64)
    raise AssertionError("Invalid score comparision type!")


def ranks_from_counts(gt, eq, fgt, feq, n_q, corrupt_side, strategy):
    """rank assembly (EmbeddingModel.py:1966-1986).  's,o' -> [n,2] = [subject_rank, object_rank]."""
    gt, eq, fgt, feq = [np.asarray(a, dtype=np.int64) for a in (gt, eq, fgt, feq)]
    if corrupt_side in ("s", "o"):
        return _cmp(gt, eq, strategy) + 1 - _cmp(fgt, feq, strategy)
    o, s = slice(0, n_q), slice(n_q, 2 * n_q)
    if corrupt_side == "s,o":
        rank_s = _cmp(gt[s], eq[s], strategy) + 1 - _cmp(fgt[s], feq[s], strategy)
        rank_o = _cmp(gt[o], eq[o], strategy) + 1 - _cmp(fgt[o], feq[o], strategy)
        return np.stack([rank_s, rank_o], axis=1)
    if corrupt_side == "s+o":
        return (_cmp(gt[o] + gt[s], eq[o] + eq[s], strategy) + 1
                - _cmp(fgt[s], feq[s], strategy) - _cmp(fgt[o], feq[o], strategy))
    raise ValueError("Invalid argument value for corruption side passed for evaluation")


def _ev_start(stats):
    if stats is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _ev_stop(stats, e0):
    if stats is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    stats.setdefault("_events", []).append((e0, e1))


def _ev_collect(stats):
    if stats is None:
        return
    torch.cuda.synchronize()
    evs = stats.pop("_events", [])
    stats["count_ms"] = stats.get("count_ms", 0.0) + sum(a.elapsed_time(b) for a, b in evs)
    stats["count_launches"] = stats.get("count_launches", 0) + len(evs)


# ------------------------------------------------------------------------------------------------
# precision 2: exact ranks at MFMA speed (bf16 prefilter + exact re-scoring of the undecided candidates)
# ------------------------------------------------------------------------------------------------
_U32 = 2.0 ** -24   # unit roundoff of fp32


def table_norm_bounds(ent, ent_f16, k_int):
    """(max ||e||, max ||e~||, max ||e~ - e||) over the rows of the fp32 table and of its half copy, float64 on the
    device (emg_eval_prefilter_bounds) — computed from the tensors as they are NOW (a stale copy only widens the
    band, it never breaks it)."""
    return D.eval_prefilter_bounds(ent, ent_f16, k_int)


def prefilter_band(Q, Qb, k_int, bounds):
    """Per query row r, a RIGOROUS bound on |a - s| for every candidate e, where
        a = fp32-accumulated MFMA product of the half-ROUNDED operands  (what emg_eval_prefilter_f16 compares),
        s = the exact path's k-ordered fp32 fmaf chain of the fp32 operands (what decides the reference rank).
    With q~ = q + dq, e~ = e + de:   sum q~e~ - sum qe = dq.e~ + q.de,  so by Cauchy-Schwarz
        |a - s| <= ||dq|| ||e~|| + ||q|| ||de||  +  g (||q~|| ||e~|| + ||q|| ||e||),
    g = 2 (k + 32) 2^-24 bounding the accumulation error of either sum (standard gamma_k, doubled).  The norms of the
    residuals are the ACTUAL ones of this query tile / this table, not worst-case roundoff: ~0.4 x 2^-11 relative.
    Evaluated in float64 by emg_eval_prefilter_band (csrc/emg_rank.hip), inflated by 1e-6 and rounded up to float."""
    return D.eval_prefilter_band(Q, Qb, k_int, bounds)


def prefilter_band_reference(Q, Qb, k_int, bounds):
    """the same bound with torch float64 operations (tests compare the kernel against it)"""
    e_max, eb_max, de_max = (float(x) for x in bounds)
    q = Q[:, :k_int].double()
    qb = Qb[:, :k_int].double()
    nq, nqb, ndq = q.norm(dim=1), qb.norm(dim=1), (qb - q).norm(dim=1)
    g = 2.0 * (k_int + 32) * _U32
    return (ndq * eb_max + nq * de_max + g * (nqb * eb_max + nq * e_max)) * (1.0 + 1e-6)


class PrefilterTables:
    """what precision 2 derives from the entity table, built once per evaluation run (like the FilterIndex): the
    half-precision copy and the norm bounds of prefilter_band.  Valid while the table does not change."""

    def __init__(self, ent, k_int):
        self.ent_f16 = D.to_f16(ent, k_int, ld_dst=D.prefilter_ld(k_int))
        self.k_int = k_int
        self._bounds = {}
        self._ent = ent
        self.undecided = {}   # (slab, side) -> undecided fraction a probe of the run's query rows found (_prefilter_probe)

    def bounds(self, e0, n):
        if (e0, n) not in self._bounds:
            self._bounds[(e0, n)] = table_norm_bounds(self._ent[e0:e0 + n], self.ent_f16[e0:e0 + n], self.k_int)
        return self._bounds[(e0, n)]


class L2Tables:
    """what the TransE-L2 exact-fast mode derives from the entity table, built once per evaluation run: the half rows
    [e | n_hi | n_lo] of the augmented contraction, the residual of the norm split and the norm bounds of the band."""

    def __init__(self, ent, k_int):
        self.ent_f16, self._res = D.to_f16_l2(ent, k_int, False)
        self.k_int = k_int
        self._bounds = {}
        self._ent = ent

    def bounds(self, e0, n):
        if (e0, n) not in self._bounds:
            b3 = D.eval_prefilter_bounds(self._ent[e0:e0 + n], self.ent_f16[e0:e0 + n], self.k_int)
            self._bounds[(e0, n)] = torch.cat([b3, self._res])
        return self._bounds[(e0, n)]


class SadTables:
    """what the TransE-L1 exact-fast mode derives from the tables, built once per evaluation run: the range of the
    fixed-point map and the 16-bit image of the entity table.  Valid while the tables do not change."""

    def __init__(self, ent, rel, k_int):
        self.range = D.eval_sad_range(ent, rel, k_int)
        self.ent_u16 = D.eval_sad_quantize(ent, k_int, self.range)
        self.k_int = k_int


def resolve_auto_precision(model_id, k_int, n_test, n_ent, entities_subset=None):
    """what precision 'auto' means for a call: the exact-fast mode (2) returns the SAME ranks as precision 0, bit for bit,
    and pays off once the 1-vs-all product is large enough to amortise the half-precision copy of the table; its kernels
    cover the common widths"""
    covered = lambda w: 32 < w <= D.prefilter_max_cols()   # noqa: E731  (the prefilter kernel pads a width up to its next instantiation)
    applies = ((model_id in (L.DISTMULT, L.COMPLEX, L.HOLE) and covered(k_int)) or (model_id == L.TRANSE_L1 and k_int >= 16)
               or (model_id == L.TRANSE_L2 and covered(k_int + 2)))
    return 2 if (applies and entities_subset is None and n_test >= 128 and n_ent >= 32768) else 0


def derived_tables(model_id, ent, rel, k_int):
    """the tables the exact-fast mode derives from the embedding tables (half-precision copy + norm bounds, 16-bit image +
    range, augmented half rows): valid while the tables do not change — a fitted model keeps them (get_ranks)"""
    if model_id == L.TRANSE_L1:
        return SadTables(ent, rel, k_int)
    if model_id == L.TRANSE_L2:
        return L2Tables(ent, k_int)
    return PrefilterTables(ent, k_int)


_pair_buffers = {}


def _pair_buffer(device, n_seg):
    """(pairs int64 [cap], per-segment counts int32 [n_seg + 1]) scratch of the prefilter, cached per device.
    A wave of the prefilter covers 32 query rows x up to 4096 entities; 2048 entries hold 1.5 % of them undecided
    (random positives on Gaussian tables leave ~0.7 %, a trained model a tenth of that) — and 2048 entries are what the
    prefilter's bitmap form needs of a segment (64 per entity tile, emg_rank_bf16.hip MODE 3; below it the slower emitting form
    runs): the buffer grows with the call, 1 GiB at 8192 query rows x 1M entities, at most 4 GiB (EMG_PAIR_LOG2 = log2 entries)."""
    per = max(64, min(int(os.environ.get("EMG_PAIR_CAP", "2048")), (1 << int(os.environ.get("EMG_PAIR_LOG2", "29"))) // max(n_seg, 1)))
    cap = n_seg * per
    key = (device.type, device.index)
    buf = _pair_buffers.get(key)
    if buf is None or buf[0].numel() < cap or buf[1].numel() < n_seg + 1:
        buf = (torch.empty(cap, dtype=torch.int64, device=device), torch.zeros(n_seg + 1, dtype=torch.int32, device=device))
        _pair_buffers[key] = buf
    return buf[0][:cap], buf[1][:n_seg + 1]


_tile_buffers = {}


def _tile_buffers_for(device, pairs, n_local):
    """(sorted pairs int64 [pairs.numel()], tile workspace uint8) scratch of the tile-major re-scoring, cached per device; the
    workspace is zero between calls (the library leaves it so)"""
    key = (device.type, device.index)
    buf = _tile_buffers.get(key)
    need = D.rescore_tiles_ws_bytes(n_local)
    if buf is None or buf[0].numel() < pairs.numel() or buf[1].numel() < need:
        buf = (torch.empty(pairs.numel(), dtype=torch.int64, device=device), torch.zeros(need, dtype=torch.uint8, device=device))
        _tile_buffers[key] = buf
    return buf[0][:pairs.numel()], buf[1]


def _rescore(model_id, Q, pos_int, slab, e0, k_int, scale, pairs, pcount, n_seg, cnt, waves, rows):
    """exact re-scoring of the prefilter's undecided pairs: segment-wise with the query rows in LDS (the default), or
    entity-tile-major (EMG_RESCORE=tiles: the pairs bucketed by tile of 32 entity rows, the tile's rows in LDS, query rows
    streamed).  Measured at C4's size (round 5, profiles/r5_*_pmc_rescore.txt): the segment form already finds 88 % of its
    rows in L2 and both forms are bound by LDS instruction issue (4800 bytes through LDS per pair), so the tile form's L2
    hits buy nothing: 19.2 against 17.1 ms per 8192 query rows."""
    mode = os.environ.get("EMG_RESCORE", "segments")
    if mode == "tiles" and rows == 32:
        sorted_pairs, tile_ws = _tile_buffers_for(Q.device, pairs, slab.shape[0])
        try:
            D.eval_rescore_pairs_tiles(model_id, Q, pos_int, slab, e0, k_int, scale, pairs, pcount, n_seg, sorted_pairs, tile_ws, cnt[0], cnt[1])
            return
        except L.EmgError:   # rows that are not 16-byte aligned / an image that does not fit LDS: the segment form
            pass
    D.eval_rescore_pairs(model_id, Q, pos_int, slab, e0, k_int, scale, pairs, pcount, n_seg, cnt[0], cnt[1], waves, rows)


# A table whose candidates the half-precision band cannot decide (a freshly initialised model, the first epochs of a fit: the
# scores crowd around the positive's) overflows the pair buffer in every tile: the prefilter pass is wasted and the exact kernel
# runs anyway (72 instead of 55 ms per 8192 x 1M pass).  So precision 2 asks first: ONE workgroup's worth of the call's query
# rows (128 triples) through the prefilter alone, the undecided fraction read back (~0.3 ms, remembered on the PrefilterTables
# of the evaluation run).  Above the fraction the pair buffer holds with room to spare the whole call takes the exact kernel.
_PROBE_TRIPLES = 128
_PROBE_MAX_UNDECIDED = 0.011   # (a wave's segment of the pair buffer holds 1.56 % of its 32 x 4096 candidates)


def _prefilter_probe(model_id, ent, rel, slab, e0, k_int, scale, T, side_mode, ent_f16, bounds, ties=False):
    Tt = torch.from_numpy(np.ascontiguousarray(T[:_PROBE_TRIPLES])).to(ent.device)
    Q, pos_int = D.eval_build_queries(model_id, ent, rel, k_int, scale, Tt, side_mode)
    n_rows, n_cand = Q.shape[0], slab.shape[0]
    if n_rows <= 128 or n_cand == 0:
        return 0.0   # (the register-stationary kernel wants more than 128 rows: such calls take the exact kernel tile by tile)
    Qb = D.to_f16(Q, k_int, ld_dst=D.prefilter_ld(k_int))
    band = prefilter_band(Q, Qb, k_int, bounds)
    n_seg = D.eval_prefilter_segments(n_rows, n_cand, k_int)
    pairs, pcount = _pair_buffer(ent.device, n_seg)
    cnt = torch.zeros((2, n_rows), dtype=torch.int32, device=ent.device)
    try:
        D.eval_prefilter_f16(model_id, Qb, pos_int, band, ent_f16[e0:e0 + n_cand], e0, k_int, scale, cnt[0], pairs, pcount,
                             cnt_eq=cnt[1] if ties else None)
    except L.EmgError:
        return 1.0 if ties else 0.0   # (the ties form needs segments that hold the bitmap: without it, the exact kernel)
    over, n_pairs = (int(v) for v in torch.stack([pcount[n_seg].long(), pcount[:n_seg].sum()]).cpu())
    return 1.0 if over else n_pairs / float(n_rows * n_cand)


def rank_triples_device(model_id, ent, rel, k_int, scale, test_triples, corrupt_side="s,o", strategy="worst",
                        filter_triples=None, entities_subset=None, query_chunk=4096, precision=0, shard=None,
                        ent_bf16=None, stats=None, ent_f16=None):
    """Ranks of ``test_triples`` (int ids) against all entities (or ``entities_subset``).

    ``shard=(rank, world)`` (multi-GPU, see parallel.py): every rank holds the tables, scores the query tile
    against ITS contiguous candidate range only and the int32 counters are all-reduced (RCCL) before the
    ranks are assembled — exact, because counts are integers.

    ``precision=1``: bf16 MFMA throughput mode for DistMult/ComplEx/HolE (ranks agree with the exact f32 path
    statistically, not bit for bit); ``ent_bf16`` optionally passes a cached bf16 copy of the table.

    ``precision=2``: EXACT ranks (bit-equal to precision 0) at MFMA speed for DistMult/ComplEx/HolE: a half-precision MFMA kernel
    decides every candidate whose score is farther from the positive's than a rigorous per-row error bound
    (prefilter_band), the others — typically 0.1-3 % — are re-scored with the exact f32 chain; shapes the prefilter
    kernel does not cover, and query tiles with too many undecided candidates, take the exact kernel (stats['fallback']).

    ``precision=2`` with TransE-L1: the same contract through a different prefilter — sums of absolute differences of
    16-bit fixed-point images of the rows (``v_sad_u16``, csrc/emg_rank_sad.hip) bound every candidate's score from both
    sides; the undecided ones are re-scored with the exact chain.  TransE-L2: ||q-e||^2 = |q|^2 - (2q.e - |e|^2) is a
    contraction over k+2 coordinates and goes through the half-precision MFMA prefilter with thresholds derived for the
    squared distance (``L2Tables``), at the widths that kernel covers.

    ``precision='auto'``: 2 where it applies and pays (contraction model, or TransE-L2 with k+2, at a width the prefilter kernel
    covers, or TransE-L1; no candidate list, at least 128 test triples against at least 32768 entities), else 0 — the ranks are the same either way.

    ``stats`` (dict, optional): receives ``count_ms`` = device time of the 1-vs-all count kernel launches
    (HIP events on the launch stream) and ``count_launches``."""
    if precision == "auto":
        n_test = int(np.asarray(test_triples).reshape(-1, 3).shape[0])
        precision = resolve_auto_precision(model_id, k_int, n_test, int(ent.shape[0]), entities_subset)
    if model_id == L.TRANSE_P:
        precision = 0   # TransE with an order of the norm other than 1 / 2: the exact chain kernel (same ranks, no prefilter form)
    if precision not in (0, 1, 2):
        raise ValueError("precision must be 0 (exact f32), 1 (bf16 MFMA), 2 (exact via half-precision prefilter) or 'auto' (0 or 2)")
    if precision == 2 and entities_subset is not None:
        precision = 0   # candidate lists go through the exact kernel: same ranks
    sad = precision == 2 and model_id == L.TRANSE_L1   # fixed-point prefilter instead of the half-precision MFMA one
    l2 = precision == 2 and model_id == L.TRANSE_L2    # MFMA prefilter on the augmented rows
    if precision == 1 and model_id not in (L.DISTMULT, L.COMPLEX, L.HOLE):
        raise ValueError("the bf16 MFMA mode needs a contraction model (DistMult, ComplEx, HolE)")
    if corrupt_side not in L.EVAL_SIDE_IDS:
        raise ValueError("Invalid argument value for corruption side passed for evaluation")
    if strategy not in ("worst", "best", "middle"):
        raise AssertionError("Invalid score comparision type!")
    side_mode = L.EVAL_SIDE_IDS[corrupt_side]
    T = np.ascontiguousarray(np.asarray(test_triples, dtype=np.int32).reshape(-1, 3))
    n = T.shape[0]
    n_ent = int(ent.shape[0])
    rank, world = shard if shard is not None else (0, 1)
    from .. import parallel
    cand = None
    slab, e0 = ent, 0
    subset_local = entities_subset
    if entities_subset is not None:
        sub = np.ascontiguousarray(np.asarray(entities_subset, dtype=np.int32))
        if world > 1:
            r0, r1 = parallel.entity_range(len(sub), rank, world)
            sub = sub[r0:r1]
        subset_local = sub
        cand = torch.from_numpy(sub).to(ent.device)
    elif world > 1:
        e0, e1 = parallel.entity_range(n_ent, rank, world)
        slab = ent[e0:e1]
    findex = None
    if filter_triples is not None:
        findex = filter_triples if isinstance(filter_triples, FilterIndex) else FilterIndex(filter_triples)
    bounds = None
    prove_ties = False   # precision 2, contraction models: the prefilter form that also counts proven ties (set by the probe below)
    if sad:
        if isinstance(ent_f16, SadTables):
            sad_range, ent_u16 = ent_f16.range, ent_f16.ent_u16
        else:
            sad_range = D.eval_sad_range(ent, rel, k_int)
            ent_u16 = D.eval_sad_quantize(ent, k_int, sad_range)
    elif l2:
        tabs = ent_f16 if isinstance(ent_f16, L2Tables) else L2Tables(ent, k_int)
        ent_f16, bounds = tabs.ent_f16, tabs.bounds(e0, slab.shape[0])
    elif precision == 2:
        tabs = ent_f16 if isinstance(ent_f16, PrefilterTables) else None
        if tabs is not None:   # built once per evaluation run by the caller
            bounds, ent_f16 = tabs.bounds(e0, slab.shape[0]), tabs.ent_f16
        else:
            if ent_f16 is None:   # half-precision copy of the table for the prefilter
                ent_f16 = D.to_f16(ent, k_int, ld_dst=D.prefilter_ld(k_int))
            bounds = table_norm_bounds(slab, ent_f16[e0:e0 + slab.shape[0]], k_int)
        if cand is None and n >= _PROBE_TRIPLES and slab.shape[0] > 0 and os.environ.get("EMG_PREFILTER_PROBE", "1") != "0":
            key = (e0, slab.shape[0], side_mode)
            und = tabs.undecided.get(key) if tabs is not None else None
            if und is None:
                und = _prefilter_probe(model_id, ent, rel, slab, e0, k_int, scale, T, side_mode, ent_f16, bounds)
                if und > _PROBE_MAX_UNDECIDED and os.environ.get("EMG_PREFILTER_TIES", "1") != "0":
                    # too many undecided candidates — on a table whose scores are small against the comparison's quantum (a fresh model,
                    # the first epochs of a fit) they are TIES with the positive, which the second form of the prefilter proves as it
                    # proves the other two outcomes (emg_rank_bf16.hip MODE 4): probe it too
                    und_t = _prefilter_probe(model_id, ent, rel, slab, e0, k_int, scale, T, side_mode, ent_f16, bounds, ties=True)
                    if und_t <= _PROBE_MAX_UNDECIDED:
                        und = -1.0 - und_t   # (negative: "the ties form decides this table", remembered like the fraction)
                if tabs is not None:
                    tabs.undecided[key] = und
            prove_ties = und < 0.0
            if stats is not None:
                stats["probe_undecided"] = -1.0 - und if prove_ties else und
                stats["prove_ties"] = prove_ties
            if und > _PROBE_MAX_UNDECIDED:   # the band decides too little here: every tile of the call by the exact kernel
                precision = 0
                if stats is not None:
                    stats["fallback"] = stats.get("fallback", 0) + (n + query_chunk - 1) // query_chunk
    pending = []  # (counters on the device, nq) per chunk: every launch is asynchronous, ONE D2H at the end
    for c0 in range(0, n, query_chunk):
        Tc = T[c0:c0 + query_chunk]
        nq = Tc.shape[0]
        Tt = torch.from_numpy(Tc).to(ent.device)
        Q, pos_int = D.eval_build_queries(model_id, ent, rel, k_int, scale, Tt, side_mode)
        n_rows = Q.shape[0]
        cnt = torch.zeros((4, n_rows), dtype=torch.int32, device=ent.device)
        have_cands = cand.numel() > 0 if cand is not None else slab.shape[0] > 0
        if precision == 1:
            # bf16 MFMA throughput mode (statistical rank agreement; see emg_rank_bf16.hip)
            kp = D.bf16_ld(k_int)
            if ent_bf16 is None:
                ent_bf16 = D.to_bf16(ent, k_int, ld_dst=kp)
            Qb = D.to_bf16(Q, k_int, ld_dst=kp)
            # the true entity ties with itself by construction: pos_int comes from the same MFMA arithmetic
            pos_int, self_ent = D.eval_pos_int_bf16(model_id, ent_bf16, k_int, scale, Tt, side_mode, Qb)
            tab, off = (ent_bf16, 0) if cand is not None else (ent_bf16[e0:e0 + slab.shape[0]], e0)
            # 'worst' reads only #(>=), 'best' only #(>): one comparison per score in the kernel's epilogue
            need = {"worst": 1, "best": 2, "middle": 0}[strategy]
            count = lambda: D.eval_count_bf16(model_id, Qb, pos_int, self_ent, tab, k_int, scale, cnt[0], cnt[1],  # noqa: E731
                                              cand=cand, ent_offset=off, need=need)
            fcount = lambda fp_, fi_: D.eval_filter_count_bf16(model_id, Qb, pos_int, self_ent, tab, off, k_int,  # noqa: E731
                                                               scale, fp_, fi_, cnt[2], cnt[3])
        else:
            tab, off = (ent, 0) if cand is not None else (slab, e0)
            count = lambda Q=Q, pos_int=pos_int, cnt=cnt, tab=tab: D.eval_count(  # noqa: E731  (bound now: re-run at the end on overflow)
                model_id, Q, pos_int, tab, k_int, scale, cnt[0], cnt[1], cand=cand)
            fcount = lambda fp_, fi_: D.eval_filter_count(model_id, Q, pos_int, tab, off, k_int, scale, fp_, fi_,  # noqa: E731
                                                          cnt[2], cnt[3])
        pre = None    # (overflow flag, undecided pairs) of this tile on the device + what an exact re-run needs
        if sad and have_cands:
            Qu = D.eval_sad_quantize(Q, k_int, sad_range)
            thr = D.eval_sad_thresholds(pos_int, k_int, sad_range)
            n_seg = D.eval_sad_segments(n_rows, slab.shape[0])
            pairs, pcount = _pair_buffer(ent.device, n_seg)
            ev = _ev_start(stats)
            D.eval_prefilter_sad(Qu, thr, ent_u16[e0:e0 + slab.shape[0]], e0, k_int, cnt[0], pairs, pcount)
            D.eval_rescore_pairs(model_id, Q, pos_int, slab, e0, k_int, scale, pairs, pcount, n_seg, cnt[0], cnt[1], 4)
            _ev_stop(stats, ev)
            pre = (torch.stack([pcount[n_seg].long(), pcount[:n_seg].sum()]), count)
        elif l2 and have_cands:
            Qh, Q2 = D.to_f16_l2(Q, k_int, True)
            band = D.eval_prefilter_band(Q2, Qh, k_int, bounds[:3])
            thr = D.eval_l2_thresholds(Q, pos_int, band, bounds, k_int)
            n_seg = D.eval_prefilter_segments(n_rows, slab.shape[0], k_int + 2)
            pairs, pcount = _pair_buffer(ent.device, n_seg)
            ev = _ev_start(stats)
            try:
                D.eval_prefilter_f16_thr(Qh, thr, ent_f16[e0:e0 + slab.shape[0]], e0, k_int + 2, cnt[0], pairs, pcount)
            except L.EmgError:      # shape outside the register-stationary kernel: the exact kernel does this tile
                pre = None
            else:
                _rescore(model_id, Q, pos_int, slab, e0, k_int, scale, pairs, pcount, n_seg, cnt, D.prefilter_waves(k_int + 2), 32)
                _ev_stop(stats, ev)
                pre = (torch.stack([pcount[n_seg].long(), pcount[:n_seg].sum()]), count)
        elif precision == 2 and have_cands:
            # half-precision MFMA prefilter, then exact re-scoring of the undecided pairs: both asynchronous; whether a
            # wave ran out of pair room (-> this tile is redone by the exact kernel) is read with the counters at the end
            kp = D.prefilter_ld(k_int)
            Qb = D.to_f16(Q, k_int, ld_dst=kp)
            band = prefilter_band(Q, Qb, k_int, bounds)
            n_seg = D.eval_prefilter_segments(n_rows, slab.shape[0], k_int)
            pairs, pcount = _pair_buffer(ent.device, n_seg)
            ev = _ev_start(stats)
            try:
                D.eval_prefilter_f16(model_id, Qb, pos_int, band, ent_f16[e0:e0 + slab.shape[0]], e0, k_int, scale, cnt[0],
                                     pairs, pcount, cnt_eq=cnt[1] if prove_ties else None)
            except L.EmgError:      # shape outside the register-stationary kernel: the exact kernel does this tile
                pre = None
            else:
                _rescore(model_id, Q, pos_int, slab, e0, k_int, scale, pairs, pcount, n_seg, cnt, D.prefilter_waves(k_int), 32)
                _ev_stop(stats, ev)
                pre = (torch.stack([pcount[n_seg].long(), pcount[:n_seg].sum()]), count)
        if have_cands and pre is None:
            ev = _ev_start(stats)
            count()  # the big kernel goes first: the host-side filter CSR below is built underneath it
            _ev_stop(stats, ev)
        csr = findex.csr(Tc, side_mode, n_ent, subset_local) if findex is not None else None
        if csr is not None:
            ptr, idx = csr
            if have_cands:
                fcount(torch.from_numpy(ptr).to(ent.device, non_blocking=True),
                       torch.from_numpy(idx).to(ent.device, non_blocking=True))
        pending.append((cnt, nq, precision == 1 and strategy != "middle", pre))
    out = []
    for cnt, nq, single, pre in pending:
        if pre is not None:
            over, n_pairs = (int(v) for v in pre[0].cpu())
            if stats is not None:
                stats["pairs"] = stats.get("pairs", 0) + (0 if over else n_pairs)
                stats["fallback"] = stats.get("fallback", 0) + int(bool(over))
            if over:   # too many undecided candidates for the pair buffer: this tile again, by the exact kernel
                cnt[0:2].zero_()
                pre[1]()
        if world > 1:
            # range-sharded evaluation: the counters of this rank's candidate range are exact now (an overflowing tile has
            # been redone locally — a rank whose range did not overflow has nothing to redo), so the sum over the ranks is
            # taken HERE, after the overflow check: every rank issues the same collectives in the same order whatever
            # its own flags said (a fresh Glorot table leaves every candidate undecided: overflow is a normal state)
            parallel.allreduce_sum_(cnt)
        c = cnt.cpu().numpy().astype(np.int64)
        if single:
            c[1] = 0  # single-counter mode: c[0] already is what the strategy reads (see `need` above)
        out.append(ranks_from_counts(c[0], c[1], c[2], c[3], nq, corrupt_side, strategy))
    _ev_collect(stats)
    if not out:
        return np.zeros((0, 2) if corrupt_side == "s,o" else (0,), dtype=np.int64)
    return np.concatenate(out, axis=0)
