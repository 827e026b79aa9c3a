"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all.  Two shardings, chosen by where the path has a
real exchange step and by what xGMI can carry (7 links x ~153 GB/s per GPU  <<  8 TB/s HBM):

TRAINING — embedding-DIMENSION (k) sharding.  Rank r owns a column slab of BOTH tables (and of the optimizer
  state).  Every score is a sum over k, so each rank computes a PARTIAL score of every triple of the global
  batch from its slab; the only collective of a step is one all-reduce(sum) of [(1+eta)*B] fp32 partial
  scores (TransE-L2 reduces sums of squares, finished by -sqrt afterwards).  Loss and dL/dscore are then
  evaluated redundantly (they are O(B*eta) scalars) and each rank back-propagates into its own columns:
  NO gradient exchange, optimizer state sharded for free.
  Why not batch-sharding + gradient all-reduce (what a port of a data-parallel trainer would do): at eta=20
  almost every scored triple produces its own k-float gradient row, so the gradient volume equals the
  gather volume; moving it over xGMI (~1 TB/s aggregate) costs ~8x the HBM-bound compute it belongs to.
  A dense all-reduce of the |E|=1M table is 1.6 GB per step (~8 ms at RCCL bus bandwidth vs 0.6 ms of compute).
  Negatives are Philox draws keyed by (seed, epoch, batch, row): identical on every rank, no broadcast.

EVALUATION — candidate-RANGE sharding.  Every rank holds the (small: 1.6 GB at |E|=1M) table; rank r scores the
  query tile against entities [e0_r, e1_r) only and the int32 (>, ==) counters are all-reduced — 8 bytes per
  query row.  Ranks are exact because counts are integers.

Everything here is plumbing (slab arithmetic + collectives); the kernels are the same single-GPU ones.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def is_active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# ------------------------------------------------------------------------------------------------
# column (k) slabs
# ------------------------------------------------------------------------------------------------
def column_slabs(k, world):
    """Split k embedding dimensions into ``world`` contiguous slabs, as even as possible.
    Returns [(start, stop)] per rank.  For complex models the SAME slab is taken from the real and from the
    imaginary half (a rank needs matching re/im columns)."""
    base, rem = divmod(int(k), int(world))
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, s + n))
        s += n
    return out


def padded_width(n):
    """slab width rounded up to a multiple of 4 floats (16-byte rows for dwordx4 loads); the padding columns
    are zero and stay zero (their score terms and gradients are exactly 0 for every model)."""
    return ((int(n) + 3) // 4) * 4


def shard_columns(table, rank, world, is_complex):
    """[rows, k_int] -> this rank's [rows, k_int_local] slab (numpy or torch).  Complex layout stays
    [re_slab | pad | im_slab | pad] so that the kernels see an ordinary (smaller) ComplEx table."""
    k_int = table.shape[1]
    k = k_int // 2 if is_complex else k_int
    a, b = column_slabs(k, world)[rank]
    w = padded_width(b - a)
    zeros = (lambda r, c: np.zeros((r, c), dtype=table.dtype)) if isinstance(table, np.ndarray) else \
        (lambda r, c: torch.zeros((r, c), dtype=table.dtype, device=table.device))
    cat = np.concatenate if isinstance(table, np.ndarray) else torch.cat
    parts = []
    for h in range(2 if is_complex else 1):
        parts.append(table[:, h * k + a:h * k + b])
        if w > b - a:
            parts.append(zeros(table.shape[0], w - (b - a)))
    return cat(parts, 1)


def unshard_columns(slabs, k, is_complex):
    """inverse of shard_columns given every rank's slab (list ordered by rank) -> [rows, k_int] numpy"""
    world = len(slabs)
    rows = slabs[0].shape[0]
    k_int = 2 * k if is_complex else k
    out = np.zeros((rows, k_int), dtype=np.float32)
    for r, (a, b) in enumerate(column_slabs(k, world)):
        w = padded_width(b - a)
        sl = np.asarray(slabs[r])
        for h in range(2 if is_complex else 1):
            out[:, h * k + a:h * k + b] = sl[:, h * w:h * w + (b - a)]
    return out


def local_k_int(k, rank, world, is_complex):
    a, b = column_slabs(k, world)[rank]
    return padded_width(b - a) * (2 if is_complex else 1)


# ------------------------------------------------------------------------------------------------
# collectives
# ------------------------------------------------------------------------------------------------
def allreduce_sum_(t, group=None):
    """in-place all-reduce(sum); no-op without a process group"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if t.is_cuda and dist.get_backend(group) == "gloo":
            # test harness only (two ranks sharing one GPU): stage through the host
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def gather_slabs(local, group=None):
    """all_gather of equally-shaped... slabs may differ in width across ranks -> gather via object-free path:
    pad to the maximum width, all_gather, trim.  Returns a list of numpy arrays ordered by rank."""
    r, world = rank_world()
    if world == 1:
        return [local.detach().cpu().numpy()]
    if local.is_cuda and dist.get_backend(group) == "gloo":  # test harness only
        local = local.detach().cpu()
    width = torch.tensor([local.shape[1]], dtype=torch.int64, device=local.device)
    widths = [torch.zeros_like(width) for _ in range(world)]
    dist.all_gather(widths, width, group=group)
    wmax = int(max(int(w.item()) for w in widths))
    buf = torch.zeros((local.shape[0], wmax), dtype=local.dtype, device=local.device)
    buf[:, :local.shape[1]] = local
    buf = buf.contiguous()
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [o[:, :int(w.item())].cpu().numpy() for o, w in zip(outs, widths)]


# ------------------------------------------------------------------------------------------------
# candidate (entity) ranges for evaluation
# ------------------------------------------------------------------------------------------------
def entity_range(n, rank, world):
    """contiguous candidate range [e0, e1) of this rank"""
    base, rem = divmod(int(n), int(world))
    e0 = rank * base + min(rank, rem)
    return e0, e0 + base + (1 if rank < rem else 0)


# ------------------------------------------------------------------------------------------------
# BATCH sharding of the training step (tables replicated, sparse gradient exchange)
# ------------------------------------------------------------------------------------------------
# The split north_star names: every rank holds both tables, rank r takes rows [r0, r1) of each (global) batch,
# scores them with their negatives and produces one gradient row per (positive, role).  A table row is OWNED by
# the rank whose contiguous id range holds it (entity_range); gradient rows travel to their owner
# (all_to_all of (row id, global slot, gradient row), sparse: only touched rows move), the owner sums the rows
# of each destination IN GLOBAL SLOT ORDER — the order a single GPU sums them in, so the result does not depend
# on the number of ranks — applies the optimizer (its state exists only at the owner: entity-sharded optimizer
# state) and the updated rows are all-gathered into every replica.
#
# Per step and rank this moves over xGMI about (2 + eta) * B_local gradient rows out and the batch's unique
# touched rows in; for eta = 20 that is as many bytes as the scoring kernels read from HBM, which is why the
# k-sharding above (scores only) is the default wherever k is wide.  `Trainer(sharded="batch")` reports the bytes.

def _split_exchange(out, inp, out_splits, in_splits, group=None):
    """all_to_all_single with per-rank row counts; CUDA tensors over gloo (test harness: several ranks on one GPU)
    are staged through the host"""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(ho, inp.detach().cpu().contiguous(), out_splits, in_splits, group=group)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group)
    return out


def owner_bounds(n_rows, world):
    """first row id of every rank's owned range, plus n_rows: int64 [world + 1]"""
    return torch.tensor([entity_range(n_rows, r, world)[0] for r in range(world)] + [int(n_rows)], dtype=torch.int64)


def exchange_rows(dest, gslot, rows, n_rows, group=None):
    """Send every gradient row to the owner of its destination.

    dest  int32 [n]   destination row ids of this rank's contribution rows
    gslot int64 [n]   their slot numbers in the GLOBAL batch's contribution layout (defines the sum order)
    rows  float [n,k]
    Returns (dest_r int32 [m], gslot_r int64 [m], rows_r float [m,k], bytes_sent): all rows, from every rank, whose
    destination this rank owns (grouped by sending rank).  Host sync: one small count exchange."""
    rank, world = rank_world()
    k = rows.shape[1]
    order = torch.argsort(dest, stable=True)            # owners hold contiguous id ranges: sort = bucket
    d_sorted = dest.index_select(0, order)
    cuts = torch.searchsorted(d_sorted.to(torch.int64), owner_bounds(n_rows, world).to(dest.device))
    send_counts = (cuts[1:] - cuts[:-1]).cpu()
    if dist.get_backend(group) == "nccl":               # RCCL moves device buffers
        rc_dev = torch.empty(world, dtype=torch.int64, device=dest.device)
        dist.all_to_all_single(rc_dev, send_counts.to(dest.device), group=group)
        recv_counts = rc_dev.cpu()
    else:
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = [int(v) for v in send_counts], [int(v) for v in recv_counts]
    m = sum(rc)
    meta = torch.stack([d_sorted.to(torch.int64), gslot.index_select(0, order)], 1)
    meta_r = _split_exchange(torch.empty((m, 2), dtype=torch.int64, device=dest.device), meta, rc, sc, group)
    rows_r = _split_exchange(torch.empty((m, k), dtype=rows.dtype, device=rows.device), rows.index_select(0, order), rc, sc,
                             group)
    sent = (len(dest) - sc[rank]) * (k * rows.element_size() + 16)
    return meta_r[:, 0].to(torch.int32).contiguous(), meta_r[:, 1].contiguous(), rows_r, sent


def allgather_rows(ids, rows, group=None):
    """every rank contributes (ids int32 [u], rows float [u,k]) of the rows it owns and has just updated; returns the
    concatenation over the OTHER ranks (ids, rows) and the bytes received.  Variable u: padded to the largest."""
    rank, world = rank_world()
    k = rows.shape[1]
    cnt = torch.tensor([ids.numel()], dtype=torch.int64)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    if dist.get_backend(group) == "nccl":
        dc = [torch.zeros(1, dtype=torch.int64, device=ids.device) for _ in range(world)]
        dist.all_gather(dc, cnt.to(ids.device), group=group)
        cnts = [c.cpu() for c in dc]
    else:
        dist.all_gather(cnts, cnt, group=group)
    cnts = [int(c) for c in cnts]
    umax = max(cnts)
    stage = ids.is_cuda and dist.get_backend(group) == "gloo"
    dev = torch.device("cpu") if stage else ids.device
    pid = torch.zeros(umax, dtype=torch.int32, device=dev)
    prow = torch.zeros((umax, k), dtype=rows.dtype, device=dev)
    pid[:ids.numel()] = ids.to(dev)
    prow[:ids.numel()] = rows.to(dev)
    gid = [torch.empty_like(pid) for _ in range(world)]
    grow = [torch.empty_like(prow) for _ in range(world)]
    dist.all_gather(gid, pid, group=group)
    dist.all_gather(grow, prow, group=group)
    oid = torch.cat([gid[r][:cnts[r]] for r in range(world) if r != rank]).to(ids.device)
    orow = torch.cat([grow[r][:cnts[r]] for r in range(world) if r != rank]).to(rows.device)
    return oid, orow, int(oid.numel()) * (k * rows.element_size() + 4)


def batch_rows(B, rank, world):
    """rows [r0, r1) of a global batch of B positives that this rank scores"""
    return entity_range(B, rank, world)


# ------------------------------------------------------------------------------------------------
# BATCH sharding, device-resident (round 4): the same exchange with everything that does not depend on the tables settled
# in a METADATA phase — who sends how many rows to whom, which destinations each owner sums, the order it sums them in,
# the list of destinations every replica applies — so that the DATA phase between the scoring kernel and the apply is three
# collectives whose sizes are already on the host: rows -> owners (all_to_all), segmented sum at the owner, sums -> everybody
# (all_gather).  One host read per step (the W x W matrix of row counts), in the metadata phase, which the Trainer issues a
# step AHEAD on a side stream (the metadata needs the batch's positives and Philox draws, not the tables).  No torch.unique,
# no argsort: the bucket order is the local grouping's (emg_prepare_batch), the owner's order a keyed grouping
# (emg_group_dest_keyed); buffers are allocated once and grow only.
# ------------------------------------------------------------------------------------------------
def _staged(t, group):
    """test harness only (several ranks sharing one GPU over gloo): collectives of CUDA tensors go through the host"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_into(out, inp, group=None):
    """out (contiguous, world x the size of inp) <- every rank's inp, in rank order (flat views: gloo wants the concatenation shape)"""
    if _staged(inp, group):
        ho = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(ho, inp.detach().cpu().contiguous().view(-1), group=group)
        out.view(-1).copy_(ho)
    else:
        dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1), group=group)
    return out


class ExchangePlan:
    """what the metadata phase of one batch and table leaves behind (see RowExchange)"""
    __slots__ = ("n", "order", "sc", "rc", "m", "dest_o", "gslot_o", "cap_u", "uniq_local", "ids_all", "mine", "sent_bytes", "recv_bytes",
                 "rows_mine", "rows_all")


class RowExchange:
    """One table's side of the batch-sharded step.  Owners hold contiguous id ranges (entity_range)."""

    def __init__(self, n_rows, k, device, group=None):
        self.n_rows, self.k, self.device, self.group = int(n_rows), int(k), device, group
        self.rank, self.world = rank_world()
        self.e0, self.e1 = entity_range(self.n_rows, self.rank, self.world)
        self.bounds = owner_bounds(self.n_rows, self.world).to(device)
        self.owned = [entity_range(self.n_rows, r, self.world) for r in range(self.world)]
        self._recv_rows = None       # grow-only buffers of the data phase
        self._gathered = None

    # ---- metadata phase (table-independent) ----
    def plan_counts(self, dest_sorted, order, gslot):
        """dest_sorted int32 [n]: destinations of this rank's contribution slots, ascending; order [n]: the local slot at each sorted
        position (equal destinations in ascending slot order — the local grouping's keys / vals); gslot int32 [n]: the slot each
        local slot has in the GLOBAL batch's layout.  Exchanges the row counts (the step's one host read) and the (destination,
        global slot) pairs; returns the plan with dest_o (owner-local row ids) / gslot_o of the rows this rank will receive."""
        W = self.world
        pl = ExchangePlan()
        pl.n = int(dest_sorted.numel())
        pl.order = order.to(torch.int64)
        cuts = torch.searchsorted(dest_sorted.to(torch.int64), self.bounds)
        send_counts = (cuts[1:] - cuts[:-1]).to(torch.int64)
        C = torch.empty((W, W), dtype=torch.int64, device=send_counts.device)
        _all_gather_into(C, send_counts, self.group)
        Ch = C.cpu()                                              # <- the host read (metadata phase; a step ahead when planned ahead)
        pl.sc = [int(v) for v in Ch[self.rank]]
        pl.rc = [int(v) for v in Ch[:, self.rank]]
        pl.m = sum(pl.rc)
        totals = [int(Ch[:, p].sum()) for p in range(W)]
        bound = [min(self.owned[p][1] - self.owned[p][0], totals[p]) for p in range(W)]   # distinct destinations per owner, at most
        pl.cap_u = max(1, max(bound))
        pl.rows_mine, pl.rows_all = bound[self.rank], sum(bound)   # rows an optimizer updates: state sharded by owner / replicated
        meta = torch.stack([dest_sorted.to(torch.int32), gslot.index_select(0, pl.order).to(torch.int32)], 1)
        meta_r = _split_exchange(torch.empty((pl.m, 2), dtype=torch.int32, device=meta.device), meta, pl.rc, pl.sc, self.group)
        pl.dest_o = (meta_r[:, 0] - self.e0).contiguous()         # row ids inside this rank's range
        pl.gslot_o = meta_r[:, 1].contiguous()
        row_b = self.k * 4
        pl.sent_bytes = (pl.n - pl.sc[self.rank]) * (row_b + 8)
        pl.recv_bytes = (W - 1) * pl.cap_u * (row_b + 4)
        return pl

    def plan_unique(self, pl, keys_sorted_o):
        """keys_sorted_o int32 [m]: the received owner-local row ids, ascending (the owner grouping's keys).  Compacts the distinct
        ones (no host read: fixed capacity cap_u, padded with the sentinel n_rows) and all-gathers every owner's list:
        pl.ids_all int32 [W * cap_u] = the GLOBAL ids every replica applies a summed gradient to (sentinels are dropped by the
        grouping), pl.uniq_local int64 [cap_u] = this owner's rows (padding -> row 0)."""
        W, cap = self.world, pl.cap_u
        dev = keys_sorted_o.device
        uniq = torch.full((cap + 1,), self.n_rows, dtype=torch.int32, device=dev)          # [cap] = a bin for everything that is no head
        if pl.m > 0:
            k64 = keys_sorted_o.to(torch.int64)
            head = torch.ones(pl.m, dtype=torch.bool, device=dev)
            head[1:] = k64[1:] != k64[:-1]
            pos = torch.cumsum(head.to(torch.int64), 0) - 1
            uniq.index_copy_(0, torch.where(head, pos, torch.full_like(pos, cap)), (keys_sorted_o + self.e0).to(torch.int32))
            uniq[cap] = self.n_rows
        mine = uniq[:cap].contiguous()
        pl.uniq_local = torch.where(mine < self.n_rows, mine.to(torch.int64) - self.e0, torch.zeros_like(mine, dtype=torch.int64))
        ids_all = torch.empty(W * cap, dtype=torch.int32, device=dev)
        _all_gather_into(ids_all, mine, self.group)
        pl.ids_all = ids_all
        pl.mine = mine            # this owner's GLOBAL ids (padding: n_rows)
        return pl

    # ---- data phase ----
    def send_rows(self, pl, contrib):
        """gradient rows -> owners: returns float [m, k], the rows this rank owns the destinations of, in the order of pl.dest_o"""
        if self._recv_rows is None or self._recv_rows.shape[0] < pl.m:
            self._recv_rows = torch.empty((int(pl.m * 1.25) + 64, self.k), dtype=contrib.dtype, device=contrib.device)
        out = self._recv_rows[:pl.m]
        return _split_exchange(out, contrib.index_select(0, pl.order), pl.rc, pl.sc, self.group)

    def gather_sums(self, pl, sums_compact):
        """the owners' summed gradient rows float [cap_u, k] (rows of pl.uniq_local) -> everybody: float [W * cap_u, k], row j <-> pl.ids_all[j]"""
        need = self.world * pl.cap_u
        if self._gathered is None or self._gathered.shape[0] < need:
            self._gathered = torch.empty((int(need * 1.25) + 64, self.k), dtype=sums_compact.dtype, device=sums_compact.device)
        return _all_gather_into(self._gathered[:need], sums_compact, self.group)
