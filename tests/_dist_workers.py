"""Worker bodies for the multi-process tests (spawned with torch.multiprocessing; rendezvous on 127.0.0.1)."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32 = np.float32


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)


def run(fn, rank, world, port, out_dir, *args):
    """wrapper: exceptions of a rank end up in a file the parent asserts on"""
    import torch.distributed as dist
    try:
        _init(rank, world, port)
        fn(rank, world, out_dir, *args)
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    except Exception:  # noqa: BLE001
        open(os.path.join(out_dir, "fail_%d" % rank), "w").write(traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


# --------------------------------------------------------------------------------- CPU (gloo) workers
def _oracle_partial(model, E, R, x, k):
    """what one rank's forward kernel produces with EMG_SCORE_PARTIAL on its column slab"""
    from oracle import emgraph_oracle as orc
    es, ep, eo = orc.lookup_embeddings(E, R, x)
    if model == "TransE_L2":
        d = (es + ep) - eo
        return np.sum(d * d, axis=1, dtype=F32)
    if model == "HolE":
        return orc.fn_complex(es, ep, eo)  # unscaled
    return orc.score_fn(model, es, ep, eo, k=k)


def kshard_scores_worker(rank, world, out_dir):
    """k-slab partial scores + all-reduce + finish == full scores (every model, uneven slabs)"""
    import torch

    from emgraph_amd import parallel
    from oracle import emgraph_oracle as orc
    rs = np.random.RandomState(0)
    for model, k in (("TransE_L1", 10), ("TransE_L2", 7), ("DistMult", 9), ("ComplEx", 5), ("HolE", 6)):
        cplx = model in ("ComplEx", "HolE")
        ki = 2 * k if cplx else k
        E = (rs.randn(40, ki) * 0.4).astype(F32)
        R = (rs.randn(3, ki) * 0.4).astype(F32)
        x = np.stack([rs.randint(0, 40, 64), rs.randint(0, 3, 64), rs.randint(0, 40, 64)], 1)
        El = parallel.shard_columns(E, rank, world, cplx)
        Rl = parallel.shard_columns(R, rank, world, cplx)
        assert El.shape[1] == parallel.local_k_int(k, rank, world, cplx) and El.shape[1] % 4 == 0
        part = torch.from_numpy(_oracle_partial(model, El, Rl, x, k))
        parallel.allreduce_sum_(part)
        tot = part.numpy()
        if model == "TransE_L2":
            tot = -np.sqrt(tot)
        elif model == "HolE":
            tot = F32(2 / k) * tot
        np.testing.assert_allclose(tot, orc.score_triples(model, E, R, x, k=k), rtol=1e-5, atol=1e-6, err_msg=model)
        # gather + unshard is the inverse of shard
        back = parallel.unshard_columns(parallel.gather_slabs(torch.from_numpy(El)), k, cplx)
        np.testing.assert_array_equal(back, E)


def eval_counts_worker(rank, world, out_dir):
    """candidate-range sharding: per-range counts (C oracle) + all-reduce == unsharded counts -> same ranks"""
    import torch

    from emgraph_amd import parallel
    from emgraph_amd.evaluation import build_filter_csr, ranks_from_counts
    from oracle import c_oracle as co
    rs = np.random.RandomState(1)
    k, n_ent, n_rel, nq = 8, 101, 3, 9
    E = (rs.randn(n_ent, 2 * k) * 0.4).astype(F32)
    R = (rs.randn(n_rel, 2 * k) * 0.4).astype(F32)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, n_rel, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 500), rs.randint(0, n_rel, 500), rs.randint(0, n_ent, 500)], 1)])
    Q, pos_int = co.build_queries(3, E, R, 2 * k, 1.0, T, 3)
    ptr, idx = build_filter_csr(F, T, 3, n_ent)
    e0, e1 = parallel.entity_range(n_ent, rank, world)
    ranges = [parallel.entity_range(n_ent, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_ent and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    gt, eq = co.count(3, Q, pos_int, E[e0:e1], 2 * k, 1.0)
    fgt, feq = co.filter_count(3, Q, pos_int, E[e0:e1], e0, 2 * k, 1.0, ptr, idx)
    cnt = torch.from_numpy(np.stack([gt, eq, fgt, feq]))
    parallel.allreduce_sum_(cnt)
    c = cnt.numpy()
    fgt0, feq0 = co.filter_count(3, Q, pos_int, E, 0, 2 * k, 1.0, ptr, idx)
    gt0, eq0 = co.count(3, Q, pos_int, E, 2 * k, 1.0)
    np.testing.assert_array_equal(c, np.stack([gt0, eq0, fgt0, feq0]))
    np.testing.assert_array_equal(ranks_from_counts(c[0], c[1], c[2], c[3], nq, "s,o", "worst"),
                                  ranks_from_counts(gt0, eq0, fgt0, feq0, nq, "s,o", "worst"))


# --------------------------------------------------------------------------------- GPU worker (2 ranks, one GPU)
def sharded_fit_worker(rank, world, out_dir, name, loss, opt):
    """the REAL k-sharded training + range-sharded evaluation path (HIP kernels), two ranks sharing cuda:0"""
    import torch

    from emgraph_amd.evaluation import evaluate_performance
    from emgraph_amd import models
    torch.cuda.set_device(0)
    rs = np.random.RandomState(5)
    n_ent, n_rel = 80, 4
    X = np.stack([rs.randint(0, n_ent, 900), rs.randint(0, n_rel, 900), rs.randint(0, n_ent, 900)], 1)
    X[:n_ent, 0] = np.arange(n_ent)
    X[:n_rel, 1] = np.arange(n_rel)
    kw = dict(k=10, eta=3, epochs=2, batches_count=3, seed=3, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05})
    if name == "TransE_L2":
        m = models.TransE(embedding_model_params={"norm": 2}, **kw)
    else:
        m = getattr(models, name)(**kw)
    m.fit(X[:800])
    ranks = evaluate_performance(X[800:], m, filter_triples=X, corrupt_side="s,o")
    ranks_sub = evaluate_performance(X[800:840], m, filter_triples=X, corrupt_side="s+o", entities_subset=list(range(0, 80, 3)))
    extra = {}
    if name in ("ComplEx", "DistMult", "HolE"):
        # bf16 MFMA mode, range-sharded: integer counters, so the 2-rank result must equal the 1-rank one exactly
        from emgraph_amd import _lib as L
        from emgraph_amd.evaluation import rank_triples_device
        E, R = m.trained_model_params
        k_int = E.shape[1]
        sc = 2.0 / 10 if name == "HolE" else 1.0
        Xi = X[800:].astype(np.int32)
        from emgraph_amd.training import alloc_table
        dev0 = torch.device("cuda")
        ent, rel = alloc_table(E.shape[0], k_int, dev0, init=E), alloc_table(R.shape[0], k_int, dev0, init=R)
        mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "HolE": L.HOLE}[name]
        extra["ranks_bf16"] = rank_triples_device(mid, ent, rel, k_int, sc, Xi, "s,o", "worst", filter_triples=X.astype(np.int32),
                                                  precision=1, shard=(rank, world))
        extra["ranks_bf16_single"] = rank_triples_device(mid, ent, rel, k_int, sc, Xi, "s,o", "worst",
                                                         filter_triples=X.astype(np.int32), precision=1)
    np.savez(os.path.join(out_dir, "res_%d.npz" % rank), E=m.trained_model_params[0], R=m.trained_model_params[1],
             ranks=ranks, ranks_sub=ranks_sub, pred=m.predict(X[800:], sharded=True), pred_local=m.predict(X[800:]), **extra)


def batch_exchange_worker(rank, world, out_dir):
    """BATCH sharding plumbing on CPU tensors (gloo): every rank holds the same global batch of contribution rows,
    keeps the rows of ITS positives, sends them to the owners of their destinations, owners sum each destination's
    rows in GLOBAL SLOT order (sequential fp32, what emg_apply_grouped does) and apply SGD, updated rows are
    all-gathered.  Every replica must end BIT-identical to the single-process result, whatever the rank count."""
    import torch

    from emgraph_amd import parallel
    rs = np.random.RandomState(11)
    n_rows, k, B, roles, lr = 37, 6, 23, 5, F32(0.1)     # uneven everything: 23 positives over `world` ranks, 37 rows
    W0 = rs.randn(n_rows, k).astype(F32)
    dest_g = rs.randint(0, n_rows - 4, roles * B).astype(np.int32)          # rows 33..36 are never touched
    dest_g[rs.choice(roles * B, 40, replace=False)] = 7                      # a hot destination
    rows_g = rs.randn(roles * B, k).astype(F32)
    # single process: stable grouping by destination keeps slot order; sequential fp32 sums
    exp = W0.copy()
    order = np.argsort(dest_g, kind="stable")
    for seg in np.split(order, np.flatnonzero(np.diff(dest_g[order])) + 1):
        g = np.zeros(k, F32)
        for i in seg:
            g = g + rows_g[i]
        exp[dest_g[seg[0]]] = W0[dest_g[seg[0]]] - lr * g
    # this rank's share: positives [r0, r1) of every role block
    r0, r1 = parallel.batch_rows(B, rank, world)
    Bl = r1 - r0
    t = np.arange(roles * Bl)
    gslot = (t // max(Bl, 1)) * B + r0 + (t % max(Bl, 1))
    dest_o, gslot_o, rows_o, sent = parallel.exchange_rows(torch.from_numpy(dest_g[gslot]), torch.from_numpy(gslot.astype(np.int64)),
                                                           torch.from_numpy(rows_g[gslot]), n_rows)
    e0, e1 = parallel.entity_range(n_rows, rank, world)
    d = dest_o.numpy()
    assert ((d >= e0) & (d < e1)).all()                                      # only rows this rank owns arrive
    W = W0.copy()
    perm = np.argsort(gslot_o.numpy(), kind="stable")
    dp = d[perm]
    order = np.argsort(dp, kind="stable")
    upd = []
    if len(dp):
        for seg in np.split(order, np.flatnonzero(np.diff(dp[order])) + 1):
            g = np.zeros(k, F32)
            for i in seg:
                g = g + rows_o.numpy()[perm[i]]
            W[dp[seg[0]]] = W0[dp[seg[0]]] - lr * g
            upd.append(dp[seg[0]])
    upd = np.array(upd, dtype=np.int32)
    oid, orow, recvd = parallel.allgather_rows(torch.from_numpy(upd), torch.from_numpy(W[upd] if len(upd) else np.zeros((0, k), F32)))
    W[oid.numpy()] = orow.numpy()
    np.testing.assert_array_equal(W, exp)
    assert sent == (len(gslot) - int(((dest_g[gslot] >= e0) & (dest_g[gslot] < e1)).sum())) * (4 * k + 16)
    np.save(os.path.join(out_dir, "W_%d.npy" % rank), W)


def row_exchange_worker(rank, world, out_dir):
    """the DEVICE-RESIDENT form of the same exchange (parallel.RowExchange, round 4) on CPU tensors over gloo: metadata phase
    (row counts, (destination, global slot) pairs, every owner's distinct destinations, padded to one capacity) then data phase
    (rows -> owners, sums -> everybody).  numpy stands in for the two groupings the HIP library does (the local one by destination,
    the owner's by (destination, global slot)) and for the sequential fp32 sums.  Bit-identical to the single-process result for
    any rank count; ranks with nothing to send or nothing to own included."""
    import torch

    from emgraph_amd import parallel
    rs = np.random.RandomState(11)
    n_rows, k, B, roles, lr = 37, 6, 23, 5, F32(0.1)
    W0 = rs.randn(n_rows, k).astype(F32)
    dest_g = rs.randint(0, n_rows - 4, roles * B).astype(np.int32)
    dest_g[rs.choice(roles * B, 40, replace=False)] = 7
    rows_g = rs.randn(roles * B, k).astype(F32)
    exp = W0.copy()
    order = np.argsort(dest_g, kind="stable")
    for seg in np.split(order, np.flatnonzero(np.diff(dest_g[order])) + 1):
        g = np.zeros(k, F32)
        for i in seg:
            g = g + rows_g[i]
        exp[dest_g[seg[0]]] = W0[dest_g[seg[0]]] - lr * g
    for trial in range(2):      # twice through the same object: its buffers are reused
        r0, r1 = parallel.batch_rows(B, rank, world)
        Bl = r1 - r0
        t = np.arange(roles * Bl)
        gslot = ((t // max(Bl, 1)) * B + r0 + (t % max(Bl, 1))).astype(np.int32)
        dest_l, rows_l = dest_g[gslot], rows_g[gslot]
        if trial == 0:
            x = parallel.RowExchange(n_rows, k, torch.device("cpu"))
        lo = np.argsort(dest_l, kind="stable")                                  # = the local grouping (emg_prepare_batch)
        pl = x.plan_counts(torch.from_numpy(dest_l[lo]), torch.from_numpy(lo), torch.from_numpy(gslot))
        d_o, g_o = pl.dest_o.numpy(), pl.gslot_o.numpy()
        assert ((d_o >= 0) & (d_o < x.e1 - x.e0)).all() and pl.m == len(d_o)
        oo = np.lexsort((g_o, d_o))                                              # = emg_group_dest_keyed: by destination, then global slot
        x.plan_unique(pl, torch.from_numpy(d_o[oo].astype(np.int32)))
        ids_all = pl.ids_all.numpy()
        assert len(ids_all) == world * pl.cap_u
        recv = x.send_rows(pl, torch.from_numpy(rows_l)).numpy()
        sums = np.full((max(1, x.e1 - x.e0), k), np.nan, F32)                   # (stale content must not matter)
        sums[pl.uniq_local.numpy()] = 0
        if pl.m:
            for seg in np.split(oo, np.flatnonzero(np.diff(d_o[oo])) + 1):
                g = np.zeros(k, F32)
                for i in seg:
                    g = g + recv[i]
                sums[d_o[seg[0]]] = g
        gathered = x.gather_sums(pl, torch.from_numpy(sums[pl.uniq_local.numpy()])).numpy()
        W = W0.copy()
        seen = set()
        for j, rid in enumerate(ids_all):
            if rid < n_rows:
                assert rid not in seen
                seen.add(int(rid))
                W[rid] = W0[rid] - lr * gathered[j]
        np.testing.assert_array_equal(W, exp)
        assert seen == set(np.unique(dest_g).tolist())
    np.save(os.path.join(out_dir, "W_%d.npy" % rank), W)


def batch_sharded_fit_worker(rank, world, out_dir, name, loss, opt, reg=None, shard_state=False):
    """the REAL batch-sharded training step (HIP kernels + exchange), two ranks sharing cuda:0 over gloo"""
    import torch

    from emgraph_amd import models
    torch.cuda.set_device(0)
    rs = np.random.RandomState(5)
    n_ent, n_rel = 80, 4
    X = np.stack([rs.randint(0, n_ent, 900), rs.randint(0, n_rel, 900), rs.randint(0, n_ent, 900)], 1)
    X[:n_ent, 0] = np.arange(n_ent)
    X[:n_rel, 1] = np.arange(n_rel)
    kw = dict(k=10, eta=3, epochs=2, batches_count=3, seed=3, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05})
    if reg is not None:
        kw.update(regularizer="LP", regularizer_params=reg)
    emp = {"sharding": "batch", "shard_state": shard_state}
    if name == "TransE_L2":
        m = models.TransE(embedding_model_params=dict(emp, norm=2), **kw)
    else:
        m = getattr(models, name)(embedding_model_params=emp, **kw)
    m.fit(X[:803])       # 803 rows / 3 batches = 268 per batch (last one 267): odd splits over two ranks
    np.savez(os.path.join(out_dir, "res_%d.npz" % rank), E=m.trained_model_params[0], R=m.trained_model_params[1],
             pred=m.predict(X[800:]), xgmi=np.array(m._trainer.xgmi_bytes), losses=np.array(m.epoch_losses),
             opt_rows=np.array(m._trainer.opt_rows),
             state_rows=np.array(-1 if m._trainer.state_ent[0] is None else m._trainer.state_ent[0].shape[0]))


def sharded_overflow_worker(rank, world, out_dir):
    """range-sharded exact-fast ranking whose pair buffer OVERFLOWS on every rank (a tiny-norm table: every candidate's
    comparison integer is 0, every candidate is undecided): each rank redoes its tile with the exact kernel before the
    counters are summed, and the ranks equal the single-process exact ranks"""
    import torch

    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import rank_triples_device
    from emgraph_amd.training import alloc_table
    torch.cuda.set_device(0)
    rs = np.random.RandomState(2)
    n_ent, n_rel, k = 40000, 5, 64
    E = (rs.randn(n_ent, 2 * k) * 1e-4).astype(F32)      # scores ~1e-7: int(score * 1e5) == 0 for every candidate
    R = (rs.randn(n_rel, 2 * k) * 1e-4).astype(F32)
    dev0 = torch.device("cuda")
    ent, rel = alloc_table(n_ent, 2 * k, dev0, init=E), alloc_table(n_rel, 2 * k, dev0, init=R)
    T = np.stack([rs.randint(0, n_ent, 160), rs.randint(0, n_rel, 160), rs.randint(0, n_ent, 160)], 1).astype(np.int32)
    st, st_t = {}, {}
    os.environ["EMG_PREFILTER_TIES"] = "0"      # (round 6: the prefilter's second form PROVES such ties; off here: the overflow path is what this test is about)
    got = rank_triples_device(L.COMPLEX, ent, rel, 2 * k, 1.0, T, "s,o", "worst", filter_triples=T, precision=2,
                              shard=(rank, world), stats=st)
    os.environ.pop("EMG_PREFILTER_TIES")
    auto = rank_triples_device(L.COMPLEX, ent, rel, 2 * k, 1.0, T, "s,o", "worst", filter_triples=T, precision="auto",
                               shard=(rank, world))
    ties = rank_triples_device(L.COMPLEX, ent, rel, 2 * k, 1.0, T, "s,o", "worst", filter_triples=T, precision=2,
                               shard=(rank, world), stats=st_t)      # ... and on: the same ranks, no tile redone
    want = rank_triples_device(L.COMPLEX, ent, rel, 2 * k, 1.0, T, "s,o", "worst", filter_triples=T, precision=0)
    np.savez(os.path.join(out_dir, "res_%d.npz" % rank), got=got, auto=auto, want=want, ties=ties, fallback=st.get("fallback", 0),
             ties_fallback=st_t.get("fallback", 0), ties_taken=int(bool(st_t.get("prove_ties"))))
