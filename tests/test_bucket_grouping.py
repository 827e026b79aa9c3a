"""The BUCKET form of emg_prepare_batch (csrc/emg_group_bucket.hip: tables of more than 131072 rows; default from 262144) against the counting form
(EMG_GROUPING=count) on the same inputs: Philox codes, destination arrays, sorted keys, the stable order, singleton flags,
factored source rows and positions must be IDENTICAL, the segment descriptors (singleton / segment / block-task lists) equal as
sets — and an apply from either grouping gives the same table bits.  Shapes: C3's own (1M x 1k, B 16384, eta 20), hub rows,
a restricted corruption pool (a few rows hit thousands of times: rows longer than 32 and buckets beyond the LDS capacity),
every bucket through the global-memory form (EMG_BUCKET_CAP), ids outside the table, rows of a larger (sharded) batch,
injected draws.  Reference behaviour being restated: the sparse gradient aggregation of EmbeddingModel.py:1388-1440."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32 = np.float32


def dev():
    from emgraph_amd import device as d
    d.require_gpu()
    return d


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def al(x):
    return (x + 255) // 256 * 256


def regions(ws, N):
    """views of a grouping workspace (csrc/emg_group.hip::layout_impl)"""
    kb = al(4 * N)
    at = 0
    out = {}
    w = ws.cpu().numpy()
    for name in ("keys", "vals", "tmpv", "srcrow", "pos_of_slot", "coef"):
        out[name] = w[at:at + 4 * N].view(np.float32 if name == "coef" else np.uint32)
        at += kb
    n_multi = N // 2 + 1
    out["multi"] = w[at:at + 12 * n_multi].view(np.uint32).reshape(-1, 3)
    at += al(12 * n_multi)
    out["single"] = w[at:at + 4 * N].view(np.uint32)
    at += kb
    task_cap = N // 8 + 2
    out["tasks"] = w[at:at + 12 * task_cap].view(np.uint32).reshape(-1, 3)
    at += al(12 * task_cap)
    out["arrive"] = w[at:at + 4 * (N // 64 + 2)].view(np.int32)
    at += al(4 * (N // 64 + 2))
    out["counters"] = w[at:at + 256].view(np.uint32)
    return out


def run_prepare(d, mode, pos, eta, sides, n_ent, n_rel, factored, cap=None, **kw):
    B = pos.shape[0]
    et = eta * len(sides)
    n_ce = (2 + et) * B
    os.environ["EMG_GROUPING"] = mode
    if cap:
        os.environ["EMG_BUCKET_CAP"] = str(cap)
    try:
        codes = torch.full((B * et,), -7, dtype=torch.int32, device="cuda")
        de = torch.full((n_ce,), -7, dtype=torch.int32, device="cuda")
        dr = torch.full((B,), -7, dtype=torch.int32, device="cuda")
        we = torch.zeros(d.apply_workspace_bytes(n_ce, n_ent), dtype=torch.uint8, device="cuda")
        wr = torch.zeros(d.apply_workspace_bytes(B, n_rel), dtype=torch.uint8, device="cuda")
        flags = torch.full((n_ce,), 9, dtype=torch.uint8, device="cuda")
        d.prepare_batch(cu(pos), eta, list(sides), kw.pop("n_choices", n_ent), codes, de, dr, n_ent, n_rel, we, wr, seed=5, counter0=3,
                        single_flags=flags, factored=factored, **kw)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("EMG_GROUPING", None)
        os.environ.pop("EMG_BUCKET_CAP", None)
    return dict(codes=codes.cpu().numpy(), de=de.cpu().numpy(), dr=dr.cpu().numpy(), flags=flags.cpu().numpy(),
                we=we, wr=wr, E=regions(we, n_ce), R=regions(wr, B), n_ce=n_ce)


def compare(a, b, B, factored, n_ent, n_rel):
    for k in ("codes", "de", "dr"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for tab, n, rows, dest in (("E", a["n_ce"], n_ent, a["de"]), ("R", B, n_rel, a["dr"])):
        x, y = a[tab], b[tab]
        valid = (dest >= 0) & (dest < rows)
        nv = int(valid.sum())
        assert x["counters"][3] == nv == y["counters"][3], (tab, "GC_VALID")
        order = np.argsort(np.where(valid, dest, np.iinfo(np.int32).max), kind="stable")[:nv]
        np.testing.assert_array_equal(y["keys"][:nv], dest[order].astype(np.uint32), err_msg=tab + " keys vs numpy")
        np.testing.assert_array_equal(y["vals"][:nv], order.astype(np.uint32), err_msg=tab + " vals vs numpy")
        np.testing.assert_array_equal(x["keys"][:nv], y["keys"][:nv], err_msg=tab + " keys")
        np.testing.assert_array_equal(x["vals"][:nv], y["vals"][:nv], err_msg=tab + " vals")
        for c in range(3):   # list lengths
            assert x["counters"][c] == y["counters"][c], (tab, "counter", c, x["counters"][:4], y["counters"][:4])
        assert x["counters"][8] == 0 and y["counters"][8] == 0 and not x["arrive"].any() and not y["arrive"].any()
        nm, ns, nt = (int(v) for v in x["counters"][:3])
        for name, m in (("multi", nm), ("tasks", nt)):
            ax, ay = x[name][:m], y[name][:m]
            np.testing.assert_array_equal(ax[np.lexsort(ax.T[::-1])], ay[np.lexsort(ay.T[::-1])], err_msg=tab + " " + name)
        np.testing.assert_array_equal(np.sort(x["single"][:ns]), np.sort(y["single"][:ns]), err_msg=tab + " single")
        cnt = np.bincount(dest[valid], minlength=rows)
        assert ns == int((cnt == 1).sum()) and nm == int(((cnt >= 2) & (cnt <= 32)).sum())
    np.testing.assert_array_equal(a["flags"], b["flags"], err_msg="flags")
    if factored:
        x, y = a["E"], b["E"]
        nv = int(x["counters"][3])
        np.testing.assert_array_equal(x["srcrow"][:nv], y["srcrow"][:nv], err_msg="srcrow")
        n_neg = a["n_ce"] - 2 * B
        np.testing.assert_array_equal(x["pos_of_slot"][:n_neg], y["pos_of_slot"][:n_neg], err_msg="pos_of_slot")
        so = x["vals"][:nv] < 2 * B
        np.testing.assert_array_equal(x["coef"][:nv][so], y["coef"][:nv][so], err_msg="coef of the subject / object positions")


CASES = {
    # name: B, eta, sides, n_ent, n_rel, what
    "c3_shape": (16384, 20, (2,), 1_000_000, 1000, "uniform"),
    "two_sides": (4096, 5, (0, 1), 200_000, 300, "uniform"),
    "hubs": (16384, 6, (2,), 500_000, 50, "zipf"),
    "restricted_pool": (8192, 10, (2,), 300_000, 7, "pool"),
    "tiny_batch_big_table": (37, 1, (1,), 1_000_000, 3, "uniform"),      # (the largest table the counting layout takes for 111 rows)
    "one_relation": (5000, 2, (2,), 140_000, 1, "uniform"),
    "dense_small_rows": (20000, 30, (2,), 131_073, 2000, "uniform"),      # the smallest table the bucket grouping takes
    "big_batch_wide_chunks": (70000, 12, (2,), 400_000, 100, "uniform"),      # 980 k contributions: chunks of 2048 slots
}


@pytest.mark.parametrize("factored", [False, True])
@pytest.mark.parametrize("case", sorted(CASES))
def test_bucket_grouping_equals_counting_grouping(case, factored):
    d = dev()
    B, eta, sides, n_ent, n_rel, what = CASES[case]
    rs = np.random.RandomState(len(case) * 7 + B)
    if what == "zipf":
        wts = 1.0 / np.arange(1, n_ent + 1)
        perm = rs.permutation(n_ent)
        s, o = perm[rs.choice(n_ent, B, p=wts / wts.sum())], perm[rs.choice(n_ent, B, p=wts / wts.sum())]
        s[: B // 2] = perm[0]           # one subject in half of the batch: a row of 8192 + contributions
    else:
        s, o = rs.randint(0, n_ent, B), rs.randint(0, n_ent, B)
    pos = np.stack([s, rs.randint(0, n_rel, B), o], 1).astype(np.int32)
    kw = {}
    if what == "pool":     # negatives from 40 entities of one neighbourhood: ~2000 contributions per row, one bucket holds them all
        kw = dict(entities_list=cu((123_000 + 3 * np.arange(40)).astype(np.int32)), n_choices=40)
    ref = run_prepare(d, "count", pos, eta, sides, n_ent, n_rel, factored, **dict(kw))
    got = run_prepare(d, "bucket", pos, eta, sides, n_ent, n_rel, factored, **dict(kw))
    compare(got, ref, B, factored, n_ent, n_rel)
    if case in ("c3_shape", "hubs", "two_sides"):       # every bucket through the global-memory form
        got2 = run_prepare(d, "bucket", pos, eta, sides, n_ent, n_rel, factored, cap=48, **dict(kw))
        compare(got2, ref, B, factored, n_ent, n_rel)


def test_bucket_grouping_drops_ids_outside_the_table_and_follows_sharded_and_injected_draws():
    d = dev()
    rs = np.random.RandomState(11)
    B, eta, n_ent, n_rel = 3000, 4, 150_000, 20
    pos = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
    pos[::17, 0] = n_ent + 5          # ids outside the table (subject / relation): dropped, their flags 0
    pos[::29, 1] = n_rel + 1
    ref = run_prepare(d, "count", pos, eta, (2,), n_ent, n_rel, False)
    got = run_prepare(d, "bucket", pos, eta, (2,), n_ent, n_rel, False)
    compare(got, ref, B, False, n_ent, n_rel)
    pos = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
    ref = run_prepare(d, "count", pos, eta, (2,), n_ent, n_rel, True, B_global=4 * B, row_offset=B)
    got = run_prepare(d, "bucket", pos, eta, (2,), n_ent, n_rel, True, B_global=4 * B, row_offset=B)
    compare(got, ref, B, True, n_ent, n_rel)
    inj_repl = cu(rs.randint(0, n_ent, B * eta).astype(np.int32))
    inj_mask = cu(rs.randint(0, 2, B * eta).astype(np.int32))
    ref = run_prepare(d, "count", pos, eta, (2,), n_ent, n_rel, True, inj_repl=inj_repl, inj_mask=inj_mask)
    got = run_prepare(d, "bucket", pos, eta, (2,), n_ent, n_rel, True, inj_repl=inj_repl, inj_mask=inj_mask)
    compare(got, ref, B, True, n_ent, n_rel)


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_apply_from_the_bucket_grouping_gives_the_counting_grouping_bits(opt):
    """the descriptor-driven apply (segments, singletons, block tasks) from either grouping: same table and state bits"""
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(3)
    B, eta, n_ent, n_rel, k = 6000, 8, 150_000, 11, 72
    pos = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
    pos[:3000, 0] = 777                # a row of 3000 contributions: block tasks
    n_ce = (2 + eta) * B
    contrib = cu(rs.randn(n_ce, k).astype(F32))
    W0 = rs.randn(n_ent, k).astype(F32)
    out = {}
    for mode in ("count", "bucket"):
        g = run_prepare(d, mode, pos, eta, (2,), n_ent, n_rel, False)
        # a workspace with room for the long-segment partial sums: group again into it
        os.environ["EMG_GROUPING"] = mode
        try:
            ws = torch.zeros(d.apply_workspace_bytes(n_ce, n_ent, k), dtype=torch.uint8, device="cuda")
            wr = torch.zeros(d.apply_workspace_bytes(B, n_rel, k), dtype=torch.uint8, device="cuda")
            codes = torch.empty(B * eta, dtype=torch.int32, device="cuda")
            de = torch.empty(n_ce, dtype=torch.int32, device="cuda")
            dr = torch.empty(B, dtype=torch.int32, device="cuda")
            d.prepare_batch(cu(pos), eta, [2], n_ent, codes, de, dr, n_ent, n_rel, ws, wr, seed=5, counter0=3)
        finally:
            os.environ.pop("EMG_GROUPING", None)
        np.testing.assert_array_equal(de.cpu().numpy(), g["de"])
        W = cu(W0.copy())
        st = [torch.full_like(W, 0.1), None] if opt == "adagrad" else [None, None]
        oid = L.OPT_ADAGRAD if opt == "adagrad" else L.OPT_SGD
        d.apply_grouped(oid, W, k, st[0], st[1], None, 1, contrib, n_ce, 0, (0.01, 0.0, 0.9, 0.999, 1e-7, 0.01), ws)
        torch.cuda.synchronize()
        out[mode] = [W.cpu().numpy()] + [s.cpu().numpy() for s in st if s is not None]
    for x, y in zip(out["count"], out["bucket"]):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    assert not np.array_equal(out["count"][0], W0)


@pytest.mark.parametrize("model,opt,deferred,reg", [("ComplEx", "sgd", None, None), ("ComplEx", "adam", False, None), ("ComplEx", "adam", True, None),
                                                    ("DistMult", "adagrad", None, None), ("TransE", "momentum", None, None),
                                                    ("ComplEx", "sgd", True, {"lambda": 1e-3, "p": 2}), ("TransE", "adagrad", True, {"lambda": 1e-3, "p": 3})])
def test_fit_on_a_large_table_gives_the_same_bits_under_either_grouping(monkeypatch, model, opt, deferred, reg):
    """the plan's step on a table of more than 131072 rows (bucket grouping) with a SHORT last batch — the workspaces are laid out for
    the plan's capacity (layout_B > B) —: tables, optimizer state and loss equal the counting grouping's, for the in-place forms
    (SGD; SGD + LP with its replay, form 7), Keras Adam with its dense pass and under the deferred pass (catch-up + window form walk
    the grouping's descriptor lists), Adagrad / momentum through the apply, factored (bilinear) and full-row (TransE) contributions.
    (Adam at 70 000 rows is how the overlap with the dense-pass-inside-the-apply was found: that size keeps the counting grouping.)"""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    dev()
    rs = np.random.RandomState(9)
    n_ent, n_rel, k, eta, nb = 140_000, 6, 36, 7, 3
    cplx = model == "ComplEx"
    ki = 2 * k if cplx else k
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, 1000), rs.randint(0, n_rel, 1000), rs.randint(0, n_ent, 1000)], 1).astype(np.int32)
    X[:300, 0] = 4242                     # a hub row: block tasks
    kw = dict(regularizer="LP", regularizer_params=reg) if reg else {}

    def run(mode):
        monkeypatch.setenv("EMG_GROUPING", mode)
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="pairwise" if model == "TransE" else "nll", optimizer=opt, optimizer_params={"lr": 0.01},
                     batches_count=nb, seed=2, deferred_dense=deferred, **kw)
        tr.set_training_set(X, 334)
        for ep in (1, 2):
            for b, (s, n) in enumerate(((0, 334), (334, 333), (667, 333))):
                tr.step(s, n, epoch=ep, batch=b + 1, prefetch=[((s + n) % 1000, 333 if b < 2 else 334, ep + (b == 2), (b + 1) % 3 + 1)])
        loss = tr.read_loss()
        Et, Rt = tr.tables_numpy()
        return Et, Rt, [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None], loss

    a, b = run("bucket"), run("count")
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    if reg:
        assert a[3] == pytest.approx(b[3], rel=1e-8)      # (the regulariser's value: FLOAT partial sums per wave added as doubles, and the partition
        #  of the rows over the waves follows the grouping's lists: 1.6e-11 apart in the round-6 soak, once past 1e-12 in a full-suite run)
    else:
        assert a[3] == b[3]
    assert not np.array_equal(a[0], E0)


def test_standalone_relation_apply_with_adams_dense_pass_after_a_bucket_grouping():
    """Keras Adam's dense pass INSIDE the descriptor-driven apply (emg_apply.hip: dense_here, tables of <= 131072 rows) finds the
    untouched rows in the grouping's offset array.  emg_prepare_batch's bucket form groups BOTH tables whenever the ENTITY table is
    large, so the relation table's offsets must be written by it too (round 5 advisor: they were not — a standalone apply of the
    relation table then read whatever the workspace held).  The workspaces start as garbage outside their control region, as the
    product's torch.empty ones do; bits must equal the counting grouping's and every untouched row must have decayed exactly once.
    Reference: Keras Adam updates every row of a table each step (training/adam.py:31-48)."""
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(21)
    B, eta, n_ent, k = 5000, 3, 300_000, 40
    for n_rel in (700, 5, 3000):
        pos = np.stack([rs.randint(0, n_ent, B), rs.randint(0, max(1, n_rel // 2), B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
        n_ce = (2 + eta) * B
        contrib = cu(rs.randn(B, k).astype(F32))
        W0, M0, V0 = rs.randn(n_rel, k).astype(F32), (rs.randn(n_rel, k) * 0.01).astype(F32), (rs.rand(n_rel, k) * 0.01).astype(F32)
        out = {}
        for mode in ("count", "bucket"):
            os.environ["EMG_GROUPING"] = mode
            try:
                we = torch.zeros(d.apply_workspace_bytes(n_ce, n_ent, k), dtype=torch.uint8, device="cuda")
                wr = torch.full((d.apply_workspace_bytes(B, n_rel, k),), 0xA5, dtype=torch.uint8, device="cuda")
                # the control region (zero before the first grouping) lies between the descriptor lists and the offsets: take it from a zero workspace's prepare
                wr_zero = torch.zeros_like(wr)
                codes = torch.empty(B * eta, dtype=torch.int32, device="cuda")
                de = torch.empty(n_ce, dtype=torch.int32, device="cuda")
                dr = torch.empty(B, dtype=torch.int32, device="cuda")
                d.prepare_batch(cu(pos), eta, [2], n_ent, codes, de, dr, n_ent, n_rel, we, wr_zero, seed=5, counter0=3)
                torch.cuda.synchronize()
                # second grouping into the SAME (now warm) workspace after scribbling over everything a grouping must rewrite:
                # the offset array included (its bytes: whatever follows the control region)
                kb = al(4 * B)
                clean_off = 6 * kb + al(12 * (B // 2 + 1)) + kb + al(12 * (B // 8 + 2))
                clean_len = al(4 * (B // 64 + 2)) + al(4 * 64) + (al(8 * ((n_rel + 1 + 4095) // 4096)) + al(4 * (n_rel + 1)))
                wr_zero[clean_off + clean_len:] = 0xA5
                d.prepare_batch(cu(pos), eta, [2], n_ent, codes, de, dr, n_ent, n_rel, we, wr_zero, seed=5, counter0=3)
            finally:
                os.environ.pop("EMG_GROUPING", None)
            W, M, V = cu(W0.copy()), cu(M0.copy()), cu(V0.copy())
            tag = torch.zeros(n_rel, dtype=torch.int32, device="cuda")
            d.apply_grouped(L.OPT_ADAM, W, k, M, V, tag, 3, contrib, B, 0, (0.01, 0.0, 0.9, 0.999, 1e-7, 0.0123), wr_zero)
            torch.cuda.synchronize()
            out[mode] = [t.cpu().numpy() for t in (W, M, V)]
        for x, y in zip(out["count"], out["bucket"]):
            np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
        untouched = np.setdiff1d(np.arange(n_rel), pos[:, 1])
        assert untouched.size > 0
        np.testing.assert_array_equal(out["bucket"][1][untouched], (np.float32(0.9) * M0[untouched]).astype(F32))   # one decay, no more
        assert not np.array_equal(out["bucket"][0][untouched], W0[untouched])


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "6"))))
def test_bucket_grouping_random_shapes_equal_counting_grouping(seed):
    """soak of the bucket grouping against the counting grouping: random batch size (1 ... 40 000), eta, corruption sides, table
    sizes from the smallest the bucket form takes (131 073 rows) to 3 M, 1 ... 5000 relations, uniform / hub-heavy / pooled
    destinations, factored or full-row contributions, now and then the global-memory form of every bucket (EMG_BUCKET_CAP) —
    codes, destinations, sorted keys, the stable order, flags, factored source rows and the descriptor lists must agree (`compare`)."""
    d = dev()
    rs = np.random.RandomState(9000 + seed)
    B = int(rs.choice([1, 7, 64, 300, 1725, 2722, 5000, 16384, 40000]))
    eta = int(rs.choice([1, 2, 5, 10, 20]))
    sides = [(2,), (0,), (1,), (0, 1)][rs.randint(0, 4)]
    n_ent = int(rs.choice([131_073, 200_000, 262_144, 1_000_000, 3_000_000]))
    n_rel = int(rs.choice([1, 3, 237, 1000, 5000]))
    if (2 + eta * len(sides)) * B > 1_200_000:
        B = 1_200_000 // (2 + eta * len(sides))
    if n_ent > 16 * (2 + eta * len(sides)) * B + (1 << 20):   # (beyond it the workspace has the radix-sort layout: neither grouping runs)
        n_ent = 1_000_000
    kind = rs.randint(0, 3)
    if kind == 0:
        s, o = rs.randint(0, n_ent, B), rs.randint(0, n_ent, B)
    elif kind == 1:      # hubs: a few entities carry most of the batch
        hubs = rs.randint(0, n_ent, 5)
        s = np.where(rs.rand(B) < 0.6, hubs[rs.randint(0, 5, B)], rs.randint(0, n_ent, B))
        o = np.where(rs.rand(B) < 0.3, hubs[rs.randint(0, 5, B)], rs.randint(0, n_ent, B))
    else:                # one neighbourhood: everything falls into one or two buckets
        base = int(rs.randint(0, n_ent - 3000))
        s, o = base + rs.randint(0, 3000, B), base + rs.randint(0, 3000, B)
    pos = np.stack([s, rs.randint(0, n_rel, B), o], 1).astype(np.int32)
    factored = bool(rs.randint(0, 2))
    kw = {}
    if rs.randint(0, 4) == 0:
        pool = (int(rs.randint(0, n_ent - 500)) + rs.permutation(500)[:int(rs.randint(2, 200))]).astype(np.int32)
        kw = dict(entities_list=cu(pool), n_choices=len(pool))
    cap = int(rs.choice([48, 300])) if rs.randint(0, 4) == 0 else None
    ref = run_prepare(d, "count", pos, eta, sides, n_ent, n_rel, factored, **dict(kw))
    got = run_prepare(d, "bucket", pos, eta, sides, n_ent, n_rel, factored, cap=cap, **dict(kw))
    compare(got, ref, B, factored, n_ent, n_rel)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "4"))))
def test_fit_on_a_large_table_random_configurations_same_bits_under_either_grouping(monkeypatch, seed):
    """soak of the training step on tables the bucket grouping takes: a random model, optimizer (Keras Adam with its dense pass inside
    the apply launch for the relation table, deferred or not; Adagrad; momentum; SGD with the in-place forms), regulariser, width,
    eta, batch split (a short last batch) and graph shape per seed — tables, optimizer state and loss must carry the counting
    grouping's bits (what `test_fit_on_a_large_table_gives_the_same_bits_under_either_grouping` checks for seven hand-picked cases)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    dev()
    rs = np.random.RandomState(11000 + seed)
    model = ("ComplEx", "DistMult", "TransE", "HolE")[rs.randint(0, 4)]
    opt = ("sgd", "adam", "adagrad", "momentum")[rs.randint(0, 4)]
    deferred = [None, False, True][rs.randint(0, 3)] if opt in ("adam",) else (True if rs.randint(0, 2) else None)
    reg = {"lambda": float(rs.choice([1e-3, 1e-2])), "p": int(rs.choice([1, 2, 3]))} if rs.randint(0, 3) == 0 else None
    if reg is None and opt != "adam":
        deferred = None
    n_ent = int(rs.choice([131_073, 140_000, 300_000]))
    n_rel, k, eta = int(rs.choice([1, 6, 300, 3000])), int(rs.choice([8, 36, 50, 100])), int(rs.choice([1, 3, 7, 20]))
    n = int(rs.choice([200, 1000, 4000]))
    cplx = model in ("ComplEx", "HolE")
    ki = 2 * k if cplx else k
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": [L.TRANSE_L1, L.TRANSE_L2][rs.randint(0, 2)], "HolE": L.HOLE}[model]
    scale = 2.0 / k if model == "HolE" else 1.0
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1).astype(np.int32)
    if rs.randint(0, 2):
        X[: n // 3, 0] = int(rs.randint(0, n_ent))     # a hub row: block tasks
    loss = ("pairwise", "nll", "self_adversarial", "multiclass_nll")[rs.randint(0, 4)]
    kw = dict(regularizer="LP", regularizer_params=reg) if reg else {}
    b0 = (n + 2) // 3
    splits = ((0, b0), (b0, b0), (2 * b0, n - 2 * b0))
    what = str((model, mid, opt, deferred, reg, n_ent, n_rel, k, eta, n, loss))

    def run(mode):
        monkeypatch.setenv("EMG_GROUPING", mode)
        tr = Trainer(mid, ki, scale, E0, R0, eta, loss=loss, optimizer=opt, optimizer_params={"lr": 0.01},
                     batches_count=3, seed=2, deferred_dense=deferred, **kw)
        tr.set_training_set(X, b0)
        for ep in (1, 2):
            for b, (s, m_) in enumerate(splits):
                nxt = splits[(b + 1) % 3]
                tr.step(s, m_, epoch=ep, batch=b + 1, prefetch=[(nxt[0], nxt[1], ep + (b == 2), (b + 1) % 3 + 1)])
        loss_v = tr.read_loss()
        Et, Rt = tr.tables_numpy()
        return Et, Rt, [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None], loss_v

    a, b = run("bucket"), run("count")
    np.testing.assert_array_equal(a[0], b[0], err_msg=what)
    np.testing.assert_array_equal(a[1], b[1], err_msg=what)
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y, err_msg=what)
    if reg:   # (the regulariser's value: FLOAT partial sums per wave of the apply, added as doubles — the partition of the rows over the
        #  waves follows the grouping's lists, so the partials differ in their last bits: 1.6e-11 relative in one seed)
        assert a[3] == pytest.approx(b[3], rel=1e-8), what
    else:
        assert a[3] == b[3], what
