"""GPU tests at BASELINE.json's FULL sizes (|E| = 1M, k = 200, eta = 20, B = 16384), through size-independent
properties — the oracle cannot run these sizes in seconds, so each test checks an invariant the domain offers:
consistency between two independent code paths, linearity of integer counters over entity ranges, a checksum of
checksums, idempotence, determinism, monotonicity of the comparison strategies.  fp32 tolerances are written at
each assert; everything integer is exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N_ENT, N_REL, K, ETA, B = 1_000_000, 1000, 200, 20, 16384
K_INT = 2 * K


def dev():
    from emgraph_amd import device
    device.require_gpu()
    return device


@pytest.fixture(scope="module")
def world():
    """trained-scale ComplEx tables of the C3/C4 shape + one batch of positives, all on the device"""
    from emgraph_amd.training import alloc_table
    dev()
    g = torch.Generator(device="cuda").manual_seed(11)
    ent = alloc_table(N_ENT, K_INT, torch.device("cuda"))
    rel = alloc_table(N_REL, K_INT, torch.device("cuda"))
    ent[:, :K_INT] = torch.randn((N_ENT, K_INT), generator=g, device="cuda") * 0.1
    rel[:, :K_INT] = torch.randn((N_REL, K_INT), generator=g, device="cuda") * 0.1
    rs = np.random.RandomState(5)
    pos = np.stack([rs.randint(0, N_ENT, B), rs.randint(0, N_REL, B), rs.randint(0, N_ENT, B)], 1).astype(np.int32)
    return ent, rel, torch.from_numpy(pos).cuda(), pos


def complex_score_f64(ent, rel, spo):
    """plain torch float64 restatement of ComplEx._fn (ComplEx.py:288-298) on gathered rows"""
    s, p, o = (ent[spo[:, 0].long(), :K_INT].double(), rel[spo[:, 1].long(), :K_INT].double(),
               ent[spo[:, 2].long(), :K_INT].double())
    sr, si, pr, pi, or_, oi = s[:, :K], s[:, K:], p[:, :K], p[:, K:], o[:, :K], o[:, K:]
    val = (pr * sr * or_ + pr * si * oi + pi * sr * oi - pi * si * or_).sum(1)
    mag = (pr.abs() * sr.abs() * or_.abs() + pr.abs() * si.abs() * oi.abs() + pi.abs() * sr.abs() * oi.abs()
           + pi.abs() * si.abs() * or_.abs()).sum(1)
    return val, mag


def test_corruptions_full_size_properties(world):
    """344k draws: range, side balance, uniformity, determinism, counter sensitivity, expand semantics"""
    from emgraph_amd import _lib as L
    d = dev()
    _, _, pos_t, pos = world
    codes = d.corrupt_codes(B, ETA, L.SIDE_SO, N_ENT, "cuda", seed=3, counter=17)
    again = d.corrupt_codes(B, ETA, L.SIDE_SO, N_ENT, "cuda", seed=3, counter=17)
    other = d.corrupt_codes(B, ETA, L.SIDE_SO, N_ENT, "cuda", seed=3, counter=18)
    c = codes.cpu().numpy()
    np.testing.assert_array_equal(c, again.cpu().numpy())                       # counter-based: a pure function
    assert (c != other.cpu().numpy()).mean() > 0.99
    repl, keep = c & 0x7FFFFFFF, (c < 0)
    assert repl.min() >= 0 and repl.max() < N_ENT
    n = c.size
    assert abs(keep.mean() - 0.5) < 4 * 0.5 / np.sqrt(n)                          # Bernoulli(1/2), 4 sigma
    hist = np.bincount(repl // (N_ENT // 100), minlength=100)[:100]
    assert np.abs(hist - n / 100).max() < 5 * np.sqrt(n / 100)                    # uniform over 100 buckets, 5 sigma
    x = d.corrupt_expand(pos_t, ETA, codes).cpu().numpy()
    tiled = np.tile(pos, (ETA, 1))                                                # eta-major: row j <-> positive j mod B
    np.testing.assert_array_equal(x[:, 1], tiled[:, 1])
    np.testing.assert_array_equal(x[keep, 0], tiled[keep, 0])                     # kept subject
    np.testing.assert_array_equal(x[~keep, 2], tiled[~keep, 2])                   # kept object
    np.testing.assert_array_equal(np.where(keep, x[:, 2], x[:, 0]), repl)
    for side, kept_col in ((L.SIDE_S, 2), (L.SIDE_O, 0)):
        cs = d.corrupt_codes(B, ETA, side, N_ENT, "cuda", seed=3, counter=17)
        xs = d.corrupt_expand(pos_t, ETA, cs).cpu().numpy()
        np.testing.assert_array_equal(xs[:, kept_col], tiled[:, kept_col])


def test_scores_full_size_two_paths_and_f64(world):
    """group-fused forward == arbitrary-triple scoring of the expanded corruptions (bitwise), and both within
    1e-4 * sum|terms| of a float64 torch restatement"""
    from emgraph_amd import _lib as L
    d = dev()
    ent, rel, pos_t, _ = world
    codes = d.corrupt_codes(B, ETA, L.SIDE_SO, N_ENT, "cuda", seed=0, counter=1)
    sp, sn = d.train_forward(L.COMPLEX, ent, rel, K_INT, 1.0, pos_t, ETA, codes)
    xneg = d.corrupt_expand(pos_t, ETA, codes)
    sn2 = d.score_triples(L.COMPLEX, ent, rel, K_INT, 1.0, xneg)
    sp2 = d.score_triples(L.COMPLEX, ent, rel, K_INT, 1.0, pos_t)
    assert torch.equal(sn, sn2) and torch.equal(sp, sp2)
    ref, mag = complex_score_f64(ent, rel, xneg)
    assert float(((sn.double() - ref).abs() / (mag + 1e-12)).max()) < 1e-4
    assert float(sn.double().sum() - ref.sum()) == pytest.approx(0.0, abs=1e-4 * float(mag.sum()) / np.sqrt(len(mag)))


def _trainer(ent, rel, **kw):
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    E = ent[:, :K_INT].cpu().numpy()
    R = rel[:, :K_INT].cpu().numpy()
    return Trainer(L.COMPLEX, K_INT, 1.0, E, R, ETA, loss="nll", optimizer="sgd", optimizer_params={"lr": 0.05},
                   batches_count=4, seed=0, **kw)


def test_training_step_full_size_invariants(world):
    """one C3 step: (1) rows outside {s, o, replacements} are bit-identical, (2) the fused + in-place + pipelined plan
    and the plain forward / loss / backward / sort-apply plan agree, (3) repeating the run is bit-identical,
    (4) checksum: sum(delta table) == -lr * sum(all gradient rows) accumulated independently in float64"""
    from emgraph_amd import _lib as L
    d = dev()
    ent, rel, pos_t, pos = world
    X = np.concatenate([pos, pos[::-1]])  # two batches resident, train on the first
    outs = []
    for kw in (dict(), dict(), dict(fused=False, inplace=False, pipeline=False)):
        tr = _trainer(ent, rel, **kw)
        tr.set_training_set(X, B)
        tr.step(0, B, epoch=1, batch=1)
        torch.cuda.synchronize()
        outs.append((tr.ent[:, :K_INT].clone(), tr.rel[:, :K_INT].clone(), tr.read_loss()))
        del tr
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    # plans agree: same gradients, different fp32 summation order inside a row's segment -> 1e-5 relative
    assert float((outs[0][0] - outs[2][0]).abs().max()) <= 1e-5 * float(outs[2][0].abs().max())
    assert outs[0][2] == pytest.approx(outs[2][2], rel=1e-6)
    E0, E1 = ent[:, :K_INT], outs[0][0]
    codes = d.corrupt_codes(B, ETA, L.SIDE_SO, N_ENT, "cuda", seed=0, counter=0)  # epoch 1, batch 1, side 0
    touched = torch.zeros(N_ENT, dtype=torch.bool, device="cuda")
    touched[pos_t[:, 0].long()] = True
    touched[pos_t[:, 2].long()] = True
    touched[(codes & 0x7FFFFFFF).long()] = True
    assert torch.equal(E1[~touched], E0[~touched])                                 # untouched rows: bit-identical
    assert float((E1[touched] - E0[touched]).abs().sum()) > 0
    # checksum of checksums: independent gradient rows (external dL/dscore path, no in-place, no apply)
    sp, sn = d.train_forward(L.COMPLEX, ent, rel, K_INT, 1.0, pos_t, ETA, codes)
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    gp = torch.empty(B, dtype=torch.float32, device="cuda")
    gn = torch.empty(B * ETA, dtype=torch.float32, device="cuda")
    d.loss(L.LOSS_NLL, sp, sn, B, ETA, 1, 1.0, 0.5, acc, gp, gn)
    ce = torch.empty(((2 + ETA) * B, K_INT), dtype=torch.float32, device="cuda")
    cr = torch.empty((B, K_INT), dtype=torch.float32, device="cuda")
    d.train_backward_ex(L.COMPLEX, ent, rel, K_INT, 1.0, pos_t, ETA, codes, ce, cr, fused_loss=-1, g_pos=gp, g_neg=gn)
    lr = 0.05
    d_ent = float((E1.double() - E0.double()).sum())
    d_rel = float((outs[0][1].double() - rel[:, :K_INT].double()).sum())
    g_ent, g_rel = float(ce.double().sum()), float(cr.double().sum())
    scale_e, scale_r = float(ce.double().abs().sum()), float(cr.double().abs().sum())
    assert abs(d_ent + lr * g_ent) < 1e-6 * lr * scale_e
    assert abs(d_rel + lr * g_rel) < 1e-6 * lr * scale_r
    assert float(acc.item()) == pytest.approx(outs[0][2], rel=1e-6)               # the loss value itself


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_factored_contributions_full_size_equal_full_rows(world, kind, monkeypatch):
    """C3 size (|E| = 1M, B = 16384, eta = 20): three consecutive steps with the negatives' gradient rows FACTORED (one
    float x a query row of the triple group) and with full gradient rows give BIT-identical tables and loss — on a
    uniform batch (70 % singletons, short segments) and on a Zipf(1.0) batch (hub rows: block tasks of the long-segment
    kernel)."""
    ent, rel, _, pos = world
    rs = np.random.RandomState(11)
    n = 3 * B
    if kind == "zipf":
        w = 1.0 / np.arange(1, N_ENT + 1)
        perm = rs.permutation(N_ENT)
        s, o = perm[rs.choice(N_ENT, n, p=w / w.sum())], perm[rs.choice(N_ENT, n, p=w / w.sum())]
    else:
        s, o = rs.randint(0, N_ENT, n), rs.randint(0, N_ENT, n)
    X = np.stack([s, rs.randint(0, int(rel.shape[0]), n), o], 1).astype(np.int32)
    outs = []
    for factored in ("1", "0"):
        monkeypatch.setenv("EMG_FACTORED", factored)
        tr = _trainer(ent, rel)
        assert tr.factored == (factored == "1")
        tr.set_training_set(X, B)
        for b in range(3):
            tr.step(b * B, B, epoch=1, batch=b + 1, prefetch=[((b + 1) * B, B, 1, b + 2)] if b < 2 else None)
        torch.cuda.synchronize()
        outs.append((tr.ent[:, :K_INT].clone(), tr.rel[:, :K_INT].clone(), tr.read_loss()))
        del tr
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    assert not torch.equal(outs[0][0], ent[:, :K_INT])


@pytest.mark.parametrize("opt,reg,kind", [("adam", None, "uniform"), ("adam", None, "zipf"), ("sgd", 2, "uniform"), ("adagrad", 3, "zipf")])
def test_deferred_dense_pass_full_size_equals_dense_pass(world, opt, reg, kind):
    """C3 size: Keras Adam's decay / the LP regulariser's gradient reach all 1M x 400 values every step.  The default for a table of
    this size leaves a row alone until a batch is about to read it and replays its missed steps then (emg_deferred_catchup;
    Adam: w alone, the apply redoes m, v) — five steps of it and of the literal dense pass must leave the same tables, the same
    optimizer state and the same loss, bit for bit, on a uniform batch and on a Zipf(1.0) batch (hub rows: block tasks)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    ent, rel, _, _ = world
    rs = np.random.RandomState(17)
    n = 5 * B
    if kind == "zipf":
        w = 1.0 / np.arange(1, N_ENT + 1)
        perm = rs.permutation(N_ENT)
        s, o = perm[rs.choice(N_ENT, n, p=w / w.sum())], perm[rs.choice(N_ENT, n, p=w / w.sum())]
    else:
        s, o = rs.randint(0, N_ENT, n), rs.randint(0, N_ENT, n)
    X = np.stack([s, rs.randint(0, N_REL, n), o], 1).astype(np.int32)
    E, R = ent[:, :K_INT].cpu().numpy(), rel[:, :K_INT].cpu().numpy()
    kw = dict(regularizer="LP", regularizer_params={"lambda": 1e-5, "p": reg}) if reg else {}
    outs = []
    for deferred in (None, False):   # (None: the default, which must be the deferred form at this size)
        tr = Trainer(L.COMPLEX, K_INT, 1.0, E, R, ETA, loss="nll", optimizer=opt, optimizer_params={"lr": 0.01}, batches_count=5,
                     seed=0, deferred_dense=deferred, **kw)
        assert tr.deferred == (deferred is None)
        tr.set_training_set(X, B)
        for b in range(5):
            tr.step(b * B, B, epoch=1, batch=b + 1, prefetch=[((b + 1) * B, B, 1, b + 2)] if b < 4 else None)
        tr.materialize()
        torch.cuda.synchronize()
        state = [t[:, :K_INT].clone() for t in tr.state_ent + tr.state_rel if t is not None]
        outs.append((tr.ent[:, :K_INT].clone(), tr.rel[:, :K_INT].clone(), state, tr.read_loss()))
        del tr
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert len(a[2]) == len(b[2]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    if reg:   # (the penalty's value: double atomics over float partial sums, grouped by row here and by step there)
        np.testing.assert_allclose(a[3], b[3], rtol=1e-7)
    else:
        assert a[3] == b[3]
    assert not torch.equal(a[0], ent[:, :K_INT])


def test_zero_learning_rate_step_is_idempotent(world):
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    ent, rel, _, pos = world
    tr = Trainer(L.COMPLEX, K_INT, 1.0, ent[:, :K_INT].cpu().numpy(), rel[:, :K_INT].cpu().numpy(), ETA, loss="nll",
                 optimizer="sgd", optimizer_params={"lr": 0.0}, batches_count=1, seed=0)
    tr.set_training_set(pos, B)
    tr.step(0, B, epoch=1, batch=1)
    torch.cuda.synchronize()
    assert torch.equal(tr.ent[:, :K_INT], ent[:, :K_INT]) and torch.equal(tr.rel[:, :K_INT], rel[:, :K_INT])


def test_ranking_full_size_invariants(world):
    """1-vs-all over 1M entities: counters are linear over entity ranges, agree with an independent torch
    reduction of the kernel's own dense scores, dense scores agree with float64, filtered <= raw,
    best <= middle <= worst; the bf16 mode keeps the self tie and tracks the exact ranks"""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, rank_triples_device
    d = dev()
    ent, rel, pos_t, pos = world
    T = pos[:96]
    Tt = pos_t[:96]
    Q, pos_int = d.eval_build_queries(L.COMPLEX, ent, rel, K_INT, 1.0, Tt, L.EVAL_SPO)
    n_rows = Q.shape[0]
    full = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
    d.eval_count(L.COMPLEX, Q, pos_int, ent, K_INT, 1.0, full[0], full[1])
    parts = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
    for a, b in ((0, 333_333), (333_333, 333_400), (333_400, N_ENT)):           # ragged ranges, counters accumulate
        d.eval_count(L.COMPLEX, Q, pos_int, ent[a:b], K_INT, 1.0, parts[0], parts[1])
    assert torch.equal(full, parts)
    assert int(full[1].min()) >= 1                                                # every positive ties with itself
    # independent reduction of the kernel's dense scores for a slice of the rows
    S = d.eval_scores_dense(L.COMPLEX, Q[:32], ent, K_INT, 1.0)                  # [32, 1M]
    ci = (S * 100000.0).to(torch.int32)                                           # truncation toward zero, as the reference
    p = pos_int[:32, None]
    assert torch.equal((ci > p).sum(1).to(torch.int32), full[0, :32])
    assert torch.equal((ci == p).sum(1).to(torch.int32), full[1, :32])
    ref = Q[:32, :K_INT].double() @ ent[:, :K_INT].double().T
    mag = Q[:32, :K_INT].double().abs() @ ent[:, :K_INT].double().abs().T
    assert float(((S.double() - ref).abs() / (mag + 1e-12)).max()) < 1e-4
    del S, ci, ref, mag
    # public ranking: filtered <= raw, strategies ordered, 's,o' columns are the 's' and 'o' runs
    F = FilterIndex(np.concatenate([pos, pos[:, [2, 1, 0]]]))
    raw = rank_triples_device(L.COMPLEX, ent, rel, K_INT, 1.0, T, "s,o", "worst")
    flt = {s: rank_triples_device(L.COMPLEX, ent, rel, K_INT, 1.0, T, "s,o", s, filter_triples=F)
           for s in ("worst", "middle", "best")}
    assert np.all(flt["worst"] <= raw) and raw.min() >= 1 and raw.max() <= N_ENT + 1
    assert np.all(flt["best"] <= flt["middle"]) and np.all(flt["middle"] <= flt["worst"])
    np.testing.assert_array_equal(flt["worst"][:, 0], rank_triples_device(L.COMPLEX, ent, rel, K_INT, 1.0, T, "s", "worst",
                                                                          filter_triples=F))
    np.testing.assert_array_equal(flt["worst"][:, 1], rank_triples_device(L.COMPLEX, ent, rel, K_INT, 1.0, T, "o", "worst",
                                                                          filter_triples=F))
    # query-chunking must not matter
    np.testing.assert_array_equal(raw, rank_triples_device(L.COMPLEX, ent, rel, K_INT, 1.0, T, "s,o", "worst", query_chunk=40))
    # bf16 throughput mode (register-stationary kernel: 192 rows > 128): statistical agreement, exact self tie
    fast = rank_triples_device(L.COMPLEX, ent, rel, K_INT, 1.0, T, "s,o", "worst", filter_triples=F, precision=1)
    rel_err = np.abs(fast - flt["worst"]) / N_ENT
    assert np.median(rel_err) < 2e-3 and rel_err.max() < 2e-2, (np.median(rel_err), rel_err.max())
    assert fast.min() >= 1


@pytest.mark.parametrize("rescore", ["segments", "tiles"])
def test_exact_fast_ranking_full_size_equals_exact(world, monkeypatch, rescore):
    """(both forms of the exact re-scoring: segment-wise with the query rows in LDS — the default —, and entity-tile-major with
    the pairs bucketed by tile of 32 entity rows, EMG_RESCORE=tiles)
    precision 2 (half-precision MFMA prefilter with a rigorous error band + exact re-scoring of the undecided
    candidates) at |E| = 1M: ranks BIT-equal to the exact f32 path for all three strategies, filtered and raw, on
    the Glorot-scale tables of the training world (scores ~1e-4: most comparison integers tie at 0, the hardest case
    for a band-based prefilter — hundreds of thousands of undecided pairs) and on trained-scale tables."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, PrefilterTables, rank_triples_device
    from emgraph_amd.training import alloc_table
    ent, rel, pos_t, pos = world
    monkeypatch.setenv("EMG_RESCORE", rescore)
    T = pos[:512]
    F = FilterIndex(pos[:200000])
    rs = np.random.RandomState(3)
    dev_ = torch.device("cuda")
    ent_t = alloc_table(N_ENT, K_INT, dev_, init=(rs.randn(N_ENT, K_INT) * 0.1).astype(np.float32))
    rel_t = alloc_table(rel.shape[0], K_INT, dev_, init=(rs.randn(rel.shape[0], K_INT) * 0.1).astype(np.float32))
    for e_, r_, label in ((ent_t, rel_t, "trained-scale"), (ent, rel, "glorot-scale")):
        tabs = PrefilterTables(e_, K_INT)
        for strategy in ("worst", "best", "middle"):
            for filt in (F, None):
                st = {}
                exact = rank_triples_device(L.COMPLEX, e_, r_, K_INT, 1.0, T, "s,o", strategy, filter_triples=filt)
                fast = rank_triples_device(L.COMPLEX, e_, r_, K_INT, 1.0, T, "s,o", strategy, filter_triples=filt, precision=2,
                                           ent_f16=tabs, stats=st)
                np.testing.assert_array_equal(fast, exact, err_msg=str((label, strategy, filt is not None, st)))
                assert st.get("pairs", 0) > 0 or st.get("fallback", 0) > 0


def test_transe_l1_exact_fast_ranking_full_size_equals_exact(world):
    """TransE-L1 at |E| = 1M, k = 200: precision 2 (v_sad_u16 sums over 16-bit fixed-point images, undecided candidates
    re-scored with the canonical f32 chain) gives ranks BIT-equal to the exact f32 kernel, filtered and raw, for all
    three strategies, on trained-scale tables with planted copies / last-bit neighbours of the true entities; and
    precision 'auto' (what evaluate_performance uses) takes this path and agrees."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, SadTables, rank_triples_device
    from emgraph_amd.training import alloc_table
    _, rel, _, pos = world
    T = pos[:384]
    F = FilterIndex(pos[:200000])
    rs = np.random.RandomState(11)
    k = 200
    E = (rs.randn(N_ENT, k) * 0.1).astype(np.float32)
    for j in range(0, len(T), 3):
        E[rs.randint(0, N_ENT, 2)] = E[T[j, 2]]
        E[rs.randint(0, N_ENT)] = np.nextafter(E[T[j, 0]], np.float32(np.inf))
    dev_ = torch.device("cuda")
    ent_t = alloc_table(N_ENT, k, dev_, init=E)
    rel_t = alloc_table(rel.shape[0], k, dev_, init=(rs.randn(rel.shape[0], k) * 0.1).astype(np.float32))
    tabs = SadTables(ent_t, rel_t, k)
    for strategy in ("worst", "best", "middle"):
        for filt in (F, None):
            st = {}
            exact = rank_triples_device(L.TRANSE_L1, ent_t, rel_t, k, 1.0, T, "s,o", strategy, filter_triples=filt)
            fast = rank_triples_device(L.TRANSE_L1, ent_t, rel_t, k, 1.0, T, "s,o", strategy, filter_triples=filt, precision=2,
                                       ent_f16=tabs, stats=st)
            np.testing.assert_array_equal(fast, exact, err_msg=str((strategy, filt is not None, st)))
            assert st.get("pairs", 0) > 0 and st.get("fallback", 0) == 0
    st = {}
    auto = rank_triples_device(L.TRANSE_L1, ent_t, rel_t, k, 1.0, T, "s,o", "worst", filter_triples=F, precision="auto", stats=st)
    np.testing.assert_array_equal(auto, rank_triples_device(L.TRANSE_L1, ent_t, rel_t, k, 1.0, T, "s,o", "worst", filter_triples=F))
    assert st.get("pairs", 0) > 0


def test_transe_l2_exact_fast_ranking_full_size_equals_exact(world):
    """TransE-L2 at |E| = 1M, k = 200 through the MFMA prefilter on the augmented rows: ranks BIT-equal to the exact f32
    kernel, filtered and raw, all three strategies; precision 'auto' takes this path (k + 2 = 202 is a covered width)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, L2Tables, rank_triples_device
    from emgraph_amd.training import alloc_table
    _, rel, _, pos = world
    T = pos[:384]
    F = FilterIndex(pos[:200000])
    rs = np.random.RandomState(12)
    k = 200
    E = (rs.randn(N_ENT, k) * 0.1).astype(np.float32)
    for j in range(0, len(T), 3):
        E[rs.randint(0, N_ENT, 2)] = E[T[j, 2]]
        E[rs.randint(0, N_ENT)] = np.nextafter(E[T[j, 0]], np.float32(np.inf))
    dev_ = torch.device("cuda")
    ent_t = alloc_table(N_ENT, k, dev_, init=E)
    rel_t = alloc_table(rel.shape[0], k, dev_, init=(rs.randn(rel.shape[0], k) * 0.1).astype(np.float32))
    tabs = L2Tables(ent_t, k)
    for strategy in ("worst", "best", "middle"):
        for filt in (F, None):
            st = {}
            exact = rank_triples_device(L.TRANSE_L2, ent_t, rel_t, k, 1.0, T, "s,o", strategy, filter_triples=filt)
            fast = rank_triples_device(L.TRANSE_L2, ent_t, rel_t, k, 1.0, T, "s,o", strategy, filter_triples=filt, precision=2,
                                       ent_f16=tabs, stats=st)
            np.testing.assert_array_equal(fast, exact, err_msg=str((strategy, filt is not None, st)))
            assert st.get("pairs", 0) > 0 and st.get("fallback", 0) == 0
    st = {}
    auto = rank_triples_device(L.TRANSE_L2, ent_t, rel_t, k, 1.0, T, "s,o", "worst", filter_triples=F, precision="auto", stats=st)
    np.testing.assert_array_equal(auto, rank_triples_device(L.TRANSE_L2, ent_t, rel_t, k, 1.0, T, "s,o", "worst", filter_triples=F))
    assert st.get("pairs", 0) > 0


# ------------------------------------------------------------------------------------------------
# the ORACLE at |E| = 1M: the C restatement's canonical chain (oracle/emg_oracle.c: orc_count / orc_filter_count, which
# restate EmbeddingModel.py:1845-2033) for 16 test triples = 32 query rows against all 1M entities, compared bit for bit
# with the device path in its exact (precision 0) and exact-fast (precision 2) forms, sides 's+o' and 's,o', filtered
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,k_int", [("ComplEx", K_INT), ("TransE_L1", K), ("TransE_L2", K)])
def test_ranks_equal_c_oracle_at_one_million_entities(world, model, k_int):
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, build_filter_csr, rank_triples_device, ranks_from_counts
    from oracle import c_oracle as co
    ent, rel, _, pos = world
    mid = {"ComplEx": L.COMPLEX, "TransE_L1": L.TRANSE_L1, "TransE_L2": L.TRANSE_L2}[model]
    # TransE: the first k columns of the same tables (16-byte aligned rows, stride K_INT)
    ent_m, rel_m = ent[:, :k_int], rel[:, :k_int]
    E, R = ent_m.cpu().numpy(), rel_m.cpu().numpy()
    n_oracle, n_dev = 16, 160              # the device ranks 160 triples (the exact-fast path wants >= 128); the oracle the first 16
    T = pos[:n_dev].copy()
    T[1] = T[0]                            # duplicates and a shared (s, p): non-trivial filter lists
    T[3, :2] = T[2, :2]
    Fil = np.concatenate([pos[:4096], T])
    F = FilterIndex(Fil)
    for side, mode in (("s+o", L.EVAL_SPO), ("s,o", L.EVAL_S_O)):
        Q, pos_int = co.build_queries(mid, E, R, k_int, 1.0, T[:n_oracle], mode)
        gt, eq = co.count(mid, Q, pos_int, E, k_int, 1.0)
        ptr, idx = build_filter_csr(Fil, T[:n_oracle], mode, N_ENT)
        fgt, feq = co.filter_count(mid, Q, pos_int, E, 0, k_int, 1.0, ptr, idx)
        for strategy in ("worst", "middle"):
            want = ranks_from_counts(gt, eq, fgt, feq, n_oracle, side, strategy)
            for precision in (0, 2):
                got = rank_triples_device(mid, ent_m, rel_m, k_int, 1.0, T, side, strategy, filter_triples=F, precision=precision)
                np.testing.assert_array_equal(got[:n_oracle], want, err_msg="%s %s %s precision %d" % (model, side, strategy, precision))


def test_scores_relative_error_on_well_conditioned_triples(world):
    """north_star's bar as worded — fp32 scores within 1e-4 RELATIVE — on the triples where |score| is a meaningful
    denominator (|score| >= 0.1 * sum |terms|: no catastrophic cancellation); printed for the record.  (The general bar,
    1e-4 * sum |terms|, is what test_scores_full_size_two_paths_and_f64 and tests/test_hip_kernels.py::score_tol assert.)"""
    from emgraph_amd import _lib as L
    d = dev()
    ent, rel, _, pos = world
    # planted triples: the object's row pulled towards the query so that the score is large relative to its terms
    Tt = torch.from_numpy(pos[:4096]).cuda()
    Q, _ = d.eval_build_queries(L.COMPLEX, ent, rel, K_INT, 1.0, Tt, L.EVAL_O)
    o = Tt[:, 2].long()
    ent2 = ent.clone()
    qh = Q[:, :K_INT] / Q[:, :K_INT].norm(dim=1, keepdim=True)
    ent2[o, :K_INT] = 0.6 * ent2[o, :K_INT] + 0.8 * ent2[o, :K_INT].norm(dim=1, keepdim=True) * qh
    got = d.score_triples(L.COMPLEX, ent2, rel, K_INT, 1.0, Tt).double()
    val, mag = complex_score_f64(ent2, rel, Tt)
    well = val.abs() >= 0.1 * mag
    assert int(well.sum()) >= 1000, int(well.sum())
    rel_err = ((got - val).abs() / val.abs())[well]
    print("ComplEx k=200, |E|=1M: %d well-conditioned triples, max relative score error %.3g, median %.3g"
          % (int(well.sum()), float(rel_err.max()), float(rel_err.median())))
    assert float(rel_err.max()) <= 1e-4


# ------------------------------------------------------------------------------------------------
# the ORACLE at C3's EXACT training shape (|E| = 1M, |R| = 1k, k = 200, eta = 20, B = 16384, NLL): two consecutive steps of the
# step fit() runs — plain SGD: fused kernel, in-place singletons, factored contributions; Adam (the reference's default,
# constants.py:55): deferred dense pass, singleton negatives replayed and updated inside the fused kernel, the stateful apply for
# the rest — against orc.train_grads_sparse (the per-triple gradient rows of
# EmbeddingModel.py:614-822's loss, float64, grouped by destination) and orc.opt_apply (training/sgd.py, adam.py) on the rows
# the two batches touch; every other row must keep its bits.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_training_steps_vs_oracle_at_c3_exact_shape(world, opt):
    from emgraph_amd import _lib as L
    from emgraph_amd.training import ADAM_BETA1, ADAM_BETA2, Trainer
    from oracle import emgraph_oracle as orc
    ent, rel, _, pos = world
    E0, R0 = ent[:, :K_INT].cpu().numpy(), rel[:, :K_INT].cpu().numpy()
    rs = np.random.RandomState(21)
    second = np.stack([rs.randint(0, N_ENT, B), rs.randint(0, N_REL, B), rs.randint(0, N_ENT, B)], 1).astype(np.int32)
    X = np.concatenate([pos, second])
    lr = 0.05 if opt == "sgd" else 0.001
    tr = Trainer(L.COMPLEX, K_INT, 1.0, E0, R0, ETA, loss="nll", optimizer=opt, optimizer_params={"lr": lr}, batches_count=2, seed=0)
    tr.set_training_set(X, B)
    assert tr.fused and tr.factored and tr.inplace and tr.deferred == (opt == "adam")   # (Adam: singleton negatives replayed + updated in the scoring kernel)
    xnegs = [orc.generate_corruptions_for_fit_philox(X[b * B:(b + 1) * B], eta=ETA, corrupt_side="s,o", entities_size=N_ENT, seed=0, counter=b)
             for b in (0, 1)]
    rows_all = np.unique(np.concatenate([X[:, 0], X[:, 2]] + [x[:, 0] for x in xnegs] + [x[:, 2] for x in xnegs]))   # every row either batch touches
    rels_all = np.unique(X[:, 1])
    Ep, Rp = E0, R0
    st = {"E": orc.opt_init(opt, (len(rows_all), K_INT)), "R": orc.opt_init(opt, (len(rels_all), K_INT))}
    for b in (0, 1):
        tr.step(b * B, B, epoch=1, batch=b + 1, prefetch=[(B, B, 1, 2)] if b == 0 else None)
        loss = tr.read_loss()
        E1, R1 = tr.tables_numpy()                      # (Adam: materialize() brings every row up to this step)
        xb = X[b * B:(b + 1) * B]
        ue, ge, ur, gr, oloss = orc.train_grads_sparse("ComplEx", Ep, Rp, xb, ETA, "nll", None, [xnegs[b]])
        assert loss == pytest.approx(oloss, rel=2e-5), (b, loss, oloss)
        ulp = 2.0 ** -23
        for tab, (ids_all, ids, g, W0, W1, key) in (("ent", (rows_all, ue, ge, Ep, E1, "E")), ("rel", (rels_all, ur, gr, Rp, R1, "R"))):
            at = np.searchsorted(ids_all, ids)
            gmax = np.abs(g).max()
            if opt == "sgd":
                # w1 = w0 - lr g: the gradient within rtol 1e-4 (+ 1e-5 of the largest) through an update stored in float32
                want = W0[ids].astype(np.float64) - lr * g
                tol = lr * (1e-4 * np.abs(g) + 1e-5 * gmax) + ulp * np.abs(W0[ids])
                assert np.all(np.abs(W1[ids] - want) <= tol), (tab, b, float(np.max(np.abs(W1[ids] - want) / tol)))
                untouched = np.ones(W0.shape[0], bool)
                untouched[ids] = False
                np.testing.assert_array_equal(W1[untouched], W0[untouched])
            else:
                # the oracle's dense-equivalent Adam on the rows either batch touches (all other rows: m = v = 0, they cannot move)
                g_all = np.zeros((len(ids_all), K_INT), dtype=np.float32)
                g_all[at] = g
                m0, v0 = st[key]["m"].copy(), st[key]["v"].copy()
                want = orc.opt_apply("adam", W0[ids_all], g_all, st[key], lr=lr)
                got_m, got_v = [t.cpu().numpy()[ids_all][:, :K_INT] for t in (tr.state_ent if tab == "ent" else tr.state_rel)]
                # gradients read back through the state: m = b1 m0 + (1 - b1) g, v = b2 v0 + (1 - b2) g^2
                g_from_m = (got_m.astype(np.float64) - ADAM_BETA1 * m0) / (1 - ADAM_BETA1)
                assert np.all(np.abs(g_from_m - g_all) <= 1e-4 * np.abs(g_all) + 1e-5 * gmax + 8 * ulp * np.abs(m0) / (1 - ADAM_BETA1)), (tab, b)
                np.testing.assert_allclose(got_v, st[key]["v"], rtol=3e-4, atol=1e-5 * (1 - ADAM_BETA2) * gmax * gmax)
                # the step lr_t m / (sqrt(v) + eps) is bounded by ~3.2 lr whatever g; where |g| is far above eps it is a smooth function
                # of g: agree within 1e-3 of the step there, and never differ by more than a step
                err = np.abs(W1[ids_all] - want)
                step = np.abs(want.astype(np.float64) - W0[ids_all])
                smooth = np.abs(g_all) > 1e-3 * gmax
                # (+ what a 1e-3 relative difference of g moves the step by: after a cancellation in m the step itself is tiny)
                t_now = b + 1
                lr_t = lr * np.sqrt(1 - ADAM_BETA2 ** t_now) / (1 - ADAM_BETA1 ** t_now)
                sens = lr_t * (1 - ADAM_BETA1) * 1e-3 * np.abs(g_all) / (np.sqrt(st[key]["v"]) + 1e-7)
                tol = 2e-3 * step + sens + 2 * ulp * np.abs(want)
                assert np.all(err[smooth] <= tol[smooth]), (tab, b, float((err[smooth] / tol[smooth]).max()))
                assert err.max() <= 4 * lr
                st[key]["m"], st[key]["v"] = got_m.copy(), got_v.copy()     # step 2 starts from the device's state (no error build-up)
                never = np.ones(W0.shape[0], bool)
                never[ids_all] = False
                np.testing.assert_array_equal(W1[never], E0[never] if tab == "ent" else R0[never])
        Ep, Rp = E1, R1


@pytest.mark.parametrize("form", ["adagrad", "momentum", "sgd_lp2_deferred", "sgd_b131072"])
def test_training_steps_vs_oracle_at_c3_shape_stateful_lp_and_large_batch(world, form):
    """the seams round 4's review named: the WINDOW forms of Adagrad / momentum (ip 4: state rows of singleton negatives
    travelling with their table rows in the scoring kernel, adagrad.py:30-46 / momentum.py:51-69), SGD + the LP regulariser with
    its dense pass DEFERRED (C3r: lp.py:107-113 folded into the step, singletons in place, emg_deferred_catchup for the rest) and
    one step at B = 131 072 (SURVEY 8d's second batch size; the bucket grouping's wide-chunk form) — each against
    orc.train_grads_sparse (float64 gradient rows of EmbeddingModel.py:614-822's loss, grouped by destination) and the optimizer
    rules restated from the reference, at |E| = 1M, k = 200, eta = 20.  Loss rtol 2e-5, gradients rtol 1e-4 (+ 1e-5 of the largest)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import ADAGRAD_INIT_ACC, DEFAULT_MOMENTUM, Trainer
    from oracle import emgraph_oracle as orc
    ent, rel, _, pos = world
    E0, R0 = ent[:, :K_INT].cpu().numpy(), rel[:, :K_INT].cpu().numpy()
    rs = np.random.RandomState(33)
    Bt = 131072 if form == "sgd_b131072" else B
    n_steps = 1 if form == "sgd_b131072" else 2
    X = np.stack([rs.randint(0, N_ENT, n_steps * Bt), rs.randint(0, N_REL, n_steps * Bt), rs.randint(0, N_ENT, n_steps * Bt)], 1).astype(np.int32)
    opt = form if form in ("adagrad", "momentum") else "sgd"
    lam, lp_p = 1e-5, 2
    lr = {"adagrad": 0.05, "momentum": 0.02}.get(form, 0.05)
    kw = dict(regularizer="LP", regularizer_params={"lambda": lam, "p": lp_p}) if form == "sgd_lp2_deferred" else {}
    tr = Trainer(L.COMPLEX, K_INT, 1.0, E0, R0, ETA, loss="nll", optimizer=opt, optimizer_params={"lr": lr}, batches_count=n_steps, seed=0, **kw)
    tr.set_training_set(X, Bt)
    assert tr.fused and tr.inplace, (tr.fused, tr.inplace)
    if form in ("adagrad", "momentum"):
        assert tr.inplace_mode == 2            # the window form (ip 4)
    if form == "sgd_lp2_deferred":
        assert tr.deferred                     # the dense regulariser pass is replayed on demand (C3r)
    ulp = 2.0 ** -23
    Ep, Rp = E0, R0
    state_prev = {"ent": None, "rel": None}
    for b in range(n_steps):
        xb = X[b * Bt:(b + 1) * Bt]
        xneg = orc.generate_corruptions_for_fit_philox(xb, eta=ETA, corrupt_side="s,o", entities_size=N_ENT, seed=0, counter=b)
        if opt != "sgd":
            for tab, st in (("ent", tr.state_ent), ("rel", tr.state_rel)):
                state_prev[tab] = st[0].cpu().numpy()[:, :K_INT].copy()
        tr.step(b * Bt, Bt, epoch=1, batch=b + 1, prefetch=[(Bt, Bt, 1, 2)] if (b == 0 and n_steps > 1) else None)
        loss = tr.read_loss()
        E1, R1 = tr.tables_numpy()
        ue, ge, ur, gr, oloss = orc.train_grads_sparse("ComplEx", Ep, Rp, xb, ETA, "nll", None, [xneg])
        if form == "sgd_lp2_deferred":       # + lambda * sum |w|^p over the FULL tables as they were before the step (lp.py:107-113)
            oloss += lam * (float(np.sum(np.abs(Ep.astype(np.float64)) ** lp_p)) + float(np.sum(np.abs(Rp.astype(np.float64)) ** lp_p)))
        assert loss == pytest.approx(oloss, rel=2e-5), (b, loss, oloss)
        for tab, ids, g, W0, W1, state in (("ent", ue, ge, Ep, E1, tr.state_ent), ("rel", ur, gr, Rp, R1, tr.state_rel)):
            gmax = np.abs(g).max()
            gtol = 1e-4 * np.abs(g) + 1e-5 * gmax
            untouched = np.ones(W0.shape[0], bool)
            untouched[ids] = False
            w0 = W0[ids].astype(np.float64)
            if form == "sgd_lp2_deferred":
                # every row moves by the regulariser's gradient lambda p |w|^(p-1) sgn(w); touched rows by the loss's as well
                reg_g = lam * lp_p * np.abs(w0) ** (lp_p - 1) * np.sign(w0)
                want = w0 - lr * (g + reg_g)
                assert np.all(np.abs(W1[ids] - want) <= lr * gtol + 2 * ulp * np.abs(w0)), (tab, b)
                wu = W0[untouched].astype(np.float64)
                want_u = wu - lr * lam * lp_p * np.abs(wu) ** (lp_p - 1) * np.sign(wu)
                assert np.all(np.abs(W1[untouched] - want_u) <= 2 * ulp * np.abs(wu) + 1e-12), (tab, b)
                assert not untouched.any() or not np.array_equal(W1[untouched], W0[untouched])
                continue
            np.testing.assert_array_equal(W1[untouched], W0[untouched])
            if opt == "sgd":
                want = w0 - lr * g
                assert np.all(np.abs(W1[ids] - want) <= lr * gtol + ulp * np.abs(w0)), (tab, b, float(np.max(np.abs(W1[ids] - want))))
                continue
            s0 = state_prev[tab][ids].astype(np.float64)
            s1 = state[0].cpu().numpy()[:, :K_INT]
            np.testing.assert_array_equal(s1[untouched], state_prev[tab][untouched])      # row-sparse state: untouched rows keep theirs
            s1 = s1[ids].astype(np.float64)
            if opt == "momentum":    # Keras SGD(momentum): v = mu v - lr g ; w += v   (momentum.py:51-69)
                want_s = DEFAULT_MOMENTUM * s0 - lr * g
                assert np.all(np.abs(s1 - want_s) <= lr * gtol + 2 * ulp * np.abs(want_s) + 2 * ulp * np.abs(s0)), (tab, b)
                assert np.all(np.abs(W1[ids] - (w0 + s1)) <= ulp * (np.abs(w0) + np.abs(s1))), (tab, b)
            else:                    # Adagrad: acc += g^2 ; w -= lr g / (sqrt(acc) + eps)   (adagrad.py:30-46)
                if b == 0:
                    assert np.all(s0 == np.float32(ADAGRAD_INIT_ACC))
                want_s = s0 + g * g
                assert np.all(np.abs(s1 - want_s) <= 2 * np.abs(g) * gtol + 2 * ulp * want_s), (tab, b)
                step = lr * g / (np.sqrt(s1) + 1e-7)           # from the device's accumulator: the step is a smooth function of g
                tol = lr * gtol / (np.sqrt(s1) + 1e-7) + 4 * ulp * np.abs(step) + ulp * np.abs(w0)
                assert np.all(np.abs(W1[ids] - (w0 - step)) <= tol), (tab, b, float(np.max(np.abs(W1[ids] - (w0 - step)) / tol)))
        Ep, Rp = E1, R1


def test_ranks_assembled_by_the_literal_oracle_at_one_million_entities(world):
    """the rank ASSEMBLY of the reference (EmbeddingModel.py:1856-1986: eval corruptions, filter lookups, perform_comparision, rank =
    cmp(all) + 1 - cmp(filter_s) - cmp(filter_o)) done by oracle.emgraph_oracle.rank_triple — its own generate_corruptions_for_eval,
    participating_entities and comparison, nothing of the product's ranks_from_counts / build_filter_csr — for 4 test triples against
    all 1M entities, every side and strategy: equal to the device ranks of the exact and the exact-fast path.  The scores it
    compares are the canonical chain's (C oracle), handed over as the comparison integers relative to the side's own positive
    (the canonical positive is scored per side, by the same chain as its candidates: DESIGN 3): d * 2^-10 keeps order and ties
    through perform_comparision's int32(score * 1e5)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, rank_triples_device
    from oracle import c_oracle as co
    from oracle import emgraph_oracle as orc
    ent, rel, _, pos = world
    E, R = ent[:, :K_INT].cpu().numpy(), rel[:, :K_INT].cpu().numpy()
    n_dev, n_or = 160, 4
    T = pos[:n_dev].copy()
    T[1] = T[0]
    T[3, :2] = T[2, :2]
    Fil = np.concatenate([pos[:4096], T])
    F = FilterIndex(Fil)
    want = {}
    for i in range(n_or):
        s_, p_, o_ = (int(v) for v in T[i])
        Q, _ = co.build_queries(L.COMPLEX, E, R, K_INT, 1.0, T[i:i + 1], L.EVAL_S_O)      # row 0: object side, row 1: subject side
        dense = co.scores_dense(L.COMPLEX, Q, E, K_INT, 1.0)
        ci = [orc.to_cmp_int(dense[0]).astype(np.int64), orc.to_cmp_int(dense[1]).astype(np.int64)]
        rel_o, rel_s = ci[0] - ci[0][o_], ci[1] - ci[1][s_]

        def scorer(rows, s_=s_, rel_o=rel_o, rel_s=rel_s):
            rows = np.asarray(rows)
            obj_block = rows[:, 0] == s_                      # (s, p, c); the positive itself is candidate o of this block
            d = np.where(obj_block, rel_o[rows[:, 2]], rel_s[rows[:, 0]])
            return (d * 2.0 ** -10).astype(np.float32)

        chk = np.array([-3, -1, 0, 1, 2, 200000], dtype=np.int64)
        assert np.all(np.diff(orc.to_cmp_int((chk * 2.0 ** -10).astype(np.float32))) > 0) and orc.to_cmp_int(np.float32(0)) == 0
        for side in ("s,o", "s+o", "s", "o"):
            for strategy in ("worst", "middle", "best"):
                want[(i, side, strategy)] = orc.rank_triple("ComplEx", E, R, T[i], corrupt_side=side, strategy=strategy,
                                                            filter_triples=Fil, score_override=scorer)
    for strategy in ("worst", "middle", "best"):
        for precision in (0, 2):
            for side in ("s,o", "s+o", "s", "o"):
                got = rank_triples_device(L.COMPLEX, ent[:, :K_INT], rel[:, :K_INT], K_INT, 1.0, T, side, strategy, filter_triples=F, precision=precision)
                for i in range(n_or):
                    np.testing.assert_array_equal(got[i], want[(i, side, strategy)],
                                                  err_msg="%s %s precision %d triple %d" % (side, strategy, precision, i))


@pytest.mark.parametrize("model", ["TransE_L1", "TransE_L2", "DistMult", "HolE"])
def test_scores_relative_error_every_model(world, model):
    """north_star's bar as worded — fp32 scores within 1e-4 relative to |score| — for the other four score functions on the 1M-entity
    table (ComplEx: test_scores_relative_error_on_well_conditioned_triples): against a float64 restatement of TransE.py:208-216,
    DistMult.py:201, HolE.py:189, on the triples whose |score| is at least a tenth of the sum of |terms| (all of them for TransE —
    a norm has no cancellation —, planted objects for the bilinear models)."""
    from emgraph_amd import _lib as L
    d = dev()
    ent, rel, _, pos = world
    mid = getattr(L, model.upper())
    k_int = K_INT if model == "HolE" else K
    scale = float(np.float32(2 / K)) if model == "HolE" else 1.0
    Tt = torch.from_numpy(pos[:4096]).cuda()
    ent2 = ent.clone()
    if not model.startswith("TransE"):
        Q, _ = d.eval_build_queries(mid, ent, rel, k_int, scale, Tt, L.EVAL_O)
        o = Tt[:, 2].long()
        qh = Q[:, :k_int] / Q[:, :k_int].norm(dim=1, keepdim=True)
        ent2[o, :k_int] = 0.6 * ent2[o, :k_int] + 0.8 * ent2[o, :k_int].norm(dim=1, keepdim=True) * qh
    e2, r2 = ent2[:, :k_int], rel[:, :k_int]
    got = d.score_triples(mid, e2, r2, k_int, scale, Tt).double()
    s, p, o = e2[Tt[:, 0].long()].double(), r2[Tt[:, 1].long()].double(), e2[Tt[:, 2].long()].double()
    if model == "TransE_L1":
        val = -((s + p) - o).abs().sum(1); mag = val.abs()
    elif model == "TransE_L2":
        val = -(((s + p) - o) ** 2).sum(1).sqrt(); mag = val.abs()
    elif model == "DistMult":
        val = (s * p * o).sum(1); mag = (s * p * o).abs().sum(1)
    else:
        val, mag = complex_score_f64(ent2, rel, Tt)
        val, mag = val * scale, mag * scale
    well = val.abs() >= 0.1 * mag
    assert int(well.sum()) >= 1000, int(well.sum())
    rel_err = ((got - val).abs() / val.abs())[well]
    print("%s k=200, |E|=1M: %d well-conditioned triples, max relative score error %.3g" % (model, int(well.sum()), float(rel_err.max())))
    assert float(rel_err.max()) <= 1e-4
