"""GPU tests of the public API (fit / predict / evaluate_performance / save-restore), mirroring the
reference's own model tests (tests/emgraph/models/test_models.py, tests/emgraph/evaluation/test_protocol.py)
plus end-to-end parity of fit() against an oracle training loop driven by the same Philox draws."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import emgraph_oracle as orc  # noqa: E402

F32 = np.float32
TOY = np.array([["a", "y", "b"], ["b", "y", "a"], ["a", "y", "c"], ["c", "y", "a"], ["a", "y", "d"], ["c", "y", "d"],
                ["b", "y", "c"], ["f", "y", "e"]])


def _models():
    from emgraph_amd.models import ComplEx, DistMult, HolE, TransE
    return {"TransE": TransE, "DistMult": DistMult, "ComplEx": ComplEx, "HolE": HolE}


def synth_graph(n_ent=60, n_rel=4, n=600, seed=0):
    rs = np.random.RandomState(seed)
    X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
    # make sure every entity/relation id occurs so that ids == labels after np.unique mapping
    X[:n_ent, 0] = np.arange(n_ent)
    X[:n_rel, 1] = np.arange(n_rel)
    return X.astype(np.int64)


# ------------------------------------------------------------------------------------------------
# fit() end to end vs an oracle loop (same batching, same Philox corruptions, same optimizer rule)
# ------------------------------------------------------------------------------------------------
def _kink_distance(model, E, R, xb, x_negs, eta, loss, reg, k):
    """how close this step comes to a point where the loss's (sub)gradient jumps: the hinge of pairwise / absolute_margin at 0
    (pairwise.py:66-70, absolute_margin.py), TransE-L1's sign per coordinate and TransE-L2's direction at distance zero (TransE.py:208-216), the LP regulariser's sign at p = 1
    (lp.py:107-113).  Relative: a score distance over 1 + |score|, a coordinate distance over the coordinate scale."""
    d = np.inf
    if loss in ("pairwise", "absolute_margin"):
        sp = orc.score_triples(model, E, R, xb, k=k).astype(np.float64)
        for xn in x_negs:
            sn = orc.score_triples(model, E, R, xn, k=k).astype(np.float64)
            t = 1.0 + sn if loss == "absolute_margin" else 1.0 - np.tile(sp, eta) + sn
            d = min(d, float(np.min(np.abs(t) / (1.0 + np.abs(sn)))))
    if model == "TransE_L1":
        for xx in [xb] + list(x_negs):
            dd = (E[xx[:, 0]] + R[xx[:, 1]]) - E[xx[:, 2]]
            nz = np.abs(dd[dd != 0])          # (an exact zero has gradient 0 in both arithmetics: s == o collisions)
            if nz.size:
                d = min(d, float(nz.min() / max(1e-30, np.abs(E).mean())))
    if model == "TransE_L2":     # the norm's gradient d / ||d|| has no direction at d = 0 (TransE.py:208-216, ord = 2)
        for xx in [xb] + list(x_negs):
            nn = np.linalg.norm(((E[xx[:, 0]] + R[xx[:, 1]]) - E[xx[:, 2]]).astype(np.float64), axis=1)
            nz = nn[nn != 0]
            if nz.size:
                d = min(d, float(nz.min() / max(1e-30, np.abs(E).mean() * np.sqrt(E.shape[1]))))
    if reg is not None and reg["p"] == 1:
        for W in (E, R):
            nz = np.abs(W[W != 0])
            if nz.size:
                d = min(d, float(nz.min() / max(1e-30, np.abs(W).mean())))
    return d


def oracle_fit(model, k, X_idx, ent0, rel0, eta, epochs, batches_count, seed, loss, loss_params, opt, lr,
               sides=("s,o",), reg=None, kink=None):
    """``kink`` (dict, optional): receives 'min' = the closest any step came to a jump of the gradient (_kink_distance)"""
    E, R = ent0.copy(), rel0.copy()
    n_ent = E.shape[0]
    stE, stR = orc.opt_init(opt, E.shape), orc.opt_init(opt, R.shape)
    losses = []
    for epoch in range(1, epochs + 1):
        tot = 0.0
        for b, xb in enumerate(orc.batches(X_idx, batches_count), 1):
            if len(xb) == 0:
                continue
            x_negs = []
            for sd, side in enumerate(sides):
                counter = ((epoch - 1) * batches_count + (b - 1)) * len(sides) + sd
                x_negs.append(orc.generate_corruptions_for_fit_philox(xb, eta=eta, corrupt_side=side,
                                                                      entities_size=n_ent, seed=seed, counter=counter))
            val, _, _ = orc.model_loss(model, E, R, xb, eta, loss, loss_params, sides, x_negs,
                                       regularizer=reg, k=k)
            tot += float(val)
            if kink is not None:
                kink["min"] = min(kink.get("min", np.inf), _kink_distance(model, E, R, xb, x_negs, eta, loss, reg, k))
            dE, dR = orc.train_grads(model, E, R, xb, eta, loss, loss_params, x_negs, k=k)
            tE = np.zeros(n_ent, bool)
            tR = np.zeros(R.shape[0], bool)
            for xx in [xb] + x_negs:
                tE[xx[:, 0]] = True
                tE[xx[:, 2]] = True
            tR[xb[:, 1]] = True
            if reg is not None:
                lam, p = reg["lam"], reg["p"]
                dE = dE + lam * p * np.abs(E.astype(np.float64)) ** (p - 1) * np.sign(E)
                dR = dR + lam * p * np.abs(R.astype(np.float64)) ** (p - 1) * np.sign(R)
                tE[:] = True
                tR[:] = True
            E = orc.opt_apply(opt, E, dE, stE, lr=lr, touched=None if opt == "adam" else tE)
            R = orc.opt_apply(opt, R, dR, stR, lr=lr, touched=None if opt == "adam" else tR)
        losses.append(tot)
    return E, R, losses


@pytest.mark.parametrize("name,loss,opt", [
    ("TransE", "pairwise", "adagrad"), ("TransE", "nll", "sgd"), ("DistMult", "nll", "adam"),
    ("ComplEx", "nll", "momentum"), ("ComplEx", "multiclass_nll", "adam"), ("HolE", "self_adversarial", "adagrad"),
    ("DistMult", "absolute_margin", "sgd"),
])
def test_fit_matches_oracle_training_loop(name, loss, opt):
    cls = _models()[name]
    k, eta, epochs, bc, seed, lr = 8, 3, 3, 4, 7, 0.05
    X = synth_graph()
    n_ent, n_rel = 60, 4
    rs = np.random.RandomState(1)
    ki = 2 * k if name in ("ComplEx", "HolE") else k
    ent0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    rel0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    m = cls(k=k, eta=eta, epochs=epochs, batches_count=bc, seed=seed, loss=loss, optimizer=opt,
            optimizer_params={"lr": lr}, initializer="constant", initializer_params={"entity": ent0, "relation": rel0})
    m.fit(X)
    assert m.ent_to_idx == {i: i for i in range(n_ent)}
    omodel = "TransE_L1" if name == "TransE" else name
    E, R, _ = oracle_fit(omodel, k, X.astype(np.int32), ent0, rel0, eta, epochs, bc, seed, loss, None, opt, lr)
    got_E, got_R = m.trained_model_params
    # 12 optimizer steps of fp32 arithmetic in a different summation order
    np.testing.assert_allclose(got_E, E, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(got_R, R, rtol=2e-3, atol=2e-5)
    # predict() == oracle scores of the fitted parameters (fp32 within 1e-4 relative)
    Xt = X[:50]
    np.testing.assert_allclose(m.predict(Xt), orc.score_triples(omodel, got_E, got_R, Xt.astype(np.int32), k=k),
                               rtol=1e-4, atol=1e-5)


def test_fit_with_lp_regulariser_and_two_sides_matches_oracle():
    from emgraph_amd.models import ComplEx
    k, eta, epochs, bc, seed, lr = 6, 2, 2, 3, 3, 0.05
    X = synth_graph(seed=4)
    rs = np.random.RandomState(2)
    ent0 = (rs.randn(60, 2 * k) * 0.3).astype(F32)
    rel0 = (rs.randn(4, 2 * k) * 0.3).astype(F32)
    for opt in ("sgd", "adagrad"):
        m = ComplEx(k=k, eta=eta, epochs=epochs, batches_count=bc, seed=seed, loss="nll", optimizer=opt,
                    optimizer_params={"lr": lr}, regularizer="LP", regularizer_params={"lambda": 0.01, "p": 2},
                    embedding_model_params={"corrupt_side": ["s", "o"]}, initializer="constant",
                    initializer_params={"entity": ent0, "relation": rel0})
        m.fit(X)
        E, R, _ = oracle_fit("ComplEx", k, X.astype(np.int32), ent0, rel0, eta, epochs, bc, seed, "nll", None, opt, lr,
                             sides=("s", "o"), reg={"lam": 0.01, "p": 2})
        np.testing.assert_allclose(m.trained_model_params[0], E, rtol=2e-3, atol=2e-5, err_msg=opt)
        np.testing.assert_allclose(m.trained_model_params[1], R, rtol=2e-3, atol=2e-5, err_msg=opt)


@pytest.mark.parametrize("opt", ["sgd", "momentum", "adagrad", "adam"])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_folded_lp_regulariser_reaches_untouched_rows(opt, p):
    """LPRegularizer (regularizers/lp.py:81-113) penalises the FULL tables, so rows no triple of a batch touches still
    move (and, with a stateful optimizer, still update their state).  2 000 entities, 90 training triples per batch:
    almost every row is 'untouched' in every step and takes the dense pass; touched rows take the apply kernel with the
    regulariser's gradient folded in.  Against the oracle loop (dense gradient, one optimizer
    update per row per step); the loss includes lambda * sum |w|^p of the pre-update tables."""
    from emgraph_amd.models import DistMult
    k, eta, epochs, bc, seed, lr, lam = 12, 2, 2, 3, 4, 0.05, 0.02
    n_ent, n_rel = 2000, 5
    rs = np.random.RandomState(7)
    X = np.stack([rs.randint(0, n_ent, 270), rs.randint(0, n_rel, 270), rs.randint(0, n_ent, 270)], 1)
    X[:n_rel, 1] = np.arange(n_rel)
    ent0 = (rs.randn(n_ent, k) * 0.3).astype(F32)
    rel0 = (rs.randn(n_rel, k) * 0.3).astype(F32)
    # every entity id must exist in the mapping: append one inert triple per unseen id? -> use from_idx-style ids directly
    ids = np.unique(np.concatenate([X[:, 0], X[:, 2]]))
    remap = {e: i for i, e in enumerate(ids)}
    Xd = np.stack([[remap[a] for a in X[:, 0]], X[:, 1], [remap[a] for a in X[:, 2]]], 1)
    n_seen = len(ids)
    ent0 = ent0[:n_seen]
    m = DistMult(k=k, eta=eta, epochs=epochs, batches_count=bc, seed=seed, loss="nll", optimizer=opt,
                 optimizer_params={"lr": lr}, regularizer="LP", regularizer_params={"lambda": lam, "p": p},
                 embedding_model_params={"negative_corruption_entities": 40}, initializer="constant",
                 initializer_params={"entity": ent0, "relation": rel0})
    m.fit(Xd)
    assert m._trainer.reg is not None and m._trainer.fused
    # oracle loop with negatives restricted to the first 40 ids: most of the ~500 rows see no triple in a batch
    E, R = ent0.copy(), rel0.copy()
    stE, stR = orc.opt_init(opt, E.shape), orc.opt_init(opt, R.shape)
    total = 0.0
    for epoch in range(1, epochs + 1):
        for b, xb in enumerate(orc.batches(Xd.astype(np.int32), bc), 1):
            counter = (epoch - 1) * bc + (b - 1)
            xn = orc.generate_corruptions_for_fit_philox(xb, eta=eta, corrupt_side="s,o", entities_size=40, seed=seed, counter=counter)
            val, _, _ = orc.model_loss("DistMult", E, R, xb, eta, "nll", None, ("s,o",), [xn], regularizer={"lam": lam, "p": p}, k=k)
            total += float(val)
            dE, dR = orc.train_grads("DistMult", E, R, xb, eta, "nll", None, [xn], k=k)
            dE = dE + lam * p * np.abs(E.astype(np.float64)) ** (p - 1) * np.sign(E)
            dR = dR + lam * p * np.abs(R.astype(np.float64)) ** (p - 1) * np.sign(R)
            allE, allR = np.ones(len(E), bool), np.ones(len(R), bool)
            E = orc.opt_apply(opt, E, dE, stE, lr=lr, touched=None if opt == "adam" else allE)
            R = orc.opt_apply(opt, R, dR, stR, lr=lr, touched=None if opt == "adam" else allR)
    np.testing.assert_allclose(m.trained_model_params[0], E, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(m.trained_model_params[1], R, rtol=2e-3, atol=2e-5)
    assert not np.allclose(m.trained_model_params[0][100:], ent0[100:], atol=1e-7)      # untouched rows did move
    np.testing.assert_allclose(sum(m.epoch_losses), total, rtol=1e-4)                   # data term + lambda * sum |w|^p


@pytest.mark.parametrize("norm,loss,opt", [(3, "pairwise", "adagrad"), (1.5, "nll", "sgd"), (np.inf, "pairwise", "sgd"),
                                           (4, "multiclass_nll", "adam"), (3, "self_adversarial", "momentum")])
def test_fit_transe_any_norm_matches_oracle_training_loop(norm, loss, opt):
    """TransE.py:208-216: `embedding_model_params['norm']` is tf.norm's ord — any positive order trains (generic kernels for orders
    other than 1 / 2), compared with the oracle loop driven by the same Philox draws"""
    cls = _models()["TransE"]
    k, eta, epochs, bc, seed, lr = 8, 3, 3, 4, 7, 0.05
    X = synth_graph()
    n_ent, n_rel = 60, 4
    rs = np.random.RandomState(1)
    ent0 = (rs.randn(n_ent, k) * 0.3).astype(F32)
    rel0 = (rs.randn(n_rel, k) * 0.3).astype(F32)
    m = cls(k=k, eta=eta, epochs=epochs, batches_count=bc, seed=seed, loss=loss, optimizer=opt, optimizer_params={"lr": lr},
            embedding_model_params={"norm": norm}, initializer="constant", initializer_params={"entity": ent0, "relation": rel0})
    m.fit(X)
    omodel = "TransE_P:%r" % float(norm)
    E, R, losses = oracle_fit(omodel, k, X.astype(np.int32), ent0, rel0, eta, epochs, bc, seed, loss, None, opt, lr)
    got_E, got_R = m.trained_model_params
    if opt != "adam":
        np.testing.assert_allclose(got_E, E, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(got_R, R, rtol=2e-3, atol=2e-5)
    else:   # (Keras Adam's normalised step amplifies the last bit where the true gradient is zero: as in the soak test below)
        for got, exp in ((got_E, E), (got_R, R)):
            err = np.abs(got - exp)
            assert np.median(err) < 2e-4 and err.max() <= 2.5 * lr * epochs * bc
    np.testing.assert_allclose(m.epoch_losses, losses, rtol=2e-4, atol=1e-6)
    Xt = X[:50]
    np.testing.assert_allclose(m.predict(Xt), orc.score_triples(omodel, got_E, got_R, Xt.astype(np.int32)), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_FUZZ_OFFSET", "0")), int(os.environ.get("EMG_FUZZ_OFFSET", "0")) + int(os.environ.get("EMG_FUZZ_SEEDS", "10"))))
def test_fit_random_configurations_match_oracle_training_loop(seed):
    """soak over the configuration space of fit(): a random model / width / eta / loss / optimizer / corruption-side list /
    graph shape (uniform or hub-heavy) per seed, trained for a few batches and compared with the oracle loop driven by the
    same Philox draws (same tolerances as the hand-picked cases above).  EMG_FUZZ_SEEDS widens it, EMG_FUZZ_OFFSET starts it elsewhere."""
    rs = np.random.RandomState(7000 + seed)
    name = str(rs.choice(["TransE", "TransE", "DistMult", "ComplEx", "HolE"]))
    norm = int(rs.choice([1, 2]))
    k = int(rs.choice([3, 5, 8, 13, 16, 24, 33, 50, 64, 100, 130, 200, 260]))
    eta = int(rs.choice([1, 2, 3, 5, 10, 20]))
    loss = str(rs.choice(["pairwise", "nll", "absolute_margin", "self_adversarial", "multiclass_nll"]))
    opt = str(rs.choice(["sgd", "momentum", "adagrad", "adam"]))
    sides = [("s,o",), ("s", "o"), ("o",), ("s",), ("s+o",)][rs.randint(0, 5)]
    n_ent, n_rel = int(rs.randint(20, 1500)), int(rs.randint(1, 9))
    n, bc, epochs, lr = int(rs.randint(60, 900)), int(rs.randint(1, 5)), int(rs.randint(1, 3)), float(rs.choice([0.01, 0.05]))
    if rs.randint(0, 2):   # hub-heavy subjects / objects: long segments in the apply
        w = 1.0 / np.arange(1, n_ent + 1)
        w /= w.sum()
        X = np.stack([rs.choice(n_ent, n, p=w), rs.randint(0, n_rel, n), rs.choice(n_ent, n, p=w)], 1)
    else:
        X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
    ids = np.unique(np.concatenate([X[:, 0], X[:, 2]]))                       # labels == ids after the np.unique mapping
    remap = np.full(n_ent, -1, np.int64)
    remap[ids] = np.arange(len(ids))
    X = np.stack([remap[X[:, 0]], X[:, 1], remap[X[:, 2]]], 1).astype(np.int64)
    rels = np.unique(X[:, 1])
    X[:, 1] = np.searchsorted(rels, X[:, 1])
    n_ent, n_rel = len(ids), len(rels)
    ki = 2 * k if name in ("ComplEx", "HolE") else k
    ent0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    rel0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    emp = {"corrupt_side": list(sides) if len(sides) > 1 else sides[0]}
    if name == "TransE":
        emp["norm"] = norm
    reg, reg_kw = None, {}
    if rs.randint(0, 3) == 0:                                # LP regulariser over the FULL tables (regularizers/lp.py:81-113)
        reg = {"lam": float(rs.choice([0.001, 0.01])), "p": int(rs.choice([1, 2, 3]))}
        reg_kw = dict(regularizer="LP", regularizer_params={"lambda": reg["lam"], "p": reg["p"]})
    m = _models()[name](k=k, eta=eta, epochs=epochs, batches_count=bc, seed=seed, loss=loss, optimizer=opt,
                        optimizer_params={"lr": lr}, embedding_model_params=emp, initializer="constant",
                        initializer_params={"entity": ent0, "relation": rel0}, **reg_kw)
    omodel = ("TransE_L%d" % norm) if name == "TransE" else name
    kink = {}
    E, R, losses = oracle_fit(omodel, k, X.astype(np.int32), ent0, rel0, eta, epochs, bc, seed, loss, None, opt, lr, sides=sides, reg=reg, kink=kink)
    what = str((name, norm, k, eta, loss, opt, sides, n_ent, n_rel, n, bc, epochs, lr, reg))
    if not np.all(np.isfinite(losses)):                      # the reference stops with this message (EmbeddingModel.py:1340-1345)
        with pytest.raises(ValueError, match=r"Loss is (nan|-?inf)"):   # ("Loss is {}. Please change the hyperparameters.", the loss as numpy prints it)
            m.fit(X)
        return
    m.fit(X)
    if max(np.abs(E).max(), np.abs(R).max()) > 20.0:         # a diverging run amplifies the last bit of every sum: nothing to compare
        assert np.all(np.isfinite(m.trained_model_params[0]))
        return
    Xt = X[:40]                                             # predict() on the fitted tables == the oracle's score of the same tables
    np.testing.assert_allclose(m.predict(Xt), orc.score_triples(omodel, m.trained_model_params[0], m.trained_model_params[1],
                                                                Xt.astype(np.int32), k=k), rtol=1e-4, atol=1e-5, err_msg=what)
    if opt != "adam":
        gE, gR = m.trained_model_params
        offE = ~np.isclose(gE, E, rtol=2e-3, atol=2e-5)
        offR = ~np.isclose(gR, R, rtol=2e-3, atol=2e-5)
        if (offE.any() or offR.any()) and kink.get("min", np.inf) < 1e-4:
            # The run came within rounding of a JUMP of its gradient — the hinge of pairwise / absolute_margin at zero, the sign of a
            # coordinate of TransE-L1's difference, the LP regulariser's sign at p = 1 (`_kink_distance`; the oracle says how close).
            # There the two arithmetics (sums in another order in an earlier step) may land on different sides: a pair's whole
            # gradient, or one coordinate of three rows, then differs by a step of lr g, and every later step carries it on (a
            # softmax loss hands a changed score to every negative of its group).  Found by the round-6 soak in 8 of 30 000 seeds
            # (883, 1139, 1819, 7356, 12209, 16148, 16504: `tools/dbg_fuzz_seed.py SEED` prints the rows; after the FIRST step that
            # differs it is one coordinate in +- pairs, or the rows of one pair).  Accepted only as that: few elements, each within a
            # few such steps, the losses within 2e-3 — a wrong kernel fails the thousands of seeds that come near no jump.
            # (a step moves a row by lr x the sum of its contributions, each of size <= 1 per coordinate for the sign / hinge gradients:
            #  a hub row collects many — the bound follows the busiest row, not a constant; 5 of 100 000 seeds sat past 4 lr per step)
            deg = np.bincount(np.concatenate([X[:, 0], X[:, 2]])).max()
            bound = lr * epochs * (4.0 * bc + 2.0 * deg * (1 + eta * len(sides))) * (1 + 1e-6)
            assert np.abs(gE - E).max() <= bound and np.abs(gR - R).max() <= max(bound, lr * epochs * 2.0 * len(X) * (1 + eta * len(sides))), (what, offE.mean(), kink)
            np.testing.assert_allclose(m.epoch_losses, losses, rtol=5e-2, err_msg=what)   # (2e-3 but for runs that diverge: 81777 of the 100 000-seed soak, k = 3 with momentum, drifts by 1.6 %)
            first = offE
            if epochs > 1:   # "few elements" is asked of the FIRST epoch (a later one spreads what the first left: seed 1139, momentum + softmax)
                m1 = _models()[name](k=k, eta=eta, epochs=1, batches_count=bc, seed=seed, loss=loss, optimizer=opt,
                                     optimizer_params={"lr": lr}, embedding_model_params=emp, initializer="constant",
                                     initializer_params={"entity": ent0, "relation": rel0}, **reg_kw)
                m1.fit(X)
                E1, _, l1 = oracle_fit(omodel, k, X.astype(np.int32), ent0, rel0, eta, 1, bc, seed, loss, None, opt, lr, sides=sides, reg=reg)
                first = ~np.isclose(m1.trained_model_params[0], E1, rtol=2e-3, atol=2e-5)
                np.testing.assert_allclose(m1.epoch_losses, l1, rtol=5e-2, err_msg=what)
            assert first.mean() <= 0.12, (what, first.mean(), kink)
            return
        np.testing.assert_allclose(gE, E, rtol=2e-3, atol=2e-5, err_msg=what)
        np.testing.assert_allclose(gR, R, rtol=2e-3, atol=2e-5, err_msg=what)
        np.testing.assert_allclose(m.epoch_losses, losses, rtol=2e-4, atol=1e-6, err_msg=what)
    else:
        # Keras Adam moves a weight by ~lr * g / (|g| + 1e-7): where the true gradient is exactly zero (equal and opposite
        # contributions, e.g. a subject shared by a positive and its corruption under a softmax loss) the last-bit noise
        # of the sum decides a step of up to lr — in the oracle's arithmetic as much as here.  Those elements cannot agree;
        # all the others must, and a wrong gradient would move most of them.
        for got, exp in ((m.trained_model_params[0], E), (m.trained_model_params[1], R)):
            err = np.abs(got - exp)
            bad = err > 2e-5 + 2e-3 * np.abs(exp)
            # (TransE-L1's sign gradient under Adam's normalised step: an undetermined sign — see the other branch — is a whole
            # +- lr_t step, so the median itself moves: seed 7356 of the round-6 soak, 2.05e-4)
            med = max(6e-4, 0.05 * lr) if (name == "TransE" and norm == 1) else 2e-4
            assert np.median(err) < med and err.max() <= 2.5 * lr * epochs * bc, (what, bad.mean(), np.median(err), err.max())


# ------------------------------------------------------------------------------------------------
# the reference's own toy-graph tests (test_models.py:218-335,338-367,389-409,967-992)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kw", [
    ("TransE", dict(batches_count=1, seed=555, epochs=20, k=10, loss="pairwise", loss_params={"margin": 5},
                    optimizer="adagrad", optimizer_params={"lr": 0.1})),
    ("DistMult", dict(batches_count=2, seed=555, epochs=20, k=10, loss="pairwise", loss_params={"margin": 5},
                      optimizer="adagrad", optimizer_params={"lr": 0.1})),
    ("ComplEx", dict(batches_count=1, seed=555, epochs=20, k=10, loss="pairwise", loss_params={"margin": 1},
                     regularizer="LP", regularizer_params={"lambda": 0.1, "p": 2}, optimizer="adagrad",
                     optimizer_params={"lr": 0.1})),
    ("HolE", dict(batches_count=1, seed=555, epochs=20, k=10, loss="pairwise", loss_params={"margin": 1},
                  regularizer="LP", regularizer_params={"lambda": 0.1, "p": 2}, optimizer="adagrad",
                  optimizer_params={"lr": 0.1})),
])
def test_fit_predict_toy_graph(name, kw):
    model = _models()[name](**kw)
    model.fit(TOY)
    y_pred = model.predict(np.array([["f", "y", "e"], ["b", "y", "d"]]))
    assert y_pred.shape == (2,)
    assert y_pred[0] > y_pred[1]


def test_retrain_is_deterministic():
    from emgraph_amd.models import ComplEx
    model = ComplEx(batches_count=1, seed=555, epochs=20, k=10, loss="pairwise", loss_params={"margin": 1},
                    regularizer="LP", regularizer_params={"lambda": 0.1, "p": 2}, optimizer="adagrad",
                    optimizer_params={"lr": 0.1})
    model.fit(TOY)
    y1 = model.predict(np.array([["f", "y", "e"], ["b", "y", "d"]]))
    model.fit(TOY)
    y2 = model.predict(np.array([["f", "y", "e"], ["b", "y", "d"]]))
    np.testing.assert_array_equal(y1, y2)  # bit-identical: no float atomics anywhere in the step


def test_missing_entity_raises_value_error():
    from emgraph_amd.models import ComplEx
    model = ComplEx(batches_count=1, seed=555, epochs=2, k=5)
    model.fit(TOY)
    with pytest.raises(ValueError):
        model.predict(["a", "y", "zzzzzzzzzzz"])
    with pytest.raises(ValueError):
        model.predict(["a", "xxxxxxxxxx", "e"])
    with pytest.raises(ValueError):
        model.predict(["zzzzzzzz", "y", "e"])


def test_not_fitted_raises_runtime_error():
    from emgraph_amd.models import TransE
    with pytest.raises(RuntimeError):
        TransE(k=5).predict(TOY[:1])
    with pytest.raises(RuntimeError):
        TransE(k=5).get_embeddings(["a"])


def test_predict_from_idx_equals_labels_and_embeddings_lookup():
    from emgraph_amd.evaluation import to_idx
    from emgraph_amd.models import DistMult
    model = DistMult(batches_count=2, seed=1, epochs=3, k=7)
    model.fit(TOY)
    y1 = model.predict(TOY)
    y2 = model.predict(to_idx(TOY, model.ent_to_idx, model.rel_to_idx), from_idx=True)
    np.testing.assert_array_equal(y1, y2)
    emb = model.get_embeddings(np.array(["a", "f"]), "entity")
    assert emb.shape == (2, 7)
    np.testing.assert_array_equal(emb[1], model.trained_model_params[0][model.ent_to_idx["f"]])
    assert model.get_embeddings(np.array(["y"]), "relation").shape == (1, 7)
    with pytest.raises(ValueError):
        model.get_embeddings(np.array(["a"]), "bogus")
    assert model.is_fitted_on(TOY) and not model.is_fitted_on(TOY[:3])
    # the reference's extension contract _fn(e_s, e_p, e_o) goes through the same HIP kernel
    E, R = model.trained_model_params
    xi = to_idx(TOY, model.ent_to_idx, model.rel_to_idx)
    np.testing.assert_allclose(model._fn(E[xi[:, 0]], R[xi[:, 1]], E[xi[:, 2]]), y1, rtol=1e-6)


def test_constructor_errors_match_reference():
    from emgraph_amd.models import ComplEx, TransE
    for kw in (dict(loss="nope"), dict(optimizer="nope"), dict(regularizer="nope"), dict(initializer="nope"),
               dict(loss="bce")):
        with pytest.raises(ValueError):
            TransE(**kw)
    with pytest.raises(ValueError):      # (any POSITIVE order trains — TransE.py:208-216 — but not this)
        TransE(k=4, epochs=1, embedding_model_params={"norm": -1}).fit(TOY)
    TransE(k=4, epochs=1, batches_count=1, embedding_model_params={"norm": 3}).fit(TOY)
    assert ComplEx(k=6).internal_k == 12
    m = TransE(k=3, eta=5)
    assert m.get_hyperparameter_dict()["eta"] == 5 and m.get_hyperparameter_dict()["optimizer"] == "adam"


def test_nan_loss_raises_value_error():
    from emgraph_amd.models import DistMult
    ent0 = np.full((6, 4), 1e19, F32)
    rel0 = np.full((1, 4), 1e19, F32)
    m = DistMult(k=4, epochs=1, batches_count=1, loss="pairwise", optimizer="sgd", initializer="constant",
                 initializer_params={"entity": ent0, "relation": rel0})
    with pytest.raises(ValueError, match="Loss is"):
        m.fit(TOY)


# ------------------------------------------------------------------------------------------------
# evaluate_performance (test_protocol.py properties; test_models.py:183-215)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def fitted_complex():
    from emgraph_amd.models import ComplEx
    X = synth_graph(n_ent=200, n_rel=5, n=3000, seed=9)
    m = ComplEx(k=16, eta=5, epochs=15, batches_count=5, seed=0, optimizer="adam", optimizer_params={"lr": 0.05})
    m.fit(X[:2700])
    return m, X[:2700], X[2700:]


def test_evaluate_performance_shapes_and_oracle_ranks(fitted_complex):
    from emgraph_amd.evaluation import evaluate_performance, hits_at_n_score, mr_score, mrr_score
    m, Xtr, Xte = fitted_complex
    E, R = m.trained_model_params
    filt = np.concatenate([Xtr, Xte])
    for side in ("s,o", "s+o", "s", "o"):
        ranks = evaluate_performance(Xte, m, filter_triples=filt, corrupt_side=side)
        assert ranks.shape == ((len(Xte), 2) if side == "s,o" else (len(Xte),))
        raw = evaluate_performance(Xte, m, corrupt_side=side)
        assert np.all(ranks <= raw) and np.any(ranks < raw)          # test_models.py:214-215
        assert raw.min() >= 2 and ranks.min() >= 1                   # SURVEY B-6
    # literal numpy oracle on the first triples: identical up to float noise around exact ties
    Xi = Xte[:25].astype(np.int32)
    got = evaluate_performance(Xte[:25], m, filter_triples=filt, corrupt_side="s,o")
    exp = orc.get_ranks("ComplEx", E, R, Xi, corrupt_side="s,o", strategy="worst", filter_triples=filt, k=16)
    assert np.abs(got - exp).max() <= 1 and (got == exp).mean() > 0.9
    # mr(s) + mr(o) == 2 * mr("s,o")  (test_protocol.py:234,300)
    rs_ = evaluate_performance(Xte, m, filter_triples=filt, corrupt_side="s")
    ro_ = evaluate_performance(Xte, m, filter_triples=filt, corrupt_side="o")
    rso = evaluate_performance(Xte, m, filter_triples=filt, corrupt_side="s,o")
    np.testing.assert_array_equal(rso[:, 0], rs_)
    np.testing.assert_array_equal(rso[:, 1], ro_)
    assert mr_score(rs_) + mr_score(ro_) == pytest.approx(2 * mr_score(rso))
    assert 0 < mrr_score(rso) <= 1 and 0 <= hits_at_n_score(rso, 10) <= 1
    for strat in ("best", "middle"):
        r2 = evaluate_performance(Xte, m, filter_triples=filt, corrupt_side="s,o", ranking_strategy=strat)
        assert np.all(r2 <= rso)


def test_evaluate_performance_entities_subset_and_unseen(fitted_complex):
    from emgraph_amd.evaluation import evaluate_performance
    m, Xtr, Xte = fitted_complex
    subset = list(range(0, 200, 4))
    ranks = evaluate_performance(Xte, m, filter_triples=np.concatenate([Xtr, Xte]), entities_subset=subset,
                                 corrupt_side="s,o")
    assert ranks.max() <= len(subset) + 1  # test_protocol.py:131: ranks bounded by the subset size
    E, R = m.trained_model_params
    exp = orc.get_ranks("ComplEx", E, R, Xte[:20].astype(np.int32), corrupt_side="s,o", strategy="worst",
                        filter_triples=np.concatenate([Xtr, Xte]), corruption_entities=np.array(subset), k=16)
    assert np.abs(ranks[:20] - exp).max() <= 1
    # unseen entities are dropped with filter_unseen=True, ValueError otherwise
    Xu = np.concatenate([Xte[:5], np.array([[100000, 0, 1]])])
    assert evaluate_performance(Xu, m, corrupt_side="o").shape == (5,)
    with pytest.raises(ValueError):
        evaluate_performance(Xu, m, corrupt_side="o", filter_unseen=False)
    with pytest.raises(AssertionError):
        evaluate_performance(Xte, m, corrupt_side="bogus")
    with pytest.raises(AssertionError):
        evaluate_performance(Xte, m, ranking_strategy="bogus")


def test_reference_protocol_cases(fitted_complex):
    """the remaining behaviours the reference's own protocol / model tests check, on a synthetic graph:
    shuffled full entity list == no list (test_protocol.py:134-170), default protocol == filtered 's,o'
    (:174-300), filter without the test triples (:77-101), too-many-entities warning (:35-57),
    predict twice (test_models.py:995-1024), output sizes (:63-98), is_fitted_on (:458-506)."""
    import warnings

    from emgraph_amd.evaluation import evaluate_performance, mrr_score
    from emgraph_amd.evaluation import protocol as proto
    m, Xtr, Xte = fitted_complex
    filt = np.concatenate([Xtr, Xte])
    r_all = evaluate_performance(Xte, m, filt, corrupt_side="s,o")
    ents = list(m.ent_to_idx.keys())
    np.random.RandomState(3).shuffle(ents)
    r_shuf = evaluate_performance(Xte, m, filt, corrupt_side="s,o", entities_subset=ents)
    assert mrr_score(r_all) == mrr_score(r_shuf)
    np.testing.assert_array_equal(r_all, r_shuf)
    r_def = evaluate_performance(Xte, m, filt, use_default_protocol=True, corrupt_side="s+o")
    np.testing.assert_array_equal(r_def, r_all)
    r_nofilt_test = evaluate_performance(Xte, m, Xtr, corrupt_side="s,o")   # filter lacks the test triples
    assert mrr_score(r_nofilt_test) > 0 and np.all(r_nofilt_test >= r_all)
    # warning above TOO_MANY_ENTITIES_TH corruption entities (threshold lowered instead of a 50k-entity graph)
    old = proto.TOO_MANY_ENTITIES_TH
    proto.TOO_MANY_ENTITIES_TH = 100
    try:
        with pytest.warns(UserWarning):
            evaluate_performance(Xte[:3], m, corrupt_side="o")
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            evaluate_performance(Xte[:3], m, corrupt_side="o", entities_subset=list(range(10)))
    finally:
        proto.TOO_MANY_ENTITIES_TH = old
    # predict twice / output sizes / is_fitted_on
    np.testing.assert_array_equal(m.predict(Xte), m.predict(Xte))
    assert m.predict(Xte).shape == (len(Xte),)
    assert m.get_embeddings(Xte[:7, 0]).shape == (7, 32)                       # ComplEx: 2k columns
    assert m.get_embeddings(Xte[:7, 1], embedding_type="relation").shape == (7, 32)
    assert m.is_fitted_on(Xtr) and not m.is_fitted_on(np.array([[99999, 0, 1]]))


def test_save_restore_roundtrip(tmp_path, fitted_complex):
    from emgraph_amd.evaluation import evaluate_performance
    from emgraph_amd.utils import restore_model, save_model
    m, Xtr, Xte = fitted_complex
    path = os.path.join(tmp_path, "model.pkl")
    save_model(m, path)
    m2 = restore_model(path)
    np.testing.assert_array_equal(m.predict(Xte), m2.predict(Xte))
    np.testing.assert_array_equal(evaluate_performance(Xte[:20], m, corrupt_side="s,o"),
                                  evaluate_performance(Xte[:20], m2, corrupt_side="s,o"))
    with pytest.raises(FileNotFoundError):
        restore_model(os.path.join(tmp_path, "nope.pkl"))


def test_early_stopping(tmp_path, fitted_complex):
    from emgraph_amd.models import DistMult
    _, Xtr, Xte = fitted_complex
    # validation triples taken FROM the training set: their filtered MRR rises while the model memorises them and
    # then plateaus, so the best snapshot is neither the first check nor the last epoch
    m = DistMult(k=8, eta=2, epochs=300, batches_count=3, seed=0, optimizer="adam", optimizer_params={"lr": 0.05})
    m.fit(Xtr, early_stopping=True, early_stopping_params={"x_valid": Xtr[:60], "criteria": "mrr", "burn_in": 1,
                                                           "check_interval": 1, "stop_interval": 2,
                                                           "x_filter": np.concatenate([Xtr, Xte])})
    assert m.is_fitted and len(m.trained_model_params) == 2
    # the run stopped early, i.e. training went on in place after the best snapshot was taken: inference must use
    # the SNAPSHOT (EmbeddingModel.py:403-453 always loads trained_model_params), not the live tables
    assert hasattr(m, "early_stopping_epoch") and m.early_stopping_epoch < 300
    E, R = m.trained_model_params
    got = m.predict(Xte[:40])
    np.testing.assert_allclose(got, orc.score_triples("DistMult", E, R, Xte[:40].astype(np.int32), k=8), rtol=1e-4, atol=1e-6)
    live_E = m._trainer.ent.cpu().numpy()
    assert not np.array_equal(live_E, E)          # the trainer did move on after the snapshot
    from emgraph_amd.utils import restore_model, save_model
    save_model(m, os.path.join(tmp_path, "es.pkl"))
    np.testing.assert_array_equal(restore_model(os.path.join(tmp_path, "es.pkl")).predict(Xte[:40]), got)
    with pytest.raises(KeyError):
        DistMult(k=4, epochs=1).fit(Xtr, early_stopping=True, early_stopping_params={})
    with pytest.raises(ValueError):
        DistMult(k=4, epochs=1).fit(Xtr, early_stopping=True,
                                    early_stopping_params={"x_valid": Xte[:10], "criteria": "bogus"})


def test_generate_corruptions_for_fit_public_function():
    from emgraph_amd.evaluation import generate_corruptions_for_fit
    X = synth_graph(n=100).astype(np.int32)
    out = generate_corruptions_for_fit(X, eta=3, corrupt_side="s,o", entities_size=60, rnd=5)
    exp = orc.generate_corruptions_for_fit_philox(X, eta=3, corrupt_side="s,o", entities_size=60, seed=5, counter=0)
    np.testing.assert_array_equal(out, exp)
    out = generate_corruptions_for_fit(X, eta=2, corrupt_side="o", entities_size=0, rnd=1)  # batch entities
    exp = orc.generate_corruptions_for_fit_philox(X, eta=2, corrupt_side="o", entities_size=0, seed=1, counter=0)
    np.testing.assert_array_equal(out, exp)
    with pytest.raises(ValueError):
        generate_corruptions_for_fit(X, corrupt_side="x")


def test_negative_corruption_entity_options():
    from emgraph_amd.models import DistMult
    X = synth_graph(n=300)
    for nce in ("batch", [1, 2, 3, 4, 5], 10):
        m = DistMult(k=4, eta=2, epochs=2, batches_count=3, embedding_model_params={"negative_corruption_entities": nce})
        m.fit(X)
        assert np.isfinite(m.trained_model_params[0]).all()


@pytest.mark.parametrize("opt", ["sgd", "momentum", "adagrad", "adam", "adam_lazy"])
@pytest.mark.parametrize("model,loss", [("ComplEx", "nll"), ("TransE_L1", "pairwise"), ("TransE_L2", "absolute_margin"),
                                        ("DistMult", "multiclass_nll")])
def test_trainer_execution_plans_agree(opt, model, loss):
    """fused + in-place singleton updates + side-stream pipelining == the plain
    forward / loss / backward / segmented-apply sequence (same arithmetic, different data movement)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = orc.MODEL_IDS[model]
    k, n_ent, n_rel, B, eta = 12, 400, 5, 96, 4
    ki = 2 * k if model in ("ComplEx", "HolE") else k
    rs = np.random.RandomState(3)
    ent0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    rel0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, 4 * B), rs.randint(0, n_rel, 4 * B), rs.randint(0, n_ent, 4 * B)], 1).astype(np.int32)
    outs = []
    for plan in (dict(fused=True, inplace=True, pipeline=True), dict(fused=False, inplace=False, pipeline=False),
                 dict(fused=True, inplace=False, pipeline=False), dict(fused=False, inplace=True, pipeline=True)):
        tr = Trainer(mid, ki, 1.0, ent0, rel0, eta, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05},
                     batches_count=4, seed=11, **plan)
        tr.set_training_set(X, B)
        for b in range(4):
            tr.step(b * B, B, 1, b + 1, prefetch=((b + 1) * B, B, 1, b + 2) if b < 3 else None)
        E, R = tr.tables_numpy()
        outs.append((E, R, tr.read_loss()))
    # how many slots were singletons in the last batch (the in-place path must actually be exercised)
    assert int(tr.slots[0]["single"].sum().item()) + int(tr.slots[1]["single"].sum().item()) > 0
    for E, R, ls in outs[1:]:
        # fused and split kernels contract/round the same formulas slightly differently (fp32, few ulp)
        np.testing.assert_allclose(E, outs[0][0], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(R, outs[0][1], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ls, outs[0][2], rtol=1e-5)
    # in-place vs contribution path differ only in data movement (and in which template instantiation the
    # compiler contracted into fmas): equal to fp32 rounding
    np.testing.assert_allclose(outs[1][0], outs[3][0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(outs[0][0], outs[2][0], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# data formats either side of the path: reference-written checkpoints, adapters, the model-selection caller
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["TransE", "ComplEx", "HolE"])
def test_reference_written_checkpoint_predicts_reference_scores(name, golden):
    """a model file pickled by the REFERENCE's save_model (tests/golden/ref_*.model.pkl) restored here predicts what
    the reference's own _fn gave for the same labelled triples (tests/golden/checkpoints.npz)"""
    from emgraph_amd.utils import restore_model
    m = restore_model(os.path.join(os.path.dirname(__file__), "golden", "ref_%s.model.pkl" % name))
    g = golden("checkpoints")
    exp = g["scores_" + name]
    got = m.predict(g["X"])
    assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0))


@pytest.mark.parametrize("name,k,params", [("DistMult", 200, {}), ("ComplEx", 100, {}), ("HolE", 200, {}), ("DistMult", 100, {}),
                                           ("TransE", 200, {}), ("TransE", 75, {"norm": 1}), ("TransE", 200, {"norm": 2}),
                                           ("TransE", 100, {"norm": 2})])
def test_default_eval_precision_returns_the_exact_ranks(name, k, params):
    """evaluate_performance's default ('auto') takes a prefilter + exact re-scoring path on large tables (half-precision
    MFMA for DistMult / ComplEx / HolE and TransE-L2, 16-bit fixed-point sums for TransE-L1): the ranks must equal the
    exact f32 kernel's (eval_precision=0) for every triple, side and strategy"""
    import emgraph_amd.models as M
    from emgraph_amd.evaluation import evaluate_performance
    X = synth_graph(n_ent=40000, n_rel=7, n=60000, seed=5)
    m = getattr(M, name)(k=k, eta=2, epochs=2, batches_count=4, seed=1, optimizer="adam", optimizer_params={"lr": 0.05},
                         embedding_model_params=dict(params))
    m.fit(X)
    Xte = X[:300]
    for strategy in ("worst", "best", "middle"):
        m.embedding_model_params["eval_precision"] = 0
        exact = evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o", ranking_strategy=strategy)
        m.embedding_model_params["eval_precision"] = 2
        forced = evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o", ranking_strategy=strategy)
        del m.embedding_model_params["eval_precision"]
        auto = evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o", ranking_strategy=strategy)
        np.testing.assert_array_equal(np.asarray(forced), np.asarray(exact))
        np.testing.assert_array_equal(np.asarray(auto), np.asarray(exact))
    # 'auto' really is the prefilter path here (undecided pairs were re-scored, no tile fell back to the exact kernel)
    from emgraph_amd.evaluation import rank_triples_device
    ent, rel = m._device_tables()
    st = {}
    rank_triples_device(m._model_id(), ent, rel, m.internal_k, m._scale(), Xte, "s,o", "worst", filter_triples=X,
                        precision="auto", stats=st)
    assert st.get("pairs", 0) > 0 and st.get("fallback", 0) == 0, st
    m.embedding_model_params["eval_precision"] = "fast"
    with pytest.raises(ValueError):
        evaluate_performance(Xte, m, filter_triples=X)


def test_fit_and_evaluate_accept_dataset_adapters(fitted_complex):
    """EmgraphBaseDatasetAdaptor inputs (EmbeddingModel.py:1218-1248, protocol.py:903-929): an adapter holding the
    train / test sets and the filter gives the same parameters and ranks as the bare arrays"""
    from emgraph_amd.datasets import NumpyDatasetAdapter
    from emgraph_amd.evaluation import evaluate_performance
    from emgraph_amd.models import DistMult
    _, Xtr, Xte = fitted_complex
    kw = dict(k=8, eta=3, epochs=4, batches_count=7, seed=2, optimizer="adam", optimizer_params={"lr": 0.02})
    m1 = DistMult(**kw)
    m1.fit(Xtr)
    ad = NumpyDatasetAdapter()
    ad.set_data(Xtr, "train")
    m2 = DistMult(**kw)
    m2.fit(ad)
    assert m1.ent_to_idx == m2.ent_to_idx and m1.rel_to_idx == m2.rel_to_idx
    np.testing.assert_array_equal(m1.trained_model_params[0], m2.trained_model_params[0])
    np.testing.assert_array_equal(m1.trained_model_params[1], m2.trained_model_params[1])
    filt = np.concatenate([Xtr, Xte])
    r1 = evaluate_performance(Xte, m1, filter_triples=filt, corrupt_side="s,o", ranking_strategy="middle")
    ev = NumpyDatasetAdapter()
    ev.use_mappings(m2.rel_to_idx, m2.ent_to_idx)
    ev.set_data(Xte, "test")
    ev.set_filter(filt)
    r2 = evaluate_performance(ev, m2, filter_triples=True, corrupt_side="s,o", ranking_strategy="middle")
    np.testing.assert_array_equal(r1, r2)
    assert m2.eval_config == {} and not m2.is_filtered        # end_evaluation ran (EmbeddingModel.py:2035-2044)
    # the reference's lower-level entry: configure_evaluation_protocol + get_ranks(adapter)
    ev.set_data(Xte, "test")
    ev.set_filter(filt)
    m2.set_filter_for_eval()
    m2.configure_evaluation_protocol({"corrupt_side": "s+o", "ranking_strategy": "worst"})
    r3 = np.array(m2.get_ranks(ev))
    m2.end_evaluation()
    np.testing.assert_array_equal(r3, evaluate_performance(Xte, m1, filter_triples=filt, corrupt_side="s+o"))
    with pytest.raises(ValueError):
        m2.fit([1, 2, 3])
    with pytest.raises(Exception, match="Expected a boolean type"):
        evaluate_performance(ev, m2, filter_triples="yes")


def test_model_selection_calls_replayed_on_this_package():
    """the exact call sequence the reference's select_best_model_ranking issued (tests/golden/
    model_selection_calls.json, recorded while the reference ran) executed against this package's ComplEx and
    evaluate_performance on data of the recorded shapes: constructor kwargs, POSITIONAL fit(X, early_stopping,
    early_stopping_params), evaluate_performance keywords, metrics on the returned ranks, retrain of the best model."""
    import json

    from emgraph_amd.evaluation import evaluate_performance, hits_at_n_score, mr_score, mrr_score
    from emgraph_amd.models import ComplEx
    with open(os.path.join(os.path.dirname(__file__), "golden", "model_selection_calls.json")) as f:
        doc = json.load(f)
    rs = np.random.RandomState(1)
    mk = lambda n: np.stack([rs.randint(0, 30, n), rs.randint(0, 3, n), rs.randint(0, 30, n)], 1)   # noqa: E731
    data = {200: mk(200), 20: mk(20), 25: mk(25)}
    data[200][:30, 0] = np.arange(30)                     # every entity / relation is seen in training
    data[200][:30, 2] = np.arange(30)[::-1]
    data[200][:3, 1] = np.arange(3)
    data[220] = np.concatenate([data[200], data[20]])
    data[245] = np.concatenate([data[200], data[20], data[25]])

    def real(a):
        if isinstance(a, dict) and "ndarray" in a:
            return data[a["ndarray"][0]]
        if isinstance(a, dict):
            return {k: real(v) for k, v in a.items()}
        return a

    model, best, history = None, (0, None), []
    for c in doc["calls"]:
        if c["call"] == "init":
            model = ComplEx(**c["kwargs"])
        elif c["call"] == "fit":
            (best[1] if model is None else model).fit(*[real(a) for a in c["args"]])
        else:
            m = best[1] if model is None else model
            ranks = evaluate_performance(*[real(a) for a in c["args"]], model=m, **{k: real(v) for k, v in c["kwargs"].items()})
            assert ranks.shape == (c["args"][0]["ndarray"][0], 2) and ranks.min() >= 1
            res = {"mrr": mrr_score(ranks), "mr": mr_score(ranks), "hits_1": hits_at_n_score(ranks, n=1),
                   "hits_3": hits_at_n_score(ranks, n=3), "hits_10": hits_at_n_score(ranks, n=10)}
            assert sorted(res) == doc["result_keys"] and 0 < res["mrr"] <= 1 and res["hits_1"] <= res["hits_3"] <= res["hits_10"]
            if model is not None:
                history.append(res)
                if res["mrr"] > best[0]:
                    best = (res["mrr"], model)
                model = None if len(history) == doc["n_history"] else model
    assert len(history) == doc["n_history"] and best[1] is not None and best[1].is_fitted


# ------------------------------------------------------------------------------------------------
# TransE with any positive order of the norm (TransE.py:208-216 hands embedding_model_params['norm'] to tf.norm as ord)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("norm", [3, 1.5, np.inf])
def test_transe_any_norm_predicts_and_ranks(norm):
    """inference with an order other than 1 / 2 (a restored or hand-set model): predict() ==
    -||e_s + r_p - e_o||_ord in float64 within fp32 tolerance, and the ranks of evaluate_performance are those of the
    scores (ties and near-ties of int(score * 1e5) allowed to move a rank by one)"""
    from emgraph_amd.evaluation import evaluate_performance
    from emgraph_amd.models import TransE
    n_ent, n_rel, k = 300, 5, 24
    X = synth_graph(n_ent, n_rel, 900, seed=4)
    m = TransE(k=k, eta=2, epochs=1, batches_count=2, seed=1, embedding_model_params={"norm": norm})
    rs = np.random.RandomState(8)
    E, R = (rs.randn(n_ent, k) * 0.5).astype(F32), (rs.randn(n_rel, k) * 0.5).astype(F32)
    m.trained_model_params = [E, R]
    m.ent_to_idx = {i: i for i in range(n_ent)}
    m.rel_to_idx = {i: i for i in range(n_rel)}
    m.is_fitted = True
    Xt = X[:64]
    d = np.abs(E[Xt[:, 0]].astype(np.float64) + R[Xt[:, 1]] - E[Xt[:, 2]])
    want = -(d.max(1) if np.isinf(norm) else (d ** norm).sum(1) ** (1.0 / norm))
    np.testing.assert_allclose(m.predict(Xt), want, rtol=2e-5)
    ranks = evaluate_performance(Xt, m, filter_triples=X, corrupt_side="o", ranking_strategy="worst")
    # float64 ranks of the same scores, filtered: known objects of (s, p, ?) other than the test object do not count
    known = {}
    for s_, p_, o_ in X:
        known.setdefault((s_, p_), set()).add(o_)
    off = 0
    for (s_, p_, o_), got in zip(Xt, ranks):
        dd = np.abs(E[s_].astype(np.float64) + R[p_] - E.astype(np.float64))
        sc = -(dd.max(1) if np.isinf(norm) else (dd ** norm).sum(1) ** (1.0 / norm))
        ci = np.trunc(sc * 1e5)
        mask = np.ones(n_ent, bool)
        mask[list(known[(s_, p_)] - {o_})] = False
        worst = int((ci[mask] >= ci[o_]).sum())
        assert abs(int(got) - worst) <= 1, (got, worst)
        off += int(got) != worst
    assert off <= 3
