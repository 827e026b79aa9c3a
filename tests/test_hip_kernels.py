"""GPU parity tests, kernel level: every C-ABI entry point against the oracle on seeded inputs.
Bar: bit-exact for integer/index work (corruptions, ranks, counts); fp32 scores/gradients within
1e-4 relative (north_star), tolerance written at each assert."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as co  # noqa: E402
from oracle import emgraph_oracle as orc  # noqa: E402

MODELS = ["TransE_L1", "TransE_L2", "DistMult", "ComplEx", "HolE"]
MID = orc.MODEL_IDS
F32 = np.float32


def dev():
    from emgraph_amd import device
    device.require_gpu()
    return device


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def kint_of(model, k):
    return 2 * k if model in ("ComplEx", "HolE") else k


def scale_of(model, k):
    return float(F32(2 / k)) if model == "HolE" else 1.0


def make_tables(model, k, n_ent, n_rel, seed, scale=0.3, pad=0):
    rs = np.random.RandomState(seed)
    ki = kint_of(model, k)
    E = (rs.randn(n_ent, ki) * scale).astype(F32)
    R = (rs.randn(n_rel, ki) * scale).astype(F32)
    return E, R, ki


def score_tol(model, E, R, x, k):
    """|err| <= 1e-4 * (sum of |terms|): relative to the magnitude the reduction actually sums."""
    es, ep, eo = orc.lookup_embeddings(np.abs(E), np.abs(R), x)
    if model.startswith("TransE"):
        mag = np.sum(es + ep + eo, axis=1)
    elif model == "DistMult":
        mag = np.sum(es * ep * eo, axis=1)
    else:
        sr, si = np.split(es, 2, 1); pr, pi = np.split(ep, 2, 1); orr, oi = np.split(eo, 2, 1)
        mag = np.sum(pr * sr * orr + pr * si * oi + pi * sr * oi + pi * si * orr, axis=1)
        if model == "HolE":
            mag = mag * (2 / k)
    return 1e-4 * mag + 1e-7


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("k", [3, 10, 50, 100, 200, 260])
def test_score_triples_vs_oracle(model, k):
    d = dev()
    E, R, ki = make_tables(model, k, 211, 7, seed=k)
    rs = np.random.RandomState(k + 1)
    n = 333
    x = np.stack([rs.randint(0, 211, n), rs.randint(0, 7, n), rs.randint(0, 211, n)], 1).astype(np.int32)
    got = d.score_triples(MID[model], cu(E), cu(R), ki, scale_of(model, k), cu(x)).cpu().numpy()
    exp = orc.score_triples(model, E, R, x, k=k)
    assert np.all(np.abs(got - exp) <= score_tol(model, E, R, x, k)), np.abs(got - exp).max()


@pytest.mark.parametrize("model", MODELS)
def test_score_triples_golden_fixtures(model, golden):
    """the reference-generated fixtures (tests/golden/scores.npz) through the HIP path"""
    d = dev()
    g = golden("scores")
    for ci, (k, n) in enumerate(g["cases"]):
        tag = "%s_c%d" % (model, ci)
        es, ep, eo = g[tag + "_es"], g[tag + "_ep"], g[tag + "_eo"]
        n = es.shape[0]
        E = np.concatenate([es, eo], 0)
        x = np.stack([np.arange(n), np.arange(n), n + np.arange(n)], 1).astype(np.int32)
        got = d.score_triples(MID[model], cu(E), cu(ep), es.shape[1], scale_of(model, int(k)), cu(x)).cpu().numpy()
        tol = score_tol(model, E, ep, x, int(k))
        assert np.all(np.abs(got - g[tag + "_y"]) <= tol), (tag, np.abs(got - g[tag + "_y"]).max())


def test_score_triples_transe_any_norm_golden_fixtures(golden):
    """the reference-executed TransE._fn with norm = 3, 1.5, 4, inf (tests/golden/scores_transe_p.npz) through the generic
    kernels (EMG_TRANSE_P, the order of the norm in `scale`): 1e-4 relative to the score, the bar of north_star"""
    from emgraph_amd import _lib as L
    d = dev()
    g = golden("scores_transe_p")
    for ci, (k, n, p) in enumerate(g["cases"]):
        tag = "c%d" % ci
        es, ep, eo = g[tag + "_es"], g[tag + "_ep"], g[tag + "_eo"]
        n = es.shape[0]
        E = np.concatenate([es, eo], 0)
        x = np.stack([np.arange(n), np.arange(n), n + np.arange(n)], 1).astype(np.int32)
        got = d.score_triples(L.TRANSE_P, cu(E), cu(ep), es.shape[1], float(p), cu(x)).cpu().numpy()
        want = g[tag + "_y"]
        assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-7), (tag, float(p), np.abs(got - want).max())


def test_score_triples_strided_and_odd_layouts():
    """row stride > k_int, unaligned (odd) strides -> scalar path; huge k -> generic fallback"""
    d = dev()
    rs = np.random.RandomState(5)
    for model, k, ld in [("DistMult", 8, 12), ("ComplEx", 6, 13), ("TransE_L1", 7, 9), ("ComplEx", 700, 1400),
                         ("DistMult", 1030, 1030)]:
        ki = kint_of(model, k)
        Ebuf = (rs.randn(50, ld) * 0.3).astype(F32)
        Rbuf = (rs.randn(5, ld) * 0.3).astype(F32)
        x = np.stack([rs.randint(0, 50, 77), rs.randint(0, 5, 77), rs.randint(0, 50, 77)], 1).astype(np.int32)
        Et, Rt = cu(Ebuf), cu(Rbuf)
        got = d.score_triples(MID[model], Et[:, :ki], Rt[:, :ki], ki, scale_of(model, k), cu(x)).cpu().numpy()
        E, R = Ebuf[:, :ki].copy(), Rbuf[:, :ki].copy()
        exp = orc.score_triples(model, E, R, x, k=k)
        assert np.all(np.abs(got - exp) <= score_tol(model, E, R, x, k)), (model, k)


def test_empty_inputs():
    d = dev()
    E, R, ki = make_tables("DistMult", 8, 10, 2, 0)
    out = d.score_triples(MID["DistMult"], cu(E), cu(R), ki, 1.0, torch.empty((0, 3), dtype=torch.int32, device="cuda"))
    assert out.numel() == 0


# ---------------------------------------------------------------- corruptions (integer: bit-exact)
def test_corruptions_injected_vs_reference_golden(golden):
    d = dev()
    g = golden("corruptions")
    for ci, (xname, eta, side, mode) in enumerate(g["fit_cases"]):
        X = g["toy_X_idx"] if xname == "toy" else g["fit_Xbig"]
        eta = int(eta)
        tag = "fit_c%d" % ci
        if mode == "size":
            n_choices, elist = (8 if xname == "toy" else 50), None
        elif mode == "list":
            elist = g["fit_entities_list"]
            n_choices = len(elist)
        else:
            elist = orc.batch_unique_entities(X)
            n_choices = len(elist)
        from emgraph_amd import _lib as L
        codes = d.corrupt_codes(X.shape[0], eta, L.SIDE_IDS[str(side)], n_choices, "cuda",
                                entities_list=cu(elist) if elist is not None else None,
                                inj_mask=cu(g[tag + "_mask"]), inj_repl=cu(g[tag + "_repl"]))
        out = d.corrupt_expand(cu(X), eta, codes).cpu().numpy()
        np.testing.assert_array_equal(out, g[tag + "_out"], err_msg=str((ci, xname, eta, side, mode)))


def test_reference_own_fit_corruption_goldens():
    """tests/emgraph/evaluation/test_protocol.py:530-605 through the HIP generator (injected draws)"""
    d = dev()
    from emgraph_amd import _lib as L
    Xi = np.array([[0, 0, 1], [2, 0, 3], [4, 0, 5], [1, 1, 6], [0, 1, 7]], np.int32)
    repl = cu(np.array([1, 3, 3, 0, 3], np.int32))
    for side, mask, exp in [("o", None, [[0, 0, 1], [2, 0, 3], [4, 0, 3], [1, 1, 0], [0, 1, 3]]),
                            ("s", None, [[1, 0, 1], [3, 0, 3], [3, 0, 5], [0, 1, 6], [3, 1, 7]]),
                            ("s,o", [1, 1, 0, 1, 1], [[0, 0, 1], [2, 0, 3], [3, 0, 5], [1, 1, 0], [0, 1, 3]])]:
        m = cu(np.array(mask, np.int32)) if mask is not None else cu(np.zeros(5, np.int32))
        codes = d.corrupt_codes(5, 1, L.SIDE_IDS[side], 5, "cuda", inj_mask=m, inj_repl=repl)
        np.testing.assert_array_equal(d.corrupt_expand(cu(Xi), 1, codes).cpu().numpy(), exp)


@pytest.mark.parametrize("side", ["s", "o", "s+o"])
def test_corruptions_philox_bit_exact(side):
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(2)
    X = np.stack([rs.randint(0, 1000, 4097), rs.randint(0, 9, 4097), rs.randint(0, 1000, 4097)], 1).astype(np.int32)
    for (eta, n_choices, seed, counter) in [(1, 1000, 0, 0), (5, 14541, 7, 123456789012), (20, 1000000, 2**40 + 3, 5)]:
        codes = d.corrupt_codes(X.shape[0], eta, L.SIDE_IDS[side], n_choices, "cuda", seed=seed, counter=counter)
        exp = orc.generate_corruptions_for_fit_philox(X, eta=eta, corrupt_side=side, entities_size=n_choices,
                                                      seed=seed, counter=counter)
        np.testing.assert_array_equal(d.corrupt_expand(cu(X), eta, codes).cpu().numpy(), exp)
        np.testing.assert_array_equal(codes.cpu().numpy(),
                                      co.corrupt_codes(X.shape[0], eta, L.SIDE_IDS[side], n_choices, seed, counter))


# ---------------------------------------------------------------- training forward / loss / backward
@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("k,eta", [(10, 1), (100, 3), (200, 20), (6, 2)])
def test_train_forward_vs_oracle(model, k, eta):
    d = dev()
    E, R, ki = make_tables(model, k, 300, 11, seed=3 * k + eta)
    rs = np.random.RandomState(9)
    B = 130
    X = np.stack([rs.randint(0, 300, B), rs.randint(0, 11, B), rs.randint(0, 300, B)], 1).astype(np.int32)
    codes = co.corrupt_codes(B, eta, 2, 300, 11, 22)
    sp, sn = d.train_forward(MID[model], cu(E), cu(R), ki, scale_of(model, k), cu(X), eta, cu(codes))
    xneg = orc.generate_corruptions_for_fit_philox(X, eta=eta, corrupt_side="s+o", entities_size=300, seed=11, counter=22)
    exp_p = orc.score_triples(model, E, R, X, k=k)
    exp_n = orc.score_triples(model, E, R, xneg, k=k)
    assert np.all(np.abs(sp.cpu().numpy() - exp_p) <= score_tol(model, E, R, X, k))
    assert np.all(np.abs(sn.cpu().numpy() - exp_n) <= score_tol(model, E, R, xneg, k))


@pytest.mark.parametrize("loss", list(orc.REQUIRE_SAME_SIZE))
@pytest.mark.parametrize("eta,n_sides", [(1, 1), (2, 1), (20, 1), (3, 2)])
def test_loss_and_grads_vs_oracle(loss, eta, n_sides):
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(17 + eta)
    B = 301
    pos = (rs.randn(B) * 3).astype(F32)
    neg = (rs.randn(n_sides * eta * B) * 3).astype(F32)
    pos[0], neg[0], neg[-1] = 90.0, -80.0, 76.0  # the +-75 clip
    params = {"margin": 1.5, "alpha": 0.7} if loss in ("pairwise", "absolute_margin", "self_adversarial") else {}
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    gp, gn = d.loss(L.LOSS_IDS[loss], cu(pos), cu(neg), B, eta, n_sides, params.get("margin", 1.0),
                    params.get("alpha", 0.5), acc)
    exp_loss = 0.0
    exp_gp = np.zeros(B, np.float64)
    exp_gn = []
    for sd in range(n_sides):
        ns = neg[sd * eta * B:(sd + 1) * eta * B]
        pin = np.tile(pos, eta) if orc.REQUIRE_SAME_SIZE[loss] else pos
        exp_loss += float(orc.loss_apply(loss, pin, ns, eta, params))
        a, b = orc.loss_grads(loss, pos, ns, eta, params)
        exp_gp += a
        exp_gn.append(b)
    np.testing.assert_allclose(acc.item(), exp_loss, rtol=2e-5)
    np.testing.assert_allclose(gp.cpu().numpy(), exp_gp, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gn.cpu().numpy(), np.concatenate(exp_gn), rtol=1e-4, atol=1e-6)


def test_losses_golden_fixtures(golden):
    d = dev()
    from emgraph_amd import _lib as L
    g = golden("losses")
    for ci, (eta, B) in enumerate(g["cases"]):
        eta, B = int(eta), int(B)
        pos, neg = g["c%d_pos" % ci], g["c%d_neg" % ci]
        for name in orc.REQUIRE_SAME_SIZE:
            margin = 3.0 if name == "self_adversarial" else 1.0
            acc = torch.zeros(1, dtype=torch.float64, device="cuda")
            d.loss(L.LOSS_IDS[name], cu(pos), cu(neg), B, eta, 1, margin, 0.5, acc)
            np.testing.assert_allclose(acc.item(), g["c%d_%s" % (ci, name)], rtol=1e-5, err_msg="%s c%d" % (name, ci))


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("k,eta", [(8, 3), (100, 5), (200, 2), (5, 2)])
def test_train_backward_vs_oracle(model, k, eta):
    d = dev()
    n_ent, n_rel, B = 97, 5, 70
    E, R, ki = make_tables(model, k, n_ent, n_rel, seed=k)
    rs = np.random.RandomState(4)
    X = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
    codes = co.corrupt_codes(B, eta, 2, n_ent, 5, 6)
    xneg = orc.generate_corruptions_for_fit_philox(X, eta=eta, corrupt_side="s+o", entities_size=n_ent, seed=5, counter=6)
    gpos = rs.randn(B).astype(F32)
    gneg = rs.randn(B * eta).astype(F32)
    ldc = ((ki + 3) // 4) * 4
    ce = torch.zeros(((2 + eta) * B, ldc), dtype=torch.float32, device="cuda")
    cr = torch.zeros((B, ldc), dtype=torch.float32, device="cuda")
    de = torch.empty((2 + eta) * B, dtype=torch.int32, device="cuda")
    dr = torch.empty(B, dtype=torch.int32, device="cuda")
    d.train_backward(MID[model], cu(E), cu(R), ki, scale_of(model, k), cu(X), eta, cu(codes), cu(gpos), cu(gneg),
                     ce, cr, de, dr)
    dE = np.zeros((n_ent, ki)); dR = np.zeros((n_rel, ki))
    np.add.at(dE, de.cpu().numpy(), ce.cpu().numpy()[:, :ki].astype(np.float64))
    np.add.at(dR, dr.cpu().numpy(), cr.cpu().numpy()[:, :ki].astype(np.float64))
    eE, eR = orc.score_grads(model, E, R, X, gpos, k=k)
    a, b = orc.score_grads(model, E, R, xneg, gneg, k=k)
    eE += a; eR += b
    # destination ids are integer work: exact
    np.testing.assert_array_equal(de.cpu().numpy()[:B], X[:, 0])
    np.testing.assert_array_equal(de.cpu().numpy()[B:2 * B], X[:, 2])
    np.testing.assert_array_equal(de.cpu().numpy()[2 * B:], codes & 0x7fffffff)
    np.testing.assert_array_equal(dr.cpu().numpy(), X[:, 1])
    scale = max(np.abs(eE).max(), 1e-6)
    np.testing.assert_allclose(dE, eE, rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(dR, eR, rtol=1e-4, atol=1e-5 * max(np.abs(eR).max(), 1e-6))


@pytest.mark.parametrize("order", [3.0, 1.5, float("inf")])
@pytest.mark.parametrize("k,eta", [(8, 3), (100, 5), (130, 2)])
def test_transe_any_norm_forward_backward_vs_oracle(order, k, eta):
    """TransE.py:208-216 hands `norm` to tf.norm as ord: orders other than 1 / 2 train through the generic kernels (EMG_TRANSE_P,
    `scale` = the order).  Scores of a positive group within 1e-4 relative, gradient rows (-g sgn(d)|d|^(ord-1) / ||d||^(ord-1);
    ord = inf: the maximum's gradient, shared by tied maxima) rtol 1e-4 against the oracle's float64 autodiff formula.  The first
    triple is dyadic with TWO tied maxima, the second is a zero vector (gradient zero)."""
    from emgraph_amd import _lib as L
    d = dev()
    model = "TransE_P:%r" % order
    n_ent, n_rel, B = 97, 5, 70
    E, R, ki = make_tables("TransE_L1", k, n_ent, n_rel, seed=k)
    rs = np.random.RandomState(4)
    X = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
    X[0], X[1] = (90, 4, 91), (92, 4, 92)          # rows nobody else uses (negatives may still draw them: they read, never write)
    X[2:, 0] %= 90; X[2:, 2] %= 90; X[2:, 1] %= 4
    R[4] = 0.0
    E[90] = 0.0; E[91] = 0.0; E[90, 0] = 0.5; E[91, 1] = 0.5; E[90, 2] = 0.25      # d = (0.5, -0.5, 0.25, 0, ...): two tied maxima
    E[92] = 0.125                                                                      # d = 0
    codes = co.corrupt_codes(B, eta, 2, n_ent, 5, 6)
    xneg = orc.generate_corruptions_for_fit_philox(X, eta=eta, corrupt_side="s+o", entities_size=n_ent, seed=5, counter=6)
    sp, sn = d.train_forward(L.TRANSE_P, cu(E), cu(R), ki, order, cu(X), eta, cu(codes))
    np.testing.assert_allclose(sp.cpu().numpy(), orc.score_triples(model, E, R, X), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sn.cpu().numpy(), orc.score_triples(model, E, R, xneg), rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(d.score_triples(L.TRANSE_P, cu(E), cu(R), ki, order, cu(X)).cpu().numpy(), sp.cpu().numpy())
    gpos = rs.randn(B).astype(F32)
    gneg = rs.randn(B * eta).astype(F32)
    ldc = ((ki + 3) // 4) * 4
    ce = torch.full(((2 + eta) * B, ldc), 7.0, dtype=torch.float32, device="cuda")    # (every slot is written, not accumulated into)
    cr = torch.full((B, ldc), 7.0, dtype=torch.float32, device="cuda")
    de = torch.empty((2 + eta) * B, dtype=torch.int32, device="cuda")
    dr = torch.empty(B, dtype=torch.int32, device="cuda")
    d.train_backward(L.TRANSE_P, cu(E), cu(R), ki, order, cu(X), eta, cu(codes), cu(gpos), cu(gneg), ce, cr, de, dr)
    dE = np.zeros((n_ent, ki)); dR = np.zeros((n_rel, ki))
    np.add.at(dE, de.cpu().numpy(), ce.cpu().numpy()[:, :ki].astype(np.float64))
    np.add.at(dR, dr.cpu().numpy(), cr.cpu().numpy()[:, :ki].astype(np.float64))
    eE, eR = orc.score_grads(model, E, R, X, gpos)
    a, b = orc.score_grads(model, E, R, xneg, gneg)
    eE += a; eR += b
    if np.isinf(order):   # a last-bit difference of two near-equal |d| moves the maximum: compare where float32 and float64 agree on it
        es, ep, eo = orc.lookup_embeddings(E, R, np.concatenate([X, xneg]))
        ad = np.abs((es + ep) - eo).astype(np.float64)
        top2 = np.sort(ad, 1)[:, -2:]
        assert np.all((top2[:, 1] - top2[:, 0] > 1e-6) | (top2[:, 1] == top2[:, 0]))
    np.testing.assert_allclose(dE, eE, rtol=1e-4, atol=1e-5 * max(np.abs(eE).max(), 1e-6))
    np.testing.assert_allclose(dR, eR, rtol=1e-4, atol=1e-5 * max(np.abs(eR).max(), 1e-6))
    assert np.all(np.isfinite(ce.cpu().numpy()[:, :ki]))     # (the zero vector included: gradient zero, not 0 / 0)
    if np.isinf(order):   # tied maxima share the gradient: relation slot 0 holds -g / 2, +g / 2 of the positive (+ its negatives' terms)
        own = np.zeros(ki); own[0], own[1] = -gpos[0] / 2, gpos[0] / 2
        rest = cr.cpu().numpy()[0, :ki] - own
        exp_rest, _ = np.zeros(ki), None
        xn0 = xneg[0::B][:eta]
        _, r0 = orc.score_grads(model, E, R, xn0, gneg[0::B][:eta])
        np.testing.assert_allclose(rest, r0[4], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("model,k", [("TransE_L1", 700), ("TransE_L2", 1030), ("DistMult", 601), ("ComplEx", 520),
                                     ("HolE", 1100)])
def test_wide_rows_train_in_column_blocks(model, k):
    """rows wider than the register-tiled gradient kernels hold (512 columns per half) — the reference accepts any k:
    the separate forward / loss / backward path splits them into column blocks (TransE-L2 with the full norms from the
    forward pass).  Trainer.step (plan path) vs the oracle's gradients through one SGD step, tables within fp32 noise."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    n_ent, n_rel, B, eta, lr = 60, 4, 48, 3, 0.5
    E, R, ki = make_tables(model, k, n_ent, n_rel, seed=k, scale=0.05)   # (scores inside NLL's [-75, 75] clip, nll.py:55-59)
    rs = np.random.RandomState(k)
    X = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
    tr = Trainer(MID[model], ki, scale_of(model, k), E, R, eta, loss="nll", optimizer="sgd", optimizer_params={"lr": lr},
                 batches_count=1, seed=3)
    assert tr.wide and not tr.fused
    tr.set_training_set(X, B)
    tr.step(0, B, 1, 1)
    gotE, gotR = tr.tables_numpy()
    xneg = orc.generate_corruptions_for_fit_philox(X, eta=eta, corrupt_side="s,o", entities_size=n_ent, seed=3, counter=0)
    dE, dR = orc.train_grads(model, E, R, X, eta, "nll", None, [xneg], k=k)
    expE, expR = E - lr * dE, R - lr * dR
    np.testing.assert_allclose(gotE, expE, rtol=2e-4, atol=2e-5 * max(np.abs(dE).max(), 1e-6) * lr + 1e-7)
    np.testing.assert_allclose(gotR, expR, rtol=2e-4, atol=2e-5 * max(np.abs(dR).max(), 1e-6) * lr + 1e-7)
    assert np.abs(dE).max() > 0 and np.abs(gotE - E).max() > 0.5 * lr * np.abs(dE).max()      # the step moved the tables


@pytest.mark.parametrize("opt", ["sgd", "momentum", "adagrad", "adam", "adam_lazy"])
def test_apply_rows_vs_oracle(opt):
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(8)
    n_rows, k, n_c = 50, 12, 400
    W = rs.randn(n_rows, k).astype(F32)
    dest = rs.randint(0, 30, n_c).astype(np.int32)  # rows 30..49 untouched, many duplicates
    contrib = rs.randn(n_c, k).astype(F32)
    Wt = cu(W)
    s0 = s1 = None
    st = orc.opt_init("adam" if opt == "adam_lazy" else opt, W.shape)
    if opt == "momentum":
        s0 = torch.zeros_like(Wt)
    elif opt == "adagrad":
        s0 = torch.full_like(Wt, 0.1)
    elif opt in ("adam", "adam_lazy"):
        s0, s1 = torch.zeros_like(Wt), torch.zeros_like(Wt)
    tag = torch.zeros(n_rows, dtype=torch.int32, device="cuda")
    ws = torch.empty(d.apply_workspace_bytes(n_c, n_rows), dtype=torch.uint8, device="cuda")
    Wexp = W.copy()
    lr, mu, b1, b2, eps = 0.05, 0.9, 0.9, 0.999, 1e-7
    for step in (1, 2, 3):
        lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        d.apply_rows(L.OPT_IDS[opt], Wt, k, s0, s1, tag, step, cu(contrib), cu(dest), n_c,
                     (lr, mu, b1, b2, eps, lr_t), ws)
        G = np.zeros((n_rows, k), np.float64)
        np.add.at(G, dest, contrib.astype(np.float64))
        touched = np.zeros(n_rows, bool); touched[dest] = True
        if opt == "adam_lazy":
            st["t"] += 1
            m, v = st["m"], st["v"]
            g = G.astype(F32)
            m[touched] = F32(b1) * m[touched] + F32(1 - b1) * g[touched]
            v[touched] = F32(b2) * v[touched] + F32(1 - b2) * g[touched] ** 2
            Wexp[touched] = Wexp[touched] - F32(lr_t) * m[touched] / (np.sqrt(v[touched]) + F32(eps))
        else:
            Wexp = orc.opt_apply(opt, Wexp, G, st, lr=lr, momentum=mu, beta1=b1, beta2=b2, eps=eps,
                                 touched=None if opt == "adam" else touched)
        np.testing.assert_allclose(Wt.cpu().numpy(), Wexp, rtol=2e-5, atol=2e-6, err_msg="%s step %d" % (opt, step))
    if opt in ("sgd", "momentum", "adagrad", "adam_lazy"):
        np.testing.assert_array_equal(Wt.cpu().numpy()[30:], W[30:])  # untouched rows bit-identical


def test_opt_ratio_keeps_a_nan_and_survives_a_zero_epsilon():
    """emg_common.hpp::opt_ratio is num * rcp(sqrt(root) + eps) on the hardware's v_rcp_f32, which flushes a denormal argument: a
    zero / denormal eps of a C-ABI caller is raised to FLT_MIN on the HOST (make_opt_params), so an untouched row of Keras Adam's
    dense pass (m = v = 0) stays finite — and there is no max on the device side, so a NaN in the state reaches the weights as a NaN
    (round-5 advisor: an fmaxf there turned it into a huge finite step, hiding a diverged model from fit()'s NaN check).
    Reference rule: training/adam.py:31-48 (Keras Adam)."""
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(4)
    n_rows, k, n_c = 40, 8, 30
    W = rs.randn(n_rows, k).astype(F32)
    dest = rs.randint(0, 20, n_c).astype(np.int32)          # rows 20.. untouched
    dest[0] = 3
    contrib = rs.randn(n_c, k).astype(F32)
    for eps in (0.0, 1e-42):
        Wt, m, v = cu(W), torch.zeros(n_rows, k, device="cuda"), torch.zeros(n_rows, k, device="cuda")
        ws = torch.empty(d.apply_workspace_bytes(n_c, n_rows), dtype=torch.uint8, device="cuda")
        tag = torch.zeros(n_rows, dtype=torch.int32, device="cuda")
        d.apply_rows(L.OPT_ADAM, Wt, k, m, v, tag, 1, cu(contrib), cu(dest), n_c, (0.05, 0.0, 0.9, 0.999, eps, 0.05), ws)
        got = Wt.cpu().numpy()
        assert np.isfinite(got).all(), eps
        np.testing.assert_array_equal(got[20:], W[20:])       # m = v = 0, g = 0: the step is 0 / FLT_MIN = 0
        assert not np.array_equal(got[:20], W[:20])
    Wt, m, v = cu(W), torch.zeros(n_rows, k, device="cuda"), torch.zeros(n_rows, k, device="cuda")
    v[3, 2] = float("nan")        # a touched row
    v[30, 5] = float("nan")       # an untouched row (dense pass)
    d.apply_rows(L.OPT_ADAM, Wt, k, m, v, torch.zeros(n_rows, dtype=torch.int32, device="cuda"), 1, cu(contrib), cu(dest), n_c, (0.05, 0.0, 0.9, 0.999, 1e-7, 0.05), ws)
    got = Wt.cpu().numpy()
    assert np.isnan(got[3, 2]) and np.isnan(got[30, 5]) and np.isnan(got).sum() == 2


def test_lp_fold_forms_agree_by_value_on_signed_zeros_and_denormals():
    """lp_fold_p2 (every device path at p = 2) is (2 lambda) * w where regularizers/lp.py:107-113 reads lambda * p * |w|^(p-1) * sign(w):
    the same value everywhere, the same bits up to the sign of a zero.  Weights of +-0, denormals, tiny and huge values with data
    gradients of +-0 / non-zero: SGD + LP through the apply equals the generic numpy expression BY VALUE (array_equal: -0 == +0),
    for p = 1, 2, 3, and the regulariser's value equals the sum of |w|^p."""
    d = dev()
    from emgraph_amd import _lib as L
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, -3e-39, 1.17549435e-38, -1.17549435e-38, 1e-20, -1e-20, 1.0, -1.0, 3e12, -3e12, 0.5, -0.25], F32)
    k = special.size
    n_rows = 6
    W = np.tile(special, (n_rows, 1))
    dest = np.array([0, 1, 1, 2], np.int32)                       # row 3.. untouched (dense pass), row 1 a segment of two
    contrib = np.zeros((4, k), F32)
    contrib[0] = -0.0                                              # g = -0 on row 0: where the two forms' bits may differ
    contrib[1], contrib[2] = 0.25, -0.25                           # g = +0 by cancellation on row 1
    contrib[3] = np.linspace(-1, 1, k).astype(F32)
    lr, lam = F32(0.05), F32(0.01)
    for p in (1, 2, 3):
        Wt = cu(W)
        acc = torch.zeros(1, dtype=torch.float64, device="cuda")
        ws = torch.empty(d.apply_workspace_bytes(4, n_rows), dtype=torch.uint8, device="cuda")
        d.group_dest(cu(dest), 4, n_rows, ws)
        tag = torch.zeros(n_rows, dtype=torch.int32, device="cuda")
        d.apply_grouped(L.OPT_SGD, Wt, k, None, None, tag, 1, cu(contrib), 4, 0, (float(lr), 0, 0, 0, 0, 0, float(lam), p), ws, lp_accum=acc)
        G = np.zeros((n_rows, k), F32)
        for i, r in enumerate(dest):
            G[r] = G[r] + contrib[i]
        with np.errstate(over="ignore", under="ignore"):
            a = np.abs(W)
            pm1 = np.ones_like(a) if p == 1 else (a if p == 2 else a * a)
            g = (G + (lam * F32(p) * pm1) * np.sign(W)).astype(F32)
            exp = (W - lr * g).astype(F32)
            val = (a.astype(np.float64) ** p).sum() if p == 1 else (lp_pow32(a, p)).astype(np.float64).sum()
        got = Wt.cpu().numpy()
        # (the device runs with denormals ON for f32 — gfx950's default mode — so denormal weights are not flushed)
        np.testing.assert_array_equal(got, exp, err_msg="p=%d" % p)
        np.testing.assert_allclose(acc.item(), val, rtol=1e-6)


def lp_pow32(a, p):
    out = a.copy()
    for _ in range(p - 1):
        out = (out * a).astype(F32)
    return out


@pytest.mark.parametrize("k,n_rows,n_c,hot", [(12, 40, 3000, 0), (50, 300, 5000, 0), (100, 64, 20000, 0), (200, 500, 9000, 0),
                                              (400, 200, 6000, 0), (7, 33, 2000, 0), (30, 50, 700, 0), (52, 2000, 40000, 500)])
def test_apply_rows_sums_in_contribution_order(k, n_rows, n_c, hot):
    """bit-exact: every destination row gets w - lr * (fp32 sum of its contributions IN INDEX ORDER).  Covers the
    16 / 32 / 64 lanes-per-segment variants, the scalar (k % 4 != 0) kernels, segments longer than a wave's window
    (one hot destination collecting `hot` rows; the workspace here is the small one of emg_apply_workspace_bytes, so
    the long-segment block tree of test_apply_rows_long_segments_block_tree is not engaged) and untouched rows."""
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(k + n_c)
    W = rs.randn(n_rows, k).astype(F32)
    dest = rs.randint(0, n_rows - 3, n_c).astype(np.int32)          # the last three rows stay untouched
    if hot:
        dest[rs.choice(n_c, hot, replace=False)] = 5
    contrib = rs.randn(n_c, k).astype(F32)
    Wt = cu(W)
    ws = torch.empty(d.apply_workspace_bytes(n_c, n_rows), dtype=torch.uint8, device="cuda")
    lr = F32(0.05)
    d.apply_rows(L.OPT_SGD, Wt, k, None, None, None, 1, cu(contrib), cu(dest), n_c, (float(lr), 0, 0, 0, 0, 0), ws)
    exp = W.copy()
    order = np.argsort(dest, kind="stable")
    bounds = np.flatnonzero(np.diff(dest[order])) + 1
    for seg in np.split(order, bounds):
        g = np.zeros(k, F32)
        for i in seg:                                               # sequential fp32 adds, index order
            g = g + contrib[i]
        exp[dest[seg[0]]] = W[dest[seg[0]]] - lr * g
    np.testing.assert_array_equal(Wt.cpu().numpy(), exp)


def test_apply_rows_is_deterministic():
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(1)
    n_rows, k, n_c = 64, 100, 20000
    W = rs.randn(n_rows, k).astype(F32)
    dest = rs.randint(0, n_rows, n_c).astype(np.int32)
    contrib = rs.randn(n_c, k).astype(F32)
    outs = []
    for _ in range(3):
        Wt = cu(W)
        ws = torch.empty(d.apply_workspace_bytes(n_c, n_rows), dtype=torch.uint8, device="cuda")
        d.apply_rows(L.OPT_SGD, Wt, k, None, None, None, 1, cu(contrib), cu(dest), n_c, (0.1, 0, 0, 0, 0, 0), ws)
        outs.append(Wt.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])


def test_lp_regularizer_and_clip(golden):
    d = dev()
    g = golden("misc")
    for p in (1, 2, 3):
        acc = torch.zeros(1, dtype=torch.float64, device="cuda")
        d.lp_regularizer(cu(g["lp_w1"]), 5, 0.5, p, 0.0, acc)
        d.lp_regularizer(cu(g["lp_w2"]), 5, 2.0, p, 0.0, acc)
        np.testing.assert_allclose(acc.item(), g["lp_p%d_list" % p], rtol=1e-5)
    # reference's own goldens (tests/emgraph/models/test_regularizers.py:7-38)
    p1 = np.array([[1, -1, 1]], F32); p2 = np.array([[2, -2, 2]], F32)
    for p, lam, exp in [(1, (1.0, 1.0), 9.0), (1, (2.0, 3.0), 24.0), (2, (1.0, 1.0), 15.0), (2, (2.0, 3.0), 42.0)]:
        acc = torch.zeros(1, dtype=torch.float64, device="cuda")
        d.lp_regularizer(cu(p1), 3, lam[0], p, 0.0, acc)
        d.lp_regularizer(cu(p2), 3, lam[1], p, 0.0, acc)
        assert acc.item() == exp
    # gradient step of the regulariser: W -= lr*lambda*p*|W|^(p-1)*sign(W)
    W = g["lp_w1"].copy()
    Wt = cu(W)
    d.lp_regularizer(Wt, 5, 0.5, 2, 0.1, None)
    np.testing.assert_allclose(Wt.cpu().numpy(), W - 0.1 * 0.5 * 2 * W, rtol=1e-6)
    # clip_by_norm(axes=1)
    Wt = cu(W * 3)
    d.clip_rows(Wt, 5, 1.0)
    nrm = np.linalg.norm(W * 3, axis=1, keepdims=True)
    np.testing.assert_allclose(Wt.cpu().numpy(), W * 3 / np.maximum(nrm, 1.0), rtol=1e-6)


# ---------------------------------------------------------------- 1-vs-all ranking
EVAL_MODES = {"s": 0, "o": 1, "s+o": 2, "s,o": 3}


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("k,n_ent", [(8, 50), (50, 300), (200, 700), (7, 131)])
def test_eval_dense_scores_are_the_canonical_chain(model, k, n_ent):
    """MFMA / VALU count kernels == k-ordered fmaf chain, BITWISE (what makes ranks exact)."""
    d = dev()
    E, R, ki = make_tables(model, k, n_ent, 4, seed=k)
    rs = np.random.RandomState(3)
    nq = 37
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    sc = scale_of(model, k)
    Q, pos_int = d.eval_build_queries(MID[model], cu(E), cu(R), ki, sc, cu(T), 3)
    Qe, pe = co.build_queries(MID[model], E, R, ki, sc, T, 3)
    np.testing.assert_array_equal(Q.cpu().numpy()[:, :ki], Qe)       # query vectors bit-exact
    np.testing.assert_array_equal(pos_int.cpu().numpy(), pe)         # positives' comparison ints bit-exact
    S = d.eval_scores_dense(MID[model], Q, cu(E), ki, sc).cpu().numpy()
    Se = co.scores_dense(MID[model], Qe, E, ki, sc)
    np.testing.assert_array_equal(S.view(np.int32), Se.view(np.int32))


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("strategy", ["worst", "best", "middle"])
def test_ranks_exact_arithmetic_vs_literal_oracle(model, strategy):
    """Dyadic embeddings make every fp32 op exact -> any summation order gives the same bits, so the
    HIP ranks must equal the LITERAL numpy restatement of the reference (incl. ties, filters, sides)."""
    from emgraph_amd.evaluation import rank_triples_device
    d = dev()
    rs = np.random.RandomState(21)
    k, n_ent, n_rel = 4, 140, 3
    ki = kint_of(model, k)
    E = (rs.randint(-4, 5, (n_ent, ki)) / 4.0).astype(F32)
    R = (rs.randint(-4, 5, (n_rel, ki)) / 4.0).astype(F32)
    T = np.stack([rs.randint(0, n_ent, 40), rs.randint(0, n_rel, 40), rs.randint(0, n_ent, 40)], 1).astype(np.int32)
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 3000), rs.randint(0, n_rel, 3000),
                                     rs.randint(0, n_ent, 3000)], 1)]).astype(np.int32)
    for side in ("s,o", "s+o", "s", "o"):
        for filt in (None, F):
            got = rank_triples_device(MID[model], cu(E), cu(R), ki, scale_of(model, k), T, side, strategy,
                                      filter_triples=filt)
            exp = orc.get_ranks(model, E, R, T, corrupt_side=side, strategy=strategy, filter_triples=filt, k=k)
            np.testing.assert_array_equal(got, exp, err_msg=str((model, side, strategy, filt is not None)))


@pytest.mark.parametrize("model", MODELS)
def test_ranks_random_embeddings_vs_canonical_oracle(model):
    """random fp32 embeddings, |E|=5000: counts bit-exact vs the C canonical-order oracle; and ranks
    within the +-1e-5 score band of the literal numpy oracle."""
    from emgraph_amd.evaluation import rank_triples_device
    d = dev()
    k, n_ent, n_rel, nq = 64, 5000, 6, 50
    E, R, ki = make_tables(model, k, n_ent, n_rel, seed=77)
    rs = np.random.RandomState(5)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, n_rel, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    sc = scale_of(model, k)
    Q, pos_int = d.eval_build_queries(MID[model], cu(E), cu(R), ki, sc, cu(T), 3)
    gt = torch.zeros(2 * nq, dtype=torch.int32, device="cuda"); eq = torch.zeros_like(gt)
    d.eval_count(MID[model], Q, pos_int, cu(E), ki, sc, gt, eq)
    Qe, pe = co.build_queries(MID[model], E, R, ki, sc, T, 3)
    egt, eeq = co.count(MID[model], Qe, pe, E, ki, sc)
    np.testing.assert_array_equal(gt.cpu().numpy(), egt)
    np.testing.assert_array_equal(eq.cpu().numpy(), eeq)
    # candidate subset + slab offsets
    cand = rs.permutation(n_ent)[:777].astype(np.int32)
    gt.zero_(); eq.zero_()
    d.eval_count(MID[model], Q, pos_int, cu(E), ki, sc, gt, eq, cand=cu(cand))
    egt, eeq = co.count(MID[model], Qe, pe, E, ki, sc, cand=cand)
    np.testing.assert_array_equal(gt.cpu().numpy(), egt)
    np.testing.assert_array_equal(eq.cpu().numpy(), eeq)
    # against the literal oracle: equal unless a candidate's score is within float noise of the positive's
    got = rank_triples_device(MID[model], cu(E), cu(R), ki, sc, T[:12], "s,o", "worst")
    exp = orc.get_ranks(model, E, R, T[:12], corrupt_side="s,o", strategy="worst", k=k)
    assert np.abs(got - exp).max() <= 2


# ---------------------------------------------------------------- bf16 MFMA throughput mode
@pytest.mark.parametrize("model", ["DistMult", "ComplEx", "HolE"])
@pytest.mark.parametrize("k,n_ent", [(8, 200), (100, 700), (37, 300)])
def test_bf16_dense_scores_match_rounded_operands(model, k, n_ent):
    """bf16 MFMA kernel == fp32-accumulated product of the bf16-ROUNDED operands (fragment layout, LDS
    swizzle and tile edges are all exercised; asymmetric random data catches transposes)"""
    d = dev()
    E, R, ki = make_tables(model, k, n_ent, 4, seed=k + 1)
    rs = np.random.RandomState(8)
    nq = 70
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    sc = scale_of(model, k)
    Q, _ = d.eval_build_queries(MID[model], cu(E), cu(R), ki, sc, cu(T), 3)
    kp = d.bf16_ld(ki)
    Eb = d.to_bf16(cu(E), ki, ld_dst=kp)
    Qb = d.to_bf16(Q, ki, ld_dst=kp)
    np.testing.assert_array_equal(Eb.float().cpu().numpy()[:, :ki], torch.from_numpy(E).to(torch.bfloat16).float().numpy())
    assert float(Eb.float().abs()[:, ki:].sum()) == 0.0  # zero padding up to the k-tile
    S = d.eval_scores_dense_bf16(MID[model], Qb, Eb, ki, sc).cpu().numpy()
    ref = Qb.double().cpu().numpy() @ Eb.double().cpu().numpy().T
    if model == "HolE":
        ref = ref * sc
    mag = np.abs(Qb.double().cpu().numpy()) @ np.abs(Eb.double().cpu().numpy()).T + 1e-12
    assert np.max(np.abs(S - ref) / mag) < 1e-5
    cand = rs.permutation(n_ent)[:77].astype(np.int32)
    S2 = d.eval_scores_dense_bf16(MID[model], Qb, Eb, ki, sc, cand=cu(cand)).cpu().numpy()
    np.testing.assert_array_equal(S2, S[:, cand])


@pytest.mark.parametrize("model", ["DistMult", "ComplEx", "HolE"])
def test_bf16_ranks_exact_on_bf16_representable_data(model):
    """values that bf16 holds exactly -> the bf16 path must reproduce the literal oracle's ranks, incl. the
    index-based self exclusion, filters, subsets, ties and all strategies"""
    from emgraph_amd.evaluation import rank_triples_device
    rs = np.random.RandomState(31)
    k, n_ent, n_rel = 4, 150, 3
    ki = kint_of(model, k)
    E = (rs.randint(-4, 5, (n_ent, ki)) / 4.0).astype(F32)
    R = (rs.randint(-4, 5, (n_rel, ki)) / 4.0).astype(F32)
    T = np.stack([rs.randint(0, n_ent, 30), rs.randint(0, n_rel, 30), rs.randint(0, n_ent, 30)], 1).astype(np.int32)
    Fl = np.concatenate([T, np.stack([rs.randint(0, n_ent, 2000), rs.randint(0, n_rel, 2000),
                                      rs.randint(0, n_ent, 2000)], 1)]).astype(np.int32)
    sub = np.arange(0, n_ent, 3)
    for side in ("s,o", "s+o", "o"):
        for strategy in ("worst", "best", "middle"):
            for filt, subset in ((None, None), (Fl, None), (Fl, sub)):
                got = rank_triples_device(MID[model], cu(E), cu(R), ki, scale_of(model, k), T, side, strategy,
                                          filter_triples=filt, entities_subset=subset, precision=1)
                exp = orc.get_ranks(model, E, R, T, corrupt_side=side, strategy=strategy, filter_triples=filt,
                                    corruption_entities=subset, k=k)
                np.testing.assert_array_equal(got, exp, err_msg=str((side, strategy, filt is not None, subset is not None)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "6"))))
def test_ranks_random_protocols_equal_literal_oracle(seed):
    """soak of the ranking protocol against the oracle's literal per-triple evaluation (generate_corruptions_for_eval +
    score + perform_comparison + filter lookups): random model, width, side, strategy, filter set, candidate
    subset and precision mode (0 exact, 2 exact-fast where a prefilter applies) on tables of small dyadic values — every
    score is exact in f32 whatever the summation order, so the ranks must be EQUAL, with all the ties such tables have"""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import rank_triples_device
    dev()
    rs = np.random.RandomState(4000 + seed)
    for _ in range(4):
        model = str(rs.choice(["TransE_L1", "TransE_L2", "DistMult", "ComplEx", "HolE"]))
        k = int(rs.choice([1, 2, 3, 4, 6, 8])) if model != "HolE" else int(rs.choice([1, 2, 4, 8]))   # HolE: 2/k exact in binary
        n_ent, n_rel, nq = int(rs.randint(5, 400)), int(rs.randint(1, 6)), int(rs.randint(1, 150))
        ki = kint_of(model, k)
        E = (rs.randint(-4, 5, (n_ent, ki)) / 4.0).astype(F32)
        R = (rs.randint(-4, 5, (n_rel, ki)) / 4.0).astype(F32)
        T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, n_rel, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
        nf = int(rs.randint(0, 3000))
        filt = None if nf < 300 else np.concatenate([T, np.stack([rs.randint(0, n_ent, nf), rs.randint(0, n_rel, nf),
                                                                  rs.randint(0, n_ent, nf)], 1)]).astype(np.int32)
        subset = None if rs.randint(0, 3) else np.unique(rs.randint(0, n_ent, max(2, n_ent // 2)))
        side, strategy = str(rs.choice(["s,o", "s+o", "s", "o"])), str(rs.choice(["worst", "best", "middle"]))
        precision = int(rs.choice([0, 2]))
        got = rank_triples_device(MID[model], cu(E), cu(R), ki, scale_of(model, k), T, side, strategy, filter_triples=filt,
                                  entities_subset=subset, precision=precision, query_chunk=int(rs.choice([7, 64, 4096])))
        exp = orc.get_ranks(model, E, R, T, corrupt_side=side, strategy=strategy, filter_triples=filt, corruption_entities=subset, k=k)
        np.testing.assert_array_equal(got, exp, err_msg=str((seed, model, k, n_ent, n_rel, nq, side, strategy, nf, subset is not None, precision)))


@pytest.mark.parametrize("B,eta,sides,n_ent,n_rel,xe", [(1000, 5, (2,), 50000, 37, 0), (16384, 3, (0, 1), 300000, 1000, 0),
                                                         (257, 2, (2,), 90, 70000, 90), (40, 1, (1,), 12, 3, 0)])
def test_prepare_batch_equals_separate_calls(B, eta, sides, n_ent, n_rel, xe):
    """emg_prepare_batch == emg_corrupt_codes (per side) + emg_build_dest + emg_group_dest: identical codes,
    destination ids, sorted keys, ORIGINAL positions (stable) and singleton flags; covers the counting grouping with the
    histogram fused into the id kernel and with caller-filled leading rows (separate histogram), a table much larger than
    the batch, and (EMG_GROUPING=sort) the radix-sort backend."""
    d = dev()
    rs = np.random.RandomState(B + eta)
    pos = cu(np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32))
    et = eta * len(sides)
    n_ce, n_cr = xe + (2 + et) * B, B
    mk = lambda n, rows: torch.zeros(d.apply_workspace_bytes(n, rows), dtype=torch.uint8, device="cuda")  # noqa: E731
    # reference: the separate entry points
    codes_ref = torch.empty(B * et, dtype=torch.int32, device="cuda")
    for sd, side in enumerate(sides):
        d.corrupt_codes(B, eta, side, n_ent, "cuda", seed=7, counter=11 + sd, out=codes_ref[sd * eta * B:(sd + 1) * eta * B])
    de_ref = torch.empty(n_ce, dtype=torch.int32, device="cuda")
    dr_ref = torch.empty(n_cr, dtype=torch.int32, device="cuda")
    de_ref[:xe] = torch.arange(xe, dtype=torch.int32, device="cuda")
    d.build_dest(pos, et, codes_ref, de_ref[xe:], dr_ref)
    we_ref, wr_ref = mk(n_ce, n_ent), mk(n_cr, n_rel)
    single_ref = torch.zeros(n_ce, dtype=torch.uint8, device="cuda")
    d.group_dest(de_ref, n_ce, n_ent, we_ref, single_ref)
    d.group_dest(dr_ref, n_cr, n_rel, wr_ref, None)
    # one call
    codes = torch.empty_like(codes_ref)
    de, dr = torch.empty_like(de_ref), torch.empty_like(dr_ref)
    de[:xe] = torch.arange(xe, dtype=torch.int32, device="cuda")
    we, wr = mk(n_ce, n_ent), mk(n_cr, n_rel)
    single = torch.zeros(n_ce, dtype=torch.uint8, device="cuda")
    d.prepare_batch(pos, eta, list(sides), n_ent, codes, de, dr, n_ent, n_rel, we, wr, seed=7, counter0=11,
                    n_extra_ent=xe, single_flags=single)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(codes.cpu().numpy(), codes_ref.cpu().numpy())
    np.testing.assert_array_equal(de.cpu().numpy(), de_ref.cpu().numpy())
    np.testing.assert_array_equal(dr.cpu().numpy(), dr_ref.cpu().numpy())
    np.testing.assert_array_equal(single.cpu().numpy(), single_ref.cpu().numpy())

    def grouped(ws, n):  # workspace layout: keys | vals | ..., each 256-byte aligned (device.apply_workspace_views)
        keys, vals = d.apply_workspace_views(ws, n)
        return keys.cpu().numpy().view(np.uint32), vals.cpu().numpy().view(np.uint32)
    for ws, ws_ref, dest, n in ((we, we_ref, de, n_ce), (wr, wr_ref, dr, n_cr)):
        keys, vals = grouped(ws, n)
        kref, vref = grouped(ws_ref, n)
        dn = dest.cpu().numpy()
        order = np.argsort(dn, kind="stable")
        np.testing.assert_array_equal(keys, dn[order].astype(np.uint32))
        np.testing.assert_array_equal(vals, order.astype(np.uint32))
        np.testing.assert_array_equal(keys, kref)
        np.testing.assert_array_equal(vals, vref)


@pytest.mark.parametrize("model,k,n_ent,nq", [("ComplEx", 200, 5000, 300), ("HolE", 200, 3001, 130),
                                              ("DistMult", 32, 700, 40), ("DistMult", 64, 1024, 128),
                                              ("ComplEx", 96, 2049, 257), ("DistMult", 7, 100, 5),
                                              ("ComplEx", 100, 4000, 200), ("ComplEx", 64, 1500, 100),
                                              ("ComplEx", 200, 9000, 40)])
def test_bf16_stationary_kernel_equals_tile_kernel(model, k, n_ent, nq):
    """the query-stationary LDS-DMA count kernels (v2: query tile in LDS; v3: query fragments in registers, for
    k_pad in {128, 224, 416} and > 128 rows; float thresholds, streamed ring) must produce EXACTLY the
    counters of the v1 tile kernel (integer compare): same MFMA k-order, so (gt, eq) agree bit for bit.
    A candidate list forces the v1 kernel; duplicated rows plant exact ties with the positive."""
    d = dev()
    E, R, ki = make_tables(model, k, n_ent, 4, seed=k + n_ent, scale=0.2)
    rs = np.random.RandomState(k)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    for j in range(0, nq, 3):      # exact ties: other entities carry the true object's / subject's row
        E[rs.randint(0, n_ent, 2)] = E[T[j, 2]]
        E[rs.randint(0, n_ent, 1)] = E[T[j, 0]]
    sc = scale_of(model, k)
    Q, _ = d.eval_build_queries(MID[model], cu(E), cu(R), ki, sc, cu(T), 3)
    ld = d.bf16_ld(ki)
    Eb, Qb = d.to_bf16(cu(E), ki, ld_dst=ld), d.to_bf16(Q, ki, ld_dst=ld)
    pos_int, self_ent = d.eval_pos_int_bf16(MID[model], Eb, ki, sc, cu(T), 3, Qb)
    n_rows = Qb.shape[0]
    got = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
    exp = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
    d.eval_count_bf16(MID[model], Qb, pos_int, self_ent, Eb, ki, sc, got[0], got[1])
    d.eval_count_bf16(MID[model], Qb, pos_int, self_ent, Eb, ki, sc, exp[0], exp[1],
                      cand=torch.arange(n_ent, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), exp.cpu().numpy())
    assert int(exp[1].min()) >= 1 and int(exp[1].max()) >= 2   # every row ties with itself; planted ties are seen
    # single-counter modes: need=1 -> #(>=) = gt + eq, need=2 -> #(>) = gt, for the stationary AND the tile kernels
    both = exp.clone()
    for need, want in ((1, both[0] + both[1]), (2, both[0])):
        for cand in (None, torch.arange(n_ent, dtype=torch.int32, device="cuda")):
            one = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
            d.eval_count_bf16(MID[model], Qb, pos_int, self_ent, Eb, ki, sc, one[0], one[1], cand=cand, need=need)
            np.testing.assert_array_equal(one[0].cpu().numpy(), want.cpu().numpy(), err_msg="need=%d" % need)
    # slab form (ent_offset) and repeated launches accumulate
    d.eval_count_bf16(MID[model], Qb, pos_int, self_ent, Eb[: n_ent // 2], ki, sc, got[0], got[1])
    d.eval_count_bf16(MID[model], Qb, pos_int, self_ent, Eb[: n_ent // 2], ki, sc, exp[0], exp[1],
                      cand=torch.arange(n_ent // 2, dtype=torch.int32, device="cuda"))
    np.testing.assert_array_equal(got.cpu().numpy(), exp.cpu().numpy())


@pytest.mark.parametrize("model,k,n_ent,nq", [("ComplEx", 200, 5000, 300), ("HolE", 200, 4127, 290), ("ComplEx", 200, 131, 140),
                                              ("ComplEx", 64, 1500, 100), ("ComplEx", 100, 4000, 200)])
def test_bf16_wide_wave_kernel_equals_register_stationary_kernel(model, k, n_ent, nq, monkeypatch):
    """v4 (64 query rows per wave, one wave per SIMD, entity-group-major stages, inline-asm MFMA / LDS reads / compare-and-count)
    against v3 in every mode it is instantiated for: both counters and the half-precision prefilter at 400 columns
    (EMG_BF16_V4=2: A/B forms, v3 stays the default there), one counter at 400 / 208 / 128 columns (EMG_BF16_V4=1, the
    default).  Same MFMA k-order: counters bit for bit, the prefilter's undecided pairs as a set.  Tables whose row count is
    not a multiple of 32 exercise the shifted last stage (NaN-patched lanes), row counts that are not multiples of 256 the
    clamped query rows, planted duplicates the ties."""
    d = dev()
    from emgraph_amd.evaluation import ranking as RK
    E, R, ki = make_tables(model, k, n_ent, 4, seed=k + n_ent, scale=0.2)
    rs = np.random.RandomState(k + 1)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    for j in range(0, nq, 3):
        E[rs.randint(0, n_ent, 2)] = E[T[j, 2]]
        E[rs.randint(0, n_ent, 1)] = E[T[j, 0]]
    sc = scale_of(model, k)
    Q, pos_f32 = d.eval_build_queries(MID[model], cu(E), cu(R), ki, sc, cu(T), 3)
    ld = d.bf16_ld(ki)
    Eb, Qb = d.to_bf16(cu(E), ki, ld_dst=ld), d.to_bf16(Q, ki, ld_dst=ld)
    pos_int, self_ent = d.eval_pos_int_bf16(MID[model], Eb, ki, sc, cu(T), 3, Qb)
    n_rows = Qb.shape[0]

    def counts(v4, need):
        monkeypatch.setenv("EMG_BF16_V4", str(v4))
        c = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
        d.eval_count_bf16(MID[model], Qb, pos_int, self_ent, Eb, ki, sc, c[0], c[1], need=need)
        torch.cuda.synchronize()
        return c.cpu().numpy()[: (2 if need == 0 else 1)]

    for need in (1, 2, 0):
        np.testing.assert_array_equal(counts(2, need), counts(0, need), err_msg="need=%d" % need)
    if ki != 400 or n_ent < 128:
        return
    # the prefilter (IEEE half operands, a rigorous band per query row)
    ldh = d.prefilter_ld(ki)
    Eh, Qh = d.to_f16(cu(E), ki, ld_dst=ldh), d.to_f16(Q, ki, ld_dst=ldh)
    band = RK.prefilter_band(Q, Qh, ki, RK.table_norm_bounds(cu(E), Eh, ki))
    n_seg = d.eval_prefilter_segments(n_rows, n_ent, ki)

    def prefilter(v4):
        monkeypatch.setenv("EMG_BF16_V4", str(v4))
        pairs, pcount = RK._pair_buffer(torch.device("cuda"), n_seg)
        cg = torch.zeros(n_rows, dtype=torch.int32, device="cuda")
        d.eval_prefilter_f16(MID[model], Qh, pos_f32, band, Eh, 0, ki, sc, cg, pairs, pcount)
        torch.cuda.synchronize()
        pc = pcount.cpu().numpy()
        assert pc[n_seg] == 0
        cap = pairs.numel() // n_seg
        pn = pairs.cpu().numpy()
        got = [np.sort(pn[i * cap: i * cap + pc[i]]) for i in range(n_seg)]   # per segment: the same pairs in any order
        return cg.cpu().numpy(), pc[:n_seg].copy(), got

    g4, c4, p4 = prefilter(2)
    g3, c3, p3 = prefilter(0)
    np.testing.assert_array_equal(g4, g3)
    np.testing.assert_array_equal(c4, c3)
    for a, b in zip(p4, p3):
        np.testing.assert_array_equal(a, b)
    assert c3.sum() > 0


@pytest.mark.parametrize("model,k,n_ent", [("ComplEx", 200, 6000), ("DistMult", 200, 3000), ("HolE", 100, 2500),
                                           ("ComplEx", 64, 1500)])
def test_bf16_filter_counts_are_what_the_count_kernel_counted(model, k, n_ent):
    """bf16 mode: the filter correction must cancel EXACTLY what the count pass counted.  For every query row, the
    (gt, eq) the filter kernel reports over the row's filter list equal the (gt, eq) the MFMA count kernel itself
    reports over the same entities (candidate list) — same MFMA k-step order, same bits, also on near-ties (rows of
    E are perturbed copies of the true entity's row, so many scores land within an ulp of the positive's) — and a slab
    (ent_offset) sees only its own entities."""
    d = dev()
    E, R, ki = make_tables(model, k, n_ent, 4, seed=k + n_ent, scale=0.2)
    rs = np.random.RandomState(k + 1)
    nq = 48
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    near = []
    for j in range(nq):            # near-ties: copies of the true object's row, some nudged by one bf16 ulp in one column
        ids = rs.randint(0, n_ent, 6)
        E[ids] = E[T[j, 2]]
        E[ids[:3], rs.randint(0, ki, 3)] *= np.float32(1.0078125)
        near.append(ids)
    sc = scale_of(model, k)
    Q, _ = d.eval_build_queries(MID[model], cu(E), cu(R), ki, sc, cu(T), 3)
    ld = d.bf16_ld(ki)
    Eb, Qb = d.to_bf16(cu(E), ki, ld_dst=ld), d.to_bf16(Q, ki, ld_dst=ld)
    pos_int, self_ent = d.eval_pos_int_bf16(MID[model], Eb, ki, sc, cu(T), 3, Qb)
    n_rows = Qb.shape[0]
    self_np = self_ent.cpu().numpy()
    lists = []
    for r in range(n_rows):        # ragged filter lists: the row's own entity, near-ties of its triple, random others, one empty
        if r == 5:
            lists.append(np.zeros(0, np.int32))
            continue
        ids = np.concatenate([[self_np[r]], near[r % nq], rs.randint(0, n_ent, rs.randint(0, 40))])
        lists.append(np.unique(ids).astype(np.int32))
    ptr = np.zeros(n_rows + 1, np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    idx = np.concatenate(lists).astype(np.int32)
    got = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
    d.eval_filter_count_bf16(MID[model], Qb, pos_int, self_ent, Eb, 0, ki, sc, cu(ptr), cu(idx), got[0], got[1])
    exp = np.zeros((2, n_rows), np.int32)
    for r in range(n_rows):
        if len(lists[r]) == 0:
            continue
        one = torch.zeros((2, 1), dtype=torch.int32, device="cuda")
        d.eval_count_bf16(MID[model], Qb[r:r + 1], pos_int[r:r + 1], self_ent[r:r + 1], Eb, ki, sc, one[0], one[1],
                          cand=cu(lists[r]))
        exp[:, r] = one.cpu().numpy()[:, 0]
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    assert exp[1].max() >= 2 and exp[0].sum() > 0            # planted ties and ordinary hits are both exercised
    # slab: entities [off, off + n_local) only
    off, n_local = n_ent // 3, n_ent // 2
    slab = torch.zeros((2, n_rows), dtype=torch.int32, device="cuda")
    d.eval_filter_count_bf16(MID[model], Qb, pos_int, self_ent, Eb[off:off + n_local], off, ki, sc, cu(ptr), cu(idx),
                             slab[0], slab[1])
    exp2 = np.zeros((2, n_rows), np.int32)
    for r in range(n_rows):
        inside = lists[r][(lists[r] >= off) & (lists[r] < off + n_local)]
        if len(inside) == 0:
            continue
        one = torch.zeros((2, 1), dtype=torch.int32, device="cuda")
        d.eval_count_bf16(MID[model], Qb[r:r + 1], pos_int[r:r + 1], self_ent[r:r + 1], Eb, ki, sc, one[0], one[1],
                          cand=cu(inside))
        exp2[:, r] = one.cpu().numpy()[:, 0]
    np.testing.assert_array_equal(slab.cpu().numpy(), exp2)


@pytest.mark.parametrize("k_int,n_ent,nq", [(400, 5000, 300), (200, 777, 65), (37, 100, 3)])
def test_prefilter_band_kernels_match_float64_math(k_int, n_ent, nq):
    """emg_eval_prefilter_bounds / emg_eval_prefilter_band (the error band of precision mode 2) against numpy float64:
    table maxima to 1e-12 relative, every row's band >= the float64 value (rounded UP) and within 1e-6 of it; a row of
    exact half values has dq = 0, a huge row dominates the maxima."""
    d = dev()
    rs = np.random.RandomState(k_int)
    E = (rs.randn(n_ent, k_int) * 0.3).astype(F32)
    E[7] *= 50.0                                   # one dominant row
    E[9] = E[9].astype(np.float16).astype(F32)     # exactly representable: no residual
    Q = (rs.randn(nq, k_int) * 0.8).astype(F32)
    Q[1] = Q[1].astype(np.float16).astype(F32)
    Et, Qt = cu(E), cu(Q)
    ld = d.bf16_ld(k_int)
    Eh, Qh = d.to_f16(Et, k_int, ld_dst=ld), d.to_f16(Qt, k_int, ld_dst=ld)
    b = d.eval_prefilter_bounds(Et, Eh, k_int).cpu().numpy()
    E64, Eh64 = E.astype(np.float64), E.astype(np.float16).astype(np.float64)
    exp_b = np.array([np.linalg.norm(E64, axis=1).max(), np.linalg.norm(Eh64, axis=1).max(),
                      np.linalg.norm(Eh64 - E64, axis=1).max()])
    np.testing.assert_allclose(b, exp_b, rtol=1e-12)
    band = d.eval_prefilter_band(Qt, Qh, k_int, cu(b)).cpu().numpy().astype(np.float64)
    Q64, Qh64 = Q.astype(np.float64), Q.astype(np.float16).astype(np.float64)
    g = 2.0 * (k_int + 32) * 2.0 ** -24
    exp = (np.linalg.norm(Qh64 - Q64, axis=1) * exp_b[1] + np.linalg.norm(Q64, axis=1) * exp_b[2]
           + g * (np.linalg.norm(Qh64, axis=1) * exp_b[1] + np.linalg.norm(Q64, axis=1) * exp_b[0])) * (1.0 + 1e-6)
    assert np.all(band >= exp * (1.0 - 1e-12))
    np.testing.assert_allclose(band, exp, rtol=1e-6)
    # a slab's maxima are its own
    bs = d.eval_prefilter_bounds(Et[100:], Eh[100:], k_int).cpu().numpy() if n_ent > 200 else None
    if bs is not None:
        assert bs[0] < b[0] and abs(bs[0] - np.linalg.norm(E64[100:], axis=1).max()) <= 1e-12 * bs[0]


def test_bf16_rank_agreement_with_exact_path():
    """random trained-scale embeddings: bf16 ranks track the exact f32 ranks (statistical contract)"""
    from emgraph_amd.evaluation import rank_triples_device
    k, n_ent, n_rel, nq = 100, 20000, 5, 200
    E, R, ki = make_tables("ComplEx", k, n_ent, n_rel, seed=3, scale=0.1)
    rs = np.random.RandomState(4)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, n_rel, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    exact = rank_triples_device(MID["ComplEx"], cu(E), cu(R), ki, 1.0, T, "s,o", "worst", filter_triples=T)
    fast = rank_triples_device(MID["ComplEx"], cu(E), cu(R), ki, 1.0, T, "s,o", "worst", filter_triples=T, precision=1)
    rel_err = np.abs(fast - exact) / n_ent
    assert np.median(rel_err) < 0.005 and rel_err.max() < 0.05, (np.median(rel_err), rel_err.max())
    assert abs((1.0 / fast).mean() - (1.0 / exact).mean()) < 0.01  # MRR agrees
    with pytest.raises(ValueError):
        rank_triples_device(MID["TransE_L1"], cu(E[:, :k]), cu(R[:, :k]), k, 1.0, T, precision=1)


# ------------------------------------------------------------------------------------------------
# one-call forms (emg_api.hip) == the fine-grained path
# ------------------------------------------------------------------------------------------------
def test_corrupt_fit_one_call_matches_oracle():
    d = dev()
    rs = np.random.RandomState(2)
    X = np.stack([rs.randint(0, 50, 37), rs.randint(0, 3, 37), rs.randint(0, 50, 37)], 1).astype(np.int32)
    for side, name in ((0, "s"), (1, "o"), (2, "s,o")):
        got = d.corrupt_fit(cu(X), 4, side, entities_size=50, seed=9, counter=5).cpu().numpy()
        exp = orc.generate_corruptions_for_fit_philox(X, eta=4, corrupt_side=name, entities_size=50, seed=9, counter=5)
        np.testing.assert_array_equal(got, exp)
    elist = np.array([3, 7, 11, 40], np.int32)
    got = d.corrupt_fit(cu(X), 2, 2, entities_list=cu(elist), seed=1, counter=0).cpu().numpy()
    exp = orc.generate_corruptions_for_fit_philox(X, entities_list=elist, eta=2, corrupt_side="s,o", seed=1, counter=0)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("model", ["TransE_L1", "DistMult", "ComplEx", "HolE"])
def test_rank_1vsall_one_call_matches_python_path(model):
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, rank_triples_device
    d = dev()
    k, n_ent, n_rel, nq = 24, 700, 5, 60
    E, R, ki = make_tables(model, k, n_ent, n_rel, seed=4)
    rs = np.random.RandomState(6)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, n_rel, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    Fl = np.concatenate([T, np.stack([rs.randint(0, n_ent, 4000), rs.randint(0, n_rel, 4000), rs.randint(0, n_ent, 4000)], 1)]).astype(np.int32)
    F = FilterIndex(Fl)
    sub = np.arange(0, n_ent, 3).astype(np.int32)
    sc = scale_of(model, k)
    for side, sm in (("s", L.EVAL_S), ("o", L.EVAL_O), ("s+o", L.EVAL_SPO), ("s,o", L.EVAL_S_O)):
        for si, strategy in enumerate(("worst", "best", "middle")):
            for subset in (None, sub):
                ptr, idx = F.csr(T, sm, n_ent, subset)
                exp = rank_triples_device(MID[model], cu(E), cu(R), ki, sc, T, side, strategy, filter_triples=F,
                                          entities_subset=subset)
                got = d.rank_1vsall(MID[model], cu(E), cu(R), ki, sc, cu(T), sm, strategy=si,
                                    cand=None if subset is None else cu(subset), filt_ptr=cu(ptr), filt_idx=cu(idx))
                np.testing.assert_array_equal(got.cpu().numpy(), exp, err_msg=str((side, strategy, subset is not None)))
        raw = d.rank_1vsall(MID[model], cu(E), cu(R), ki, sc, cu(T), sm).cpu().numpy()
        np.testing.assert_array_equal(raw, rank_triples_device(MID[model], cu(E), cu(R), ki, sc, T, side, "worst"))
    if model != "TransE_L1":  # bf16 mode through the same call
        ptr, idx = F.csr(T, L.EVAL_S_O, n_ent, None)
        got = d.rank_1vsall(MID[model], cu(E), cu(R), ki, sc, cu(T), L.EVAL_S_O, filt_ptr=cu(ptr), filt_idx=cu(idx), precision_mode=1)
        exp = rank_triples_device(MID[model], cu(E), cu(R), ki, sc, T, "s,o", "worst", filter_triples=F, precision=1)
        np.testing.assert_array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("model,k,n_ent,nq", [("ComplEx", 200, 30000, 300), ("DistMult", 200, 9000, 200), ("HolE", 100, 5000, 150),
                                              ("ComplEx", 300, 8000, 200), ("DistMult", 777, 3000, 140),
                                              ("DistMult", 24, 700, 60), ("TransE_L1", 40, 900, 40), ("ComplEx", 200, 4000, 50),
                                              ("TransE_L1", 200, 20000, 300), ("TransE_L1", 37, 3000, 70), ("TransE_L2", 64, 2000, 90),
                                              ("TransE_L2", 200, 20000, 300), ("TransE_L2", 126, 6000, 140)])
def test_rank_1vsall_one_call_precision_2_equals_precision_0(model, k, n_ent, nq):
    """emg_rank_1vsall(precision_mode = 2) == precision_mode 0 for every side and strategy, filtered, with planted exact
    ties: through the half-precision prefilter where its kernel applies (the first three shapes), through the fixed-point
    prefilter for TransE-L1 (any width), through the MFMA prefilter on the augmented rows for TransE-L2 (k + 2 in a covered
    width), through the exact kernel where none does (an uncovered width, fewer than 129 query rows of a contraction
    model, a candidate list)"""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex
    d = dev()
    E, R, ki = make_tables(model, k, n_ent, 5, seed=k + nq, scale=0.15)
    rs = np.random.RandomState(nq)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 5, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    for j in range(0, nq, 5):
        E[rs.randint(0, n_ent, 2)] = E[T[j, 2]]
    Fl = np.concatenate([T, np.stack([rs.randint(0, n_ent, 3000), rs.randint(0, 5, 3000), rs.randint(0, n_ent, 3000)], 1)]).astype(np.int32)
    F = FilterIndex(Fl)
    sc = scale_of(model, k)
    Et, Rt = cu(E), cu(R)
    for sm in (L.EVAL_S, L.EVAL_O, L.EVAL_SPO, L.EVAL_S_O):
        ptr, idx = F.csr(T, sm, n_ent, None)
        for si in range(3):
            exp = d.rank_1vsall(MID[model], Et, Rt, ki, sc, cu(T), sm, strategy=si, filt_ptr=cu(ptr), filt_idx=cu(idx))
            got = d.rank_1vsall(MID[model], Et, Rt, ki, sc, cu(T), sm, strategy=si, filt_ptr=cu(ptr), filt_idx=cu(idx),
                                precision_mode=2)
            np.testing.assert_array_equal(got.cpu().numpy(), exp.cpu().numpy(), err_msg=str((sm, si)))
    sub = cu(np.arange(0, n_ent, 3).astype(np.int32))
    ptr, idx = F.csr(T, L.EVAL_S_O, n_ent, np.arange(0, n_ent, 3).astype(np.int32))
    np.testing.assert_array_equal(
        d.rank_1vsall(MID[model], Et, Rt, ki, sc, cu(T), L.EVAL_S_O, cand=sub, filt_ptr=cu(ptr), filt_idx=cu(idx), precision_mode=2).cpu().numpy(),
        d.rank_1vsall(MID[model], Et, Rt, ki, sc, cu(T), L.EVAL_S_O, cand=sub, filt_ptr=cu(ptr), filt_idx=cu(idx)).cpu().numpy())


@pytest.mark.parametrize("model,k", [("TransE_L1", 200), ("TransE_L2", 198), ("DistMult", 200)])
def test_rank_1vsall_one_call_precision_2_overflow_falls_back_to_the_exact_kernel(model, k):
    """tables so small that every comparison integer is 0: every candidate is undecided for the plain prefilter, the pair buffer
    overflows and emg_rank_1vsall(precision_mode = 2) must get the ranks of precision_mode 0 all the same — TransE: by redoing the
    tile with the exact kernel; DistMult (round 6): by the prefilter's second form, which proves the ties, and the exact kernel
    only if that overflows too"""
    from emgraph_amd import _lib as L
    d = dev()
    n_ent, nq = 12000, 200
    E, R, ki = make_tables(model, k, n_ent, 5, seed=3, scale=2e-5)
    rs = np.random.RandomState(9)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 5, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    Et, Rt = cu(E), cu(R)
    for si in range(3):
        exp = d.rank_1vsall(MID[model], Et, Rt, ki, scale_of(model, k), cu(T), L.EVAL_S_O, strategy=si).cpu().numpy()
        got = d.rank_1vsall(MID[model], Et, Rt, ki, scale_of(model, k), cu(T), L.EVAL_S_O, strategy=si, precision_mode=2).cpu().numpy()
        np.testing.assert_array_equal(got, exp)
    assert exp.max() > 1000      # (ties everywhere: 'worst' ranks are large)


@pytest.mark.parametrize("model,loss,opt,sides", [("ComplEx", "nll", "adam", ("s,o",)), ("TransE_L2", "pairwise", "sgd", ("s", "o")),
                                                  ("DistMult", "self_adversarial", "adagrad", ("s,o",)),
                                                  ("HolE", "multiclass_nll", "momentum", ("s,o",))])
def test_train_step_one_call_matches_trainer(model, loss, opt, sides):
    """emg_train_step (prepare + score/loss/grad + both applies in one library call) leaves bit-identical tables,
    optimizer state and loss as Trainer's unpipelined step over three batches"""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    d = dev()
    k, n_ent, n_rel, B, eta = 20, 300, 6, 128, 3
    E, R, ki = make_tables(model, k, n_ent, n_rel, seed=12)
    rs = np.random.RandomState(3)
    X = np.stack([rs.randint(0, n_ent, 3 * B), rs.randint(0, n_rel, 3 * B), rs.randint(0, n_ent, 3 * B)], 1).astype(np.int32)
    sc = scale_of(model, k)
    tr = Trainer(MID[model], ki, sc, E, R, eta, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05},
                 corrupt_sides=sides, batches_count=3, seed=5, pipeline=False)
    tr.set_training_set(X, B)
    tr2 = Trainer(MID[model], ki, sc, E, R, eta, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05},
                  corrupt_sides=sides, batches_count=3, seed=5, pipeline=False)   # only as a holder of tables / state
    ws = torch.empty(d.train_step_workspace_bytes(B, eta * len(sides), ki, n_ent, n_rel), dtype=torch.uint8, device="cuda")
    Xt = cu(X)
    for b in range(3):
        tr.step(b * B, B, epoch=1, batch=b + 1)
        tr2.step_count += 1
        d.train_step(MID[model], tr2.ent, tr2.rel, ki, sc, Xt[b * B:(b + 1) * B], eta, tr2.sides, tr2.loss_id, tr2.loss_accum[0:1],
                     tr2.opt_id, tr2.step_count, tr2._hyper(tr2.lr), ws, margin=tr2.margin, alpha=tr2.alpha,
                     states=(tr2.state_ent[0], tr2.state_ent[1], tr2.state_rel[0], tr2.state_rel[1]),
                     tags=(tr2.tag_ent, tr2.tag_rel), n_choices=n_ent, seed=5, counter0=b * len(sides), inplace=tr.inplace)
    torch.cuda.synchronize()
    assert torch.equal(tr.ent, tr2.ent) and torch.equal(tr.rel, tr2.rel)
    for a_, b_ in zip(tr.state_ent + tr.state_rel, tr2.state_ent + tr2.state_rel):
        assert (a_ is None and b_ is None) or torch.equal(a_, b_)
    # (fused-loss path: float-valued partials, exact in any order; separate-loss path — multiclass_nll, self_adversarial — double partials whose
    #  last bit follows the arrival order of the workgroups: 7 of 40 000 runs of the round-6 soak)
    assert tr.read_loss() == pytest.approx(tr2.read_loss(), rel=1e-12)


@pytest.mark.parametrize("order,loss,opt", [(3.0, "pairwise", "adagrad"), (float("inf"), "nll", "sgd")])
def test_train_step_one_call_transe_any_norm(order, loss, opt):
    """emg_train_step with EMG_TRANSE_P (the order of the norm in `scale`): the generic unfused step — same tables, state and loss as
    the Trainer's step (which goes through emg_plan_step)"""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    d = dev()
    k, n_ent, n_rel, B, eta = 20, 300, 6, 128, 3
    E, R, ki = make_tables("TransE_L1", k, n_ent, n_rel, seed=12)
    rs = np.random.RandomState(3)
    X = np.stack([rs.randint(0, n_ent, 3 * B), rs.randint(0, n_rel, 3 * B), rs.randint(0, n_ent, 3 * B)], 1).astype(np.int32)
    mk = lambda: Trainer(L.TRANSE_P, ki, order, E, R, eta, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05}, batches_count=3, seed=5)  # noqa: E731
    tr, tr2 = mk(), mk()
    tr.set_training_set(X, B)
    assert tr.generic and not tr.fused and not tr.inplace
    ws = torch.empty(d.train_step_workspace_bytes(B, eta, ki, n_ent, n_rel), dtype=torch.uint8, device="cuda")
    Xt = cu(X)
    for b in range(3):
        tr.step(b * B, B, epoch=1, batch=b + 1)
        tr2.step_count += 1
        d.train_step(L.TRANSE_P, tr2.ent, tr2.rel, ki, order, Xt[b * B:(b + 1) * B], eta, tr2.sides, tr2.loss_id, tr2.loss_accum[0:1],
                     tr2.opt_id, tr2.step_count, tr2._hyper(tr2.lr), ws, margin=tr2.margin, alpha=tr2.alpha,
                     states=(tr2.state_ent[0], tr2.state_ent[1], tr2.state_rel[0], tr2.state_rel[1]),
                     tags=(tr2.tag_ent, tr2.tag_rel), n_choices=n_ent, seed=5, counter0=b, inplace=True)   # (inplace is ignored for this model)
    torch.cuda.synchronize()
    assert torch.equal(tr.ent, tr2.ent) and torch.equal(tr.rel, tr2.rel)
    assert not torch.equal(tr.ent[:, :ki].cpu(), torch.from_numpy(E))
    # (fused-loss path: float-valued partials, exact in any order; separate-loss path — multiclass_nll, self_adversarial — double partials whose
    #  last bit follows the arrival order of the workgroups: 7 of 40 000 runs of the round-6 soak)
    assert tr.read_loss() == pytest.approx(tr2.read_loss(), rel=1e-12)


# ------------------------------------------------------------------------------------------------
# the rank path against the reference's own execution (tests/golden/ranks.npz, see make_golden.py::gen_ranks)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k_int", [1, 4, 8])
def test_count_epilogue_vs_reference_perform_comparision(golden, k_int):
    """EmbeddingModel.perform_comparision executed from the reference on planted ties / zeros / negative scores /
    sub-quantum differences; the HIP count kernels' compare epilogue must give the same worst / best / middle.
    DistMult with s = p = (1, 0, ...): score(e) = fmaf(1, e_0, 0) = e_0 exactly, so an entity table whose first
    column holds the golden's corruption scores reproduces them bit for bit (k_int 1: unaligned MFMA kernel,
    4 / 8: pipelined MFMA kernel).  TransE-L1 with s = p = 0 gives -|e_0|: checked on the all-negative cases."""
    from emgraph_amd.evaluation.ranking import _cmp
    d = dev()
    g = golden("ranks")
    for ci in range(int(g["cmp_n"])):
        corr, pos = g["cmp_c%d_corr" % ci], float(g["cmp_c%d_pos" % ci])
        n = len(corr)
        for model in ("DistMult", "TransE_L1"):
            if model == "TransE_L1" and (corr.max() > 0 or pos > 0):
                continue
            sign = -1.0 if model == "TransE_L1" else 1.0
            E = np.zeros((n + 2, k_int), F32)
            E[:n, 0] = sign * corr
            E[n, 0] = 0.0 if model == "TransE_L1" else 1.0          # the kept subject
            E[n + 1, 0] = sign * pos                                 # the true object
            R = np.zeros((1, k_int), F32)
            R[0, 0] = 0.0 if model == "TransE_L1" else 1.0
            T = np.array([[n, 0, n + 1]], np.int32)
            Q, pos_int = d.eval_build_queries(MID[model], cu(E), cu(R), k_int, 1.0, cu(T), 1)   # object side
            assert int(pos_int[0]) == int(orc.to_cmp_int(F32(pos)))
            cnt = torch.zeros((2, 1), dtype=torch.int32, device="cuda")
            d.eval_count(MID[model], Q, pos_int, cu(E), k_int, 1.0, cnt[0], cnt[1],
                         cand=torch.arange(n, dtype=torch.int32, device="cuda"))
            gt, eq = (np.int64(v) for v in cnt.cpu().numpy()[:, 0])
            for strat in ("worst", "best", "middle"):
                assert int(_cmp(gt, eq, strat)) == int(g["cmp_c%d_%s" % (ci, strat)]), (ci, model, strat)


@pytest.mark.parametrize("name", ["TransE", "DistMult", "ComplEx", "HolE"])
@pytest.mark.parametrize("precision", [0, 1])
def test_ranks_vs_reference_pieces_on_device(golden, name, precision):
    """filtered and raw ranks assembled from the reference's own _fn / generate_corruptions_for_eval /
    perform_comparision / SQLite lookups (make_golden.py::gen_ranks) through the whole device path, every side
    and strategy, one-call form included.  The embeddings are multiples of 1/4 (exact in bf16 too), so the bf16
    MFMA mode must reproduce them as well."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import FilterIndex, rank_triples_device
    if precision == 1 and name == "TransE":
        pytest.skip("TransE is not a contraction: no bf16 MFMA mode")
    d = dev()
    g = golden("ranks")
    E, R, F, T = g["rk_%s_E" % name], g["rk_%s_R" % name], g["flt_filter"], g["flt_test"]
    om = "TransE_L1" if name == "TransE" else name
    ki = E.shape[1]
    sc = scale_of(om, 4)
    Et, Rt = cu(E), cu(R)
    for si, strat in enumerate(("worst", "best", "middle")):
        exp_so, exp_spo, exp_raw = (g["rk_%s_%s_%s" % (name, strat, s)] for s in ("s,o", "s+o", "raw"))
        kw = dict(precision=precision)
        np.testing.assert_array_equal(rank_triples_device(MID[om], Et, Rt, ki, sc, T, "s,o", strat, filter_triples=F, **kw), exp_so)
        np.testing.assert_array_equal(rank_triples_device(MID[om], Et, Rt, ki, sc, T, "s+o", strat, filter_triples=F, **kw), exp_spo)
        np.testing.assert_array_equal(rank_triples_device(MID[om], Et, Rt, ki, sc, T, "s", strat, filter_triples=F, **kw), exp_so[:, 0])
        np.testing.assert_array_equal(rank_triples_device(MID[om], Et, Rt, ki, sc, T, "o", strat, filter_triples=F, **kw), exp_so[:, 1])
        np.testing.assert_array_equal(rank_triples_device(MID[om], Et, Rt, ki, sc, T, "s,o", strat, **kw), exp_raw)
        ptr, idx = FilterIndex(F).csr(T, L.EVAL_S_O, E.shape[0])
        got = d.rank_1vsall(MID[om], Et, Rt, ki, sc, cu(T), L.EVAL_S_O, strategy=si, filt_ptr=cu(ptr), filt_idx=cu(idx),
                            precision_mode=precision)
        np.testing.assert_array_equal(got.cpu().numpy(), exp_so)


@pytest.mark.parametrize("k,opt", [(400, "sgd"), (100, "adam_lazy"), (50, "sgd"), (72, "adagrad")])
def test_apply_rows_long_segments_block_tree(k, opt):
    """destinations hit by more than 64 contributions (hub entities of a skewed graph, relation rows) are reduced by
    apply_long_kernel: 64-row blocks of the segment summed left to right, then the block sums added left to right —
    a tree defined by the segment alone.  Bit-exact against that definition; segments of <= 64 rows keep the plain
    left-to-right sum.  Long segments sit on ADJACENT destination ids (adjacent sorted positions: the partial-row
    indexing must not collide) and have lengths around the block boundaries."""
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(k)
    n_rows = 300
    lens = {3: 65, 4: 64, 5: 129, 6: 128, 7: 1000, 8: 5003, 9: 66, 10: 63, 200: 70, 201: 777}
    dest = [rs.randint(20, 190, 4000)]
    for row, n in lens.items():
        dest.append(np.full(n, row))
    dest = np.concatenate(dest).astype(np.int32)
    rs.shuffle(dest)
    n_c = len(dest)
    contrib = rs.randn(n_c, k).astype(F32)
    W = rs.randn(n_rows, k).astype(F32)
    Wt = cu(W)
    s0 = s1 = None
    if opt == "adagrad":
        s0 = torch.full_like(Wt, 0.1)
    elif opt == "adam_lazy":
        s0, s1 = torch.zeros_like(Wt), torch.zeros_like(Wt)
    ws = torch.empty(d.apply_workspace_bytes(n_c, n_rows, k), dtype=torch.uint8, device="cuda")
    assert ws.numel() > d.apply_workspace_bytes(n_c, n_rows)
    lr, b1, b2, eps = F32(0.05), F32(0.9), F32(0.999), F32(1e-7)
    d.apply_rows(L.OPT_IDS[opt], Wt, k, s0, s1, None, 1, cu(contrib), cu(dest), n_c, (float(lr), 0.9, 0.9, 0.999, 1e-7, float(lr)), ws)
    exp = W.copy()
    exp_s0 = None if s0 is None else s0.cpu().numpy().copy()     # (state rows: pure multiply-adds of the summed gradient, so BIT-exact)
    exp_s1 = None if s1 is None else np.zeros_like(W)
    order = np.argsort(dest, kind="stable")
    for seg in np.split(order, np.flatnonzero(np.diff(dest[order])) + 1):
        def seq(idx):
            g = np.zeros(k, F32)
            for i in idx:
                g = g + contrib[i]
            return g
        if len(seg) > 64:
            g = np.zeros(k, F32)
            for b0 in range(0, len(seg), 64):
                g = g + seq(seg[b0:b0 + 64])
        else:
            g = seq(seg)
        r = dest[seg[0]]
        if opt == "sgd":
            exp[r] = W[r] - lr * g
        elif opt == "adagrad":
            a = F32(0.1) + g * g
            exp_s0[r] = a
            exp[r] = W[r] - lr * g / (np.sqrt(a) + eps)
        else:
            m = (F32(1) - b1) * g
            v = (F32(1) - b2) * g * g
            exp_s0[r], exp_s1[r] = m, v
            exp[r] = W[r] - lr * m / (np.sqrt(v) + eps)
    if opt == "sgd":
        np.testing.assert_array_equal(Wt.cpu().numpy(), exp)
    else:
        # the reduction tree is pinned bit for bit through the state rows; the weight's x / (sqrt(v) + eps) runs through the hardware's
        # 1-ulp sqrt and reciprocal (emg_common.hpp::opt_ratio): the STEP within 4 ulp of the correctly rounded one
        np.testing.assert_array_equal(s0.cpu().numpy(), exp_s0)
        if s1 is not None:
            np.testing.assert_array_equal(s1.cpu().numpy(), exp_s1)
        step = np.abs(exp - W)
        assert np.all(np.abs(Wt.cpu().numpy() - exp) <= 4 * 2.0 ** -23 * step + 2.0 ** -23 * np.abs(exp))


# ------------------------------------------------------------------------------------------------
# precision 2: exact ranks through the bf16 prefilter (must equal precision 0 bit for bit)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,k,n_ent,nq,scale", [("ComplEx", 200, 30000, 300, 0.1), ("HolE", 200, 9000, 200, 0.3),
                                                     ("DistMult", 200, 20000, 150, 0.1), ("DistMult", 128, 5000, 140, 0.05),
                                                     ("ComplEx", 64, 12000, 260, 1.0), ("ComplEx", 100, 7000, 80, 0.1),
                                                     ("ComplEx", 200, 5000, 40, 0.1), ("ComplEx", 50, 15000, 200, 0.2),
                                                     ("DistMult", 150, 9000, 180, 0.1), ("DistMult", 300, 6000, 150, 0.1),
                                                     ("HolE", 30, 20000, 170, 0.3), ("DistMult", 100, 4000, 130, 0.02),
                                                     ("ComplEx", 128, 8000, 150, 0.1), ("DistMult", 350, 5000, 140, 0.1),
                                                     ("DistMult", 130, 6000, 140, 0.1), ("ComplEx", 85, 5000, 150, 0.2),
                                                     ("DistMult", 353, 4000, 130, 0.1), ("HolE", 9, 30000, 140, 0.5),
                                                     ("DistMult", 401, 3000, 130, 0.1), ("ComplEx", 256, 9000, 200, 0.1),
                                                     ("HolE", 400, 6000, 140, 0.3), ("DistMult", 600, 5000, 150, 0.05),
                                                     ("ComplEx", 330, 4000, 70, 0.1), ("DistMult", 801, 3000, 130, 0.1)])
def test_prefilter_ranks_equal_exact_ranks(model, k, n_ent, nq, scale):
    """precision=2 (bf16 MFMA prefilter with a rigorous per-row error band + exact f32 re-scoring of the undecided
    candidates) == precision=0 (exact f32 MFMA chain) for every side, strategy and filter setting; exact ties are
    planted (other entities carry the true entity's row) and the scales make the comparison integers dense
    (scale 1.0: scores of order 10, quantum 1e-5) or sparse.  Shapes outside the prefilter kernel (k_int = 200 real,
    <= 128 query rows) silently take the exact kernel and must agree as well."""
    from emgraph_amd.evaluation import rank_triples_device
    d = dev()
    E, R, ki = make_tables(model, k, n_ent, 5, seed=k + n_ent, scale=scale)
    rs = np.random.RandomState(n_ent)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 5, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    for j in range(0, nq, 4):
        E[rs.randint(0, n_ent, 2)] = E[T[j, 2]]
        E[rs.randint(0, n_ent, 1)] = E[T[j, 0]]
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 5000), rs.randint(0, 5, 5000), rs.randint(0, n_ent, 5000)], 1)]).astype(np.int32)
    sc = scale_of(model, k)
    Et, Rt = cu(E), cu(R)
    used = 0
    for side in ("s,o", "s+o", "o"):
        for strategy in ("worst", "best", "middle"):
            for filt in (None, F):
                st = {}
                exact = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt)
                fast = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt, precision=2, stats=st)
                np.testing.assert_array_equal(fast, exact, err_msg=str((side, strategy, filt is not None)))
                used += st.get("pairs", 0) + st.get("fallback", 0)
    kint_ok = ki <= 800      # every width up to 800 runs the prefilter (padded to its next instantiation: 8 waves x 256 rows up
                             # to 400 columns, 4 waves x 128 rows above); 801 takes the exact kernel
    if kint_ok and 2 * nq > 128:
        assert used > 0        # the prefilter ran and handed candidates (at least the ties) to the exact re-scoring


@pytest.mark.parametrize("k,n_ent,nq,crowd", [(200, 9000, 300, 0.0), (200, 4096 * 3 + 77, 140, 0.0), (64, 130000, 200, 0.0),
                                               (300, 5000, 130, 0.0), (200, 12000, 200, 1e-4)])
def test_prefilter_bitmap_form_equals_the_emitting_form(monkeypatch, k, n_ent, nq, crowd):
    """the prefilter kernel's two forms on the same inputs (csrc/emg_rank_bf16.hip, MODE 3 + prefilter_compact_kernel against MODE 2,
    EMG_PRE_BITMAP=0): the decided counts are identical, every segment holds the same SET of (row, entity) pairs and the same
    count; a table whose scores crowd around the positives' overflows its segments — the bitmap form then records such a segment as
    EMPTY and raises the overflow flag (the caller redoes the tile exactly), never a count that would make the re-scoring pass read
    bitmap words as pairs.  Shapes: a last tile that is not full, a last chunk that is not full (4096 x 3 + 77 entities), several
    chunks, the wide form above 400 columns (k = 300: four waves x 128 query rows), query rows that end inside a workgroup.
    Reference behaviour behind both: perform_comparision's counts (EmbeddingModel.py:2010-2033)."""
    d = dev()
    from emgraph_amd.evaluation import ranking as RK
    E, R, ki = make_tables("ComplEx", k, n_ent, 3, seed=k + nq, scale=0.1)
    if crowd:
        E = (E[:1] + crowd * E).astype(F32)
    rs = np.random.RandomState(nq)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 3, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    Et, Rt = cu(E), cu(R)
    tabs = RK.PrefilterTables(Et, ki)
    Q, pos_int = d.eval_build_queries(MID["ComplEx"], Et, Rt, ki, 1.0, cu(T), 3)
    Qb = d.to_f16(Q, ki, ld_dst=d.prefilter_ld(ki))
    band = RK.prefilter_band(Q, Qb, ki, tabs.bounds(0, n_ent))
    n_rows = Q.shape[0]
    n_seg = d.eval_prefilter_segments(n_rows, n_ent, ki)
    cap = 2048
    out = {}
    for form in ("0", "1"):
        monkeypatch.setenv("EMG_PRE_BITMAP", form)
        pairs = torch.full((n_seg * cap,), -1, dtype=torch.int64, device="cuda")
        pcount = torch.full((n_seg + 1,), 77, dtype=torch.int32, device="cuda")      # (the call clears it)
        cnt = torch.zeros(n_rows, dtype=torch.int32, device="cuda")
        d.eval_prefilter_f16(MID["ComplEx"], Qb, pos_int, band, tabs.ent_f16, 0, ki, 1.0, cnt, pairs, pcount)
        torch.cuda.synchronize()
        out[form] = (cnt.cpu().numpy(), pairs.cpu().numpy().reshape(n_seg, cap), pcount.cpu().numpy())
    monkeypatch.delenv("EMG_PRE_BITMAP")
    (c0, p0, n0), (c1, p1, n1) = out["0"], out["1"]
    if not crowd:
        assert n0[n_seg] == 0 and n1[n_seg] == 0
        np.testing.assert_array_equal(c1, c0)
        np.testing.assert_array_equal(n1, n0)
        assert n0[:n_seg].sum() > 0
        for sgm in np.nonzero(n0[:n_seg])[0]:
            a, b = np.sort(p0[sgm, :n0[sgm]]), np.sort(p1[sgm, :n1[sgm]])
            np.testing.assert_array_equal(b, a, err_msg="segment %d" % sgm)
            rows, ents = b >> 32, b & 0xffffffff
            assert rows.max() < n_rows and ents.max() < n_ent
            assert np.all(np.diff((p1[sgm, :n1[sgm]] & 0xffffffff) >> 7) >= 0)     # tiles ascending: the order the re-scoring kernels sweep in
    else:
        assert n0[n_seg] != 0 and n1[n_seg] != 0                 # both forms say: redo this query tile exactly
        np.testing.assert_array_equal(c1, c0)                    # (the decided counts do not depend on the room for pairs)
        over = np.nonzero(n1[:n_seg] == 0)[0]
        assert over.size > 0 and n1[:n_seg].max() <= cap
        # ... and the public path returns the exact kernel's ranks on this table
        from emgraph_amd.evaluation import rank_triples_device
        monkeypatch.setenv("EMG_PREFILTER_PROBE", "0")
        st = {}
        fast = rank_triples_device(MID["ComplEx"], Et, Rt, ki, 1.0, T, "s,o", "worst", precision=2, ent_f16=tabs, stats=st)
        monkeypatch.delenv("EMG_PREFILTER_PROBE")
        np.testing.assert_array_equal(fast, rank_triples_device(MID["ComplEx"], Et, Rt, ki, 1.0, T, "s,o", "worst"))
        assert st["fallback"] >= 1


@pytest.mark.parametrize("model,k,scale", [("ComplEx", 200, 2.45e-3), ("DistMult", 200, 4e-3), ("HolE", 100, 3e-3), ("DistMult", 300, 2e-3),
                                           ("ComplEx", 200, 0.02), ("DistMult", 64, 0.05)])
def test_prefilter_proves_ties_on_tables_of_small_scores(monkeypatch, model, k, scale):
    """The reference compares int32(score * 1e5) (EmbeddingModel.py:2010-2014): on a freshly initialised table (Glorot limit 2.45e-3 at
    1M x 400) every score truncates to 0 and EVERY candidate ties with the positive.  The prefilter's plain form calls a tie undecided;
    its second form (emg_eval_prefilter_f16_ties, MODE 4 of csrc/emg_rank_bf16.hip) proves it: an accumulator inside the positive's
    integer cell by more than the band is counted into cnt_eq.  precision 2 must (a) return the exact kernel's ranks for every side /
    strategy / filter, (b) have taken the ties form without a single tile redone by the exact kernel (tables of scale 2e-3 ... 5e-2:
    all ties, or a mix of ties and decided candidates), and (c) fall back exactly as before with EMG_PREFILTER_TIES=0.
    Planted: rows scaled up so that some candidates are decided greater / smaller, duplicates of the positive's row (exact ties)."""
    from emgraph_amd.evaluation import rank_triples_device
    from emgraph_amd.evaluation import ranking as RK
    d = dev()
    n_ent, nq = 30000, 260
    E, R, ki = make_tables(model, k, n_ent, 4, seed=k, scale=scale)
    rs = np.random.RandomState(k)
    if scale >= 3e-3:                                # (the pure case — every candidate a tie — must not redo a single tile)
        big = rs.randint(0, n_ent, 300)
        E[big] *= 30.0                               # a few hundred entities whose scores reach the cell's ends or leave it
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    E[rs.randint(0, n_ent, 20)] = E[T[:20, 2]]       # exact duplicates of some true objects
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 3000), rs.randint(0, 4, 3000), rs.randint(0, n_ent, 3000)], 1)]).astype(np.int32)
    sc = scale_of(model, k)
    Et, Rt = cu(E), cu(R)
    tabs = RK.PrefilterTables(Et, ki)
    took = 0
    for side in ("s,o", "s+o", "o"):
        for strategy in ("worst", "best", "middle"):
            for filt in (None, F):
                exact = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt)
                st = {}
                fast = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt, precision=2, ent_f16=tabs, stats=st)
                np.testing.assert_array_equal(fast, exact, err_msg=str((side, strategy, filt is not None, st)))
                if st.get("prove_ties"):
                    took += 1
                    if scale < 3e-3:
                        assert st.get("fallback", 0) == 0 and st.get("pairs", 0) < 0.001 * fast.size * n_ent, st
    assert took > 0 or scale >= 3e-3, "the ties form was never taken on a table of tiny scores"
    monkeypatch.setenv("EMG_PREFILTER_TIES", "0")
    st = {}
    old = rank_triples_device(MID[model], Et, Rt, ki, sc, T, "s,o", "worst", precision=2, stats=st)
    monkeypatch.delenv("EMG_PREFILTER_TIES")
    np.testing.assert_array_equal(old, rank_triples_device(MID[model], Et, Rt, ki, sc, T, "s,o", "worst"))
    assert not st.get("prove_ties")


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "6"))))
def test_ties_prefilter_random_tables_equal_exact(seed):
    """soak of the ties-proving prefilter: tables between "every score truncates to the comparison integer 0" and "a few ties":
    scales 1e-4 ... 5e-2, a random share of rows scaled up by 3 ... 100 (scores at the cell's ends, outside it, far outside it),
    duplicates of true entities, random model / width / table size / side / strategy / filter — precision 2 (whichever form its
    probes pick) must return the ranks of precision 0"""
    from emgraph_amd.evaluation import rank_triples_device
    dev()
    rs = np.random.RandomState(21000 + seed)
    for _ in range(3):
        model = ("DistMult", "ComplEx", "HolE")[rs.randint(0, 3)]
        k = int(rs.choice([16, 50, 100, 104, 200, 300]))
        n_ent, nq = int(rs.randint(4200, 60000)), int(rs.randint(130, 500))
        scale = 10.0 ** rs.uniform(-4, -1.3)
        E, R, ki = make_tables(model, k, n_ent, 5, seed=seed * 11 + k, scale=scale)
        if rs.randint(0, 2):
            R = (R * F32(10.0 ** rs.uniform(0, 1.5))).astype(F32)                     # (Glorot relations are larger than Glorot entities)
        nbig = int(rs.choice([0, 0, 30, 1000, n_ent // 4]))
        if nbig:
            E[rs.randint(0, n_ent, nbig)] *= F32(rs.choice([3.0, 10.0, 30.0, 100.0]))
        pool = rs.randint(0, n_ent, 80)
        T = np.stack([rs.choice(pool, nq), rs.randint(0, 5, nq), rs.choice(pool, nq)], 1).astype(np.int32)
        E[rs.randint(0, n_ent, 25)] = E[rs.choice(pool, 25)]
        side = ("s,o", "s+o", "s", "o")[rs.randint(0, 4)]
        strategy = ("worst", "best", "middle")[rs.randint(0, 3)]
        filt = np.concatenate([T, np.stack([rs.randint(0, n_ent, 2000), rs.randint(0, 5, 2000), rs.randint(0, n_ent, 2000)], 1)]).astype(np.int32) if rs.randint(0, 2) else None
        sc = scale_of(model, k)
        Et, Rt = cu(E), cu(R)
        exact = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt)
        st = {}
        fast = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt, precision=2, stats=st)
        np.testing.assert_array_equal(fast, exact, err_msg=str((model, k, n_ent, nq, scale, nbig, side, strategy, filt is not None, st)))


def test_prefilter_probe_sends_an_undecidable_table_to_the_exact_kernel(monkeypatch):
    """precision 2 first runs 128 of the call's triples through the prefilter alone and reads the undecided fraction
    (ranking._prefilter_probe).  A table whose rows are all but equal (a freshly initialised model looks like this to the band)
    leaves every candidate undecided: the whole call then takes the exact kernel — ONE pass instead of a wasted prefilter pass, an
    overflowing pair buffer and the exact pass after it — and says so in stats; a table the band does decide keeps the prefilter.
    Ranks equal the exact path's either way, with the probe or without it (EMG_PREFILTER_PROBE=0: the old behaviour)."""
    from emgraph_amd.evaluation import rank_triples_device
    from emgraph_amd.evaluation import ranking as RK
    n_ent, k, nq = 40000, 200, 300
    rs = np.random.RandomState(5)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 3, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    E0, R, ki = make_tables("ComplEx", k, n_ent, 3, seed=11, scale=0.1)
    crowd = (E0[:1] + 1e-4 * E0).astype(np.float32)     # every row = one row + a ten-thousandth of its own: scores crowd together
    for name, E, undecidable in (("crowded", crowd, True), ("spread", E0, False)):
        Et, Rt = cu(E), cu(R)
        exact = rank_triples_device(MID["ComplEx"], Et, Rt, ki, 1.0, T, "s,o", "worst")
        tabs = RK.PrefilterTables(Et, ki)
        st = {}
        fast = rank_triples_device(MID["ComplEx"], Et, Rt, ki, 1.0, T, "s,o", "worst", precision=2, ent_f16=tabs, stats=st)
        np.testing.assert_array_equal(fast, exact, err_msg=name)
        assert (st["probe_undecided"] > RK._PROBE_MAX_UNDECIDED) == undecidable, (name, st)
        if undecidable:
            assert st["fallback"] == 1 and st.get("pairs", 0) == 0 and st["count_launches"] == 1, st
        else:
            assert st.get("fallback", 0) == 0 and st["pairs"] > 0, st
        assert len(tabs.undecided) == 1                       # remembered on the run's tables: the next call does not probe again
        st2 = {}
        again = rank_triples_device(MID["ComplEx"], Et, Rt, ki, 1.0, T, "s,o", "worst", precision=2, ent_f16=tabs, stats=st2)
        np.testing.assert_array_equal(again, exact)
        assert st2["probe_undecided"] == st["probe_undecided"]
        monkeypatch.setenv("EMG_PREFILTER_PROBE", "0")
        st3 = {}
        old = rank_triples_device(MID["ComplEx"], Et, Rt, ki, 1.0, T, "s,o", "worst", precision=2, stats=st3)
        monkeypatch.delenv("EMG_PREFILTER_PROBE")
        np.testing.assert_array_equal(old, exact)
        assert "probe_undecided" not in st3


@pytest.mark.parametrize("k,n_ent,nq,scale", [(200, 30000, 300, 0.1), (100, 9000, 200, 1.0), (50, 20000, 150, 0.002),
                                              (37, 5000, 140, 0.3), (16, 40000, 130, 0.05), (200, 5000, 40, 0.1), (3, 9, 6, 0.5),
                                              (1, 130, 3, 0.5), (520, 1500, 70, 0.05)])
def test_sad_prefilter_ranks_equal_exact_ranks_transe_l1(k, n_ent, nq, scale):
    """TransE-L1, precision=2 (v_sad_u16 sums over 16-bit fixed-point images bound every score from both sides; the
    undecided candidates are re-scored with the canonical f32 chain) == precision=0 for every side, strategy and filter
    setting.  Planted: exact ties (copies of the true entity's row), near ties (copies perturbed in the last bits and
    by about one comparison quantum, 1e-5), one entity far outside the bulk (stretches the fixed-point range), a tiny
    scale (scores below the quantum: every comparison integer ties at 0), an odd width (unaligned rows) and widths that
    are not a multiple of the image's 16-column tile."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import SadTables, rank_triples_device
    d = dev()
    E, R, ki = make_tables("TransE_L1", k, n_ent, 5, seed=k + n_ent, scale=scale)
    rs = np.random.RandomState(n_ent + k)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 5, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    for j in range(0, nq, 4):
        E[rs.randint(0, n_ent, 2)] = E[T[j, 2]]
        E[rs.randint(0, n_ent, 1)] = E[T[j, 0]]
        near = E[T[j, 2]].copy()
        near[rs.randint(0, ki)] += F32(1e-5) * F32(rs.choice([-1.5, -1.0, -0.5, 0.5, 1.0, 1.5]))   # about one quantum of the comparison
        E[rs.randint(0, n_ent)] = near
        E[rs.randint(0, n_ent)] = np.nextafter(E[T[j, 0]], F32(np.inf))                               # last-bit neighbours
    far = rs.randint(0, n_ent)
    if far not in T[:, [0, 2]]:
        E[far] *= F32(6.0)
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 5000), rs.randint(0, 5, 5000), rs.randint(0, n_ent, 5000)], 1)]).astype(np.int32)
    Et, Rt = cu(E), cu(R)
    assert rank_triples_device(L.TRANSE_L1, Et, Rt, ki, 1.0, T[:0], "s,o", "worst", precision=2).shape == (0, 2)
    tabs = SadTables(Et, Rt, ki)
    used = 0
    for side in ("s,o", "s+o", "o", "s"):
        for strategy in ("worst", "best", "middle"):
            for filt in (None, F):
                st = {}
                exact = rank_triples_device(L.TRANSE_L1, Et, Rt, ki, 1.0, T, side, strategy, filter_triples=filt)
                fast = rank_triples_device(L.TRANSE_L1, Et, Rt, ki, 1.0, T, side, strategy, filter_triples=filt, precision=2,
                                           stats=st, ent_f16=tabs if side != "o" else None, query_chunk=200 if side == "s" else 4096)
                np.testing.assert_array_equal(fast, exact, err_msg=str((side, strategy, filt is not None, st)))
                used += st.get("pairs", 0) + st.get("fallback", 0)
    assert used > 0        # the prefilter ran and handed candidates (at least the ties) to the exact re-scoring


@pytest.mark.parametrize("k,n_ent,nq,scale,huge", [(200, 30000, 300, 0.1, False), (126, 9000, 200, 1.0, False), (398, 6000, 150, 0.05, False),
                                                   (200, 20000, 150, 0.0005, False), (100, 5000, 140, 0.3, False), (200, 5000, 40, 0.1, False),
                                                   (200, 8000, 160, 0.1, True), (150, 7000, 150, 0.1, False), (50, 30000, 200, 0.2, False),
                                                   (75, 9000, 140, 0.1, False), (351, 3000, 130, 0.1, False), (7, 20000, 150, 0.5, False),
                                                   (500, 5000, 150, 0.1, False), (798, 3000, 140, 0.05, False)])   # (4-wave prefilter form)
def test_l2_prefilter_ranks_equal_exact_ranks_transe_l2(k, n_ent, nq, scale, huge):
    """TransE-L2, precision=2 (||q-e||^2 as a contraction over k+2 coordinates through the half-precision MFMA prefilter,
    thresholds derived for the squared distance, undecided candidates re-scored with the canonical f32 chain) ==
    precision=0 for every side, strategy and filter setting.  Planted: exact ties, last-bit and one-quantum neighbours,
    identical query/entity rows (distance exactly 0), a tiny scale (all comparison integers tie -> everything undecided
    -> overflow -> exact kernel), <= 128 query rows (exact kernel) and an entity whose
    squared norm does not fit a half (the clamped split shows up in the band, never in the ranks)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import L2Tables, rank_triples_device
    dev()
    E, R, ki = make_tables("TransE_L2", k, n_ent, 5, seed=k + n_ent, scale=scale)
    rs = np.random.RandomState(n_ent + k)
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 5, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    for j in range(0, nq, 4):
        E[rs.randint(0, n_ent, 2)] = E[T[j, 2]]
        E[rs.randint(0, n_ent, 1)] = E[T[j, 0]]
        near = E[T[j, 2]].copy()
        near[rs.randint(0, ki)] += F32(1e-5) * F32(rs.choice([-1.5, -1.0, -0.5, 0.5, 1.0, 1.5]))
        E[rs.randint(0, n_ent)] = near
        E[rs.randint(0, n_ent)] = np.nextafter(E[T[j, 0]], F32(np.inf))
    R[4] = 0                                            # relation 4: q = s, so the subject itself sits at distance exactly 0
    if huge:
        far = rs.randint(0, n_ent)
        if far not in T[:, [0, 2]]:
            E[far] = F32(30.0)                          # |e|^2 = 180000 > the largest half
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 5000), rs.randint(0, 5, 5000), rs.randint(0, n_ent, 5000)], 1)]).astype(np.int32)
    Et, Rt = cu(E), cu(R)
    tabs = L2Tables(Et, ki)
    used = 0
    for side in ("s,o", "s+o", "o", "s"):
        for strategy in ("worst", "best", "middle"):
            for filt in (None, F):
                st = {}
                exact = rank_triples_device(L.TRANSE_L2, Et, Rt, ki, 1.0, T, side, strategy, filter_triples=filt)
                fast = rank_triples_device(L.TRANSE_L2, Et, Rt, ki, 1.0, T, side, strategy, filter_triples=filt, precision=2,
                                           stats=st, ent_f16=tabs if side != "o" else None, query_chunk=200 if side == "s" else 4096)
                np.testing.assert_array_equal(fast, exact, err_msg=str((side, strategy, filt is not None, st)))
                used += st.get("pairs", 0) + st.get("fallback", 0)
    if k + 2 <= 800 and nq > 128:
        assert used > 0        # the prefilter ran


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "6"))))
def test_contraction_prefilter_random_shapes_equal_exact(seed, monkeypatch):
    """soak of the half-precision MFMA prefilter in its three forms — the bitmap form in the 64-rows-per-wave kernel (400 / 208
    columns), the bitmap form in the register-stationary kernel (every other width, EMG_PRE_V4=0 everywhere) and the emitting form
    (EMG_PRE_BITMAP=0) — on random shapes: DistMult / ComplEx / HolE, widths 3 … 400 (6 … 800 columns), tables of 1500 … 70 000
    rows (one chunk, several chunks, a last tile that is not full), 130 … 700 test triples (query rows that end inside a
    workgroup), scales over four decades, heavy-tailed tables, few distinct query entities and copies of them elsewhere in the
    table (exact ties): precision 2 must return the ranks of precision 0 for a random side / strategy / filter."""
    from emgraph_amd.evaluation import rank_triples_device
    dev()
    rs = np.random.RandomState(5000 + seed)
    for _ in range(4):
        model = ("DistMult", "ComplEx", "HolE")[rs.randint(0, 3)]
        k = int(rs.choice([200, 104, 100, 52])) if rs.randint(0, 3) == 0 else int(rs.randint(3, 401))   # (a third of the draws: the v4 widths)
        if model == "DistMult" and rs.randint(0, 2):
            k = min(2 * k, 800)
        n_ent, nq = int(rs.randint(1500, 70000)), int(rs.randint(130, 700))
        scale = 10.0 ** rs.uniform(-3, 0.5)
        E, R, ki = make_tables(model, k, n_ent, 6, seed=seed * 7 + k, scale=scale)
        if rs.randint(0, 2):
            E = (rs.standard_t(3, E.shape).astype(F32) * F32(scale)).astype(F32)
        pool = rs.randint(0, n_ent, 60)
        T = np.stack([rs.choice(pool, nq), rs.randint(0, 6, nq), rs.choice(pool, nq)], 1).astype(np.int32)
        E[rs.randint(0, n_ent, 30)] = E[rs.choice(pool, 30)]
        side = ("s,o", "s+o", "s", "o")[rs.randint(0, 4)]
        strategy = ("worst", "best", "middle")[rs.randint(0, 3)]
        filt = np.concatenate([T, np.stack([rs.randint(0, n_ent, 3000), rs.randint(0, 6, 3000), rs.randint(0, n_ent, 3000)], 1)]).astype(np.int32) if rs.randint(0, 2) else None
        sc = scale_of(model, k)
        Et, Rt = cu(E), cu(R)
        exact = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt)
        for env in ({}, {"EMG_PRE_V4": "0"}, {"EMG_PRE_BITMAP": "0"}):
            for kk, vv in env.items():
                monkeypatch.setenv(kk, vv)
            fast = rank_triples_device(MID[model], Et, Rt, ki, sc, T, side, strategy, filter_triples=filt, precision=2)
            for kk in env:
                monkeypatch.delenv(kk)
            np.testing.assert_array_equal(fast, exact, err_msg=str((model, k, n_ent, nq, scale, side, strategy, filt is not None, env)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "6"))))
def test_transe_prefilters_random_shapes_equal_exact(seed):
    """soak of both TransE prefilters: random widths, table scales over four decades, heavy-tailed tables, relation scales
    far from the entity scale, query sets with many repeated entities — precision 2 must return the ranks of precision 0"""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import rank_triples_device
    dev()
    rs = np.random.RandomState(1000 + seed)
    for _ in range(5):
        l2 = bool(rs.randint(0, 2))
        k = int(rs.randint(1, 420)) if l2 else int(rs.randint(1, 261))
        n_ent, nq = int(rs.randint(1500, 20000)), int(rs.randint(130, 300))
        scale = 10.0 ** rs.uniform(-3, 1)
        E = (rs.standard_t(3, (n_ent, k)) if rs.randint(0, 2) else rs.randn(n_ent, k)).astype(F32) * F32(scale)
        R = (rs.randn(6, k) * scale * 10.0 ** rs.uniform(-2, 1)).astype(F32)
        pool = rs.randint(0, n_ent, 40)                                   # few distinct entities: many equal scores
        T = np.stack([rs.choice(pool, nq), rs.randint(0, 6, nq), rs.choice(pool, nq)], 1).astype(np.int32)
        E[rs.randint(0, n_ent, 30)] = E[rs.choice(pool, 30)]             # copies of query entities elsewhere in the table
        mid = L.TRANSE_L2 if l2 else L.TRANSE_L1
        Et, Rt = cu(E), cu(R)
        side, strategy = str(rs.choice(["s,o", "s+o", "s", "o"])), str(rs.choice(["worst", "best", "middle"]))
        filt = T if rs.randint(0, 2) else None
        exact = rank_triples_device(mid, Et, Rt, k, 1.0, T, side, strategy, filter_triples=filt)
        fast = rank_triples_device(mid, Et, Rt, k, 1.0, T, side, strategy, filter_triples=filt, precision=2)
        np.testing.assert_array_equal(fast, exact, err_msg=str((seed, l2, k, n_ent, nq, scale, side, strategy)))


def test_sad_images_and_thresholds_bound_the_exact_chain():
    """the pieces of the fixed-point prefilter against float64 math: the image is rint((x + R) / delta) over the range
    emg_eval_sad_range reports, and for every (query, entity) pair the exact comparison integer lies on the side the
    thresholds promise: S < lo => int(score 1e5) > pos_int, S > hi => int(score 1e5) < pos_int"""
    from emgraph_amd import _lib as L
    d = dev()
    rs = np.random.RandomState(5)
    n_ent, nq, k = 3000, 96, 72
    E = (rs.randn(n_ent, k) * 0.2).astype(F32)
    R = (rs.randn(4, k) * 0.2).astype(F32)
    Et, Rt = cu(E), cu(R)
    rng = d.eval_sad_range(Et, Rt, k)
    r = rng.cpu().numpy()
    assert r[0] == np.abs(E).max() and r[1] == np.abs(R).max()
    Rh = (r[0] + r[1]) * (1.0 + 1e-6)
    delta = 2.0 * Rh / 65535.0
    img = d.eval_sad_quantize(Et, k, rng).cpu().numpy().view(np.uint16)
    assert img.shape[1] == d.eval_sad_ld(k) == 80 and not img[:, k:].any()
    np.testing.assert_array_equal(img[:, :k], np.rint((E.astype(np.float64) + Rh) * (65535.0 / (2.0 * Rh))).astype(np.uint16))
    T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 4, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
    Q, pos_int = d.eval_build_queries(L.TRANSE_L1, Et, Rt, k, 1.0, cu(T), L.EVAL_O)
    thr = d.eval_sad_thresholds(pos_int, k, rng).cpu().numpy().view(np.uint32).astype(np.int64)
    qimg = d.eval_sad_quantize(Q, k, rng).cpu().numpy().view(np.uint16).astype(np.int64)
    S = np.abs(qimg[:, None, :] - img[None, :, :].astype(np.int64)).sum(-1)                      # [nq, n_ent]
    Qh = Q.cpu().numpy()
    acc = np.zeros((nq, n_ent), F32)
    for c in range(k):                                                                           # the canonical f32 chain
        acc = acc + np.abs(Qh[:, c:c + 1] - E[None, :, c])
    ci = (-(acc) * F32(100000.0)).astype(np.int32)                                               # truncation toward zero
    p = pos_int.cpu().numpy()[:, None]
    assert np.all(np.abs(acc.astype(np.float64) - delta * S) <= k * delta * 1.001 + 1e-5 * acc)
    assert np.all(ci[S < thr[0][:, None]] > np.broadcast_to(p, S.shape)[S < thr[0][:, None]])
    assert np.all(ci[S > thr[1][:, None]] < np.broadcast_to(p, S.shape)[S > thr[1][:, None]])
    undecided = ((S >= thr[0][:, None]) & (S <= thr[1][:, None])).mean()
    assert undecided < 0.05, undecided


def test_filter_index_gpu_sort_equals_numpy_sort():
    """FilterIndex built through the library's radix sort (large filter sets on a GPU host) == the numpy merge-sort build:
    same stable order, same CSR for every side"""
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import ranking as RK
    dev()
    rs = np.random.RandomState(3)
    n_ent, n_rel, n = 50000, 37, 250000
    F = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1).astype(np.int64)
    key = F[:, 0] * n_rel + F[:, 1]
    np.testing.assert_array_equal(RK._stable_argsort(key), np.argsort(key, kind="stable"))
    fast = RK.FilterIndex(F)
    slow = RK.FilterIndex(F[:90000])      # below the size threshold: numpy path
    ref = RK.FilterIndex.__new__(RK.FilterIndex)
    ref.n_rel, ref.max_entity, ref._sides = fast.n_rel, fast.max_entity, {}
    for name, kcol, vcol in (("obj", 0, 2), ("sub", 2, 0)):
        k2 = F[:, kcol] * fast.n_rel + F[:, 1]
        o2 = np.argsort(k2, kind="stable")
        ref._sides[name] = (k2[o2], F[o2, vcol])
    T = F[:500].astype(np.int32)
    for sm in (L.EVAL_S, L.EVAL_O, L.EVAL_SPO, L.EVAL_S_O):
        a, b = fast.csr(T, sm, n_ent), ref.csr(T, sm, n_ent)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    assert slow._side("obj")[0].shape[0] == 90000


def test_device_initialisers_distribution_and_determinism():
    """emg_init_table: U[a, b) and N(mean, std) moments, bounds, determinism in (seed, stream), independence of the two
    tables, untouched row padding (initializers/*.py: the reference's TF draws are unpinned, the distribution is the contract)"""
    from emgraph_amd.training import alloc_table
    d = dev()
    rows, k = 20000, 50            # 50 floats per row: stride padded to 52
    t = alloc_table(rows, k, torch.device("cuda"), fill=7.0)
    base = t.as_strided((rows, 52), (52, 1))
    base.fill_(7.0)
    d.init_table(t, k, "uniform", -0.25, 0.75, 11, 1)
    u = t.cpu().numpy().astype(np.float64)
    assert u.min() >= -0.25 and u.max() < 0.75
    assert abs(u.mean() - 0.25) < 2e-3 and abs(u.var() - 1.0 / 12) < 2e-3
    assert float(base[:, 50:].min()) == 7.0 and float(base[:, 50:].max()) == 7.0      # padding untouched
    t2 = alloc_table(rows, k, torch.device("cuda"))
    d.init_table(t2, k, "uniform", -0.25, 0.75, 11, 1)
    assert torch.equal(t, t2)                                                          # same (seed, stream) -> same table
    d.init_table(t2, k, "uniform", -0.25, 0.75, 11, 2)
    assert not torch.equal(t, t2) and abs(np.corrcoef(u.ravel(), t2.cpu().numpy().ravel())[0, 1]) < 5e-3
    d.init_table(t2, k, "normal", 0.5, 0.1, 3, 1)
    g = t2.cpu().numpy().astype(np.float64)
    assert abs(g.mean() - 0.5) < 1e-3 and abs(g.std() - 0.1) < 1e-3
    z = (g - 0.5) / 0.1
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3.0) < 5e-2          # skewness, kurtosis of a normal


# ------------------------------------------------------------------------------------------------
# the counting grouping (histogram -> look-back scan -> scatter -> ordering) and the descriptor-driven apply at the edges
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["one_row", "all_one_destination", "sparse_big_table", "many_tiles_dense", "ids_out_of_range",
                                  "lengths_around_thresholds", "single_contribution"])
def test_counting_grouping_edge_shapes(case):
    """emg_group_dest + emg_apply_grouped through the counting backend, bit for bit against the stable order / the sequential
    sums: a one-row table, every contribution on one destination (one segment = the whole batch: block tasks), a table of
    600 k rows hit by 3 000 contributions (147 scan tiles, almost all empty: the look-back runs over several rounds of 64
    predecessors), a dense batch over 70 tiles, destination ids outside the table (dropped), segment lengths on both
    sides of the singleton / 32-row / 64-row thresholds, and a single contribution."""
    d = dev()
    from emgraph_amd import _lib as L
    rs = np.random.RandomState(sum(map(ord, case)))
    k = 72   # 18 chunks of 16 bytes: the descriptor-driven kernel (more than 16 chunks, aligned rows)
    drop = None
    if case == "one_row":
        n_rows, dest = 1, np.zeros(500, np.int32)
    elif case == "all_one_destination":
        n_rows, dest = 50, np.full(3000, 17, np.int32)
    elif case == "sparse_big_table":
        n_rows = 600_000
        dest = rs.randint(0, n_rows, 3000).astype(np.int32)
        dest[:40] = dest[40:80]            # a few pairs
    elif case == "many_tiles_dense":
        n_rows = 70 * 4096 + 5
        dest = rs.randint(0, n_rows, 400_000).astype(np.int32)
    elif case == "ids_out_of_range":
        n_rows = 300
        dest = rs.randint(0, n_rows, 4000).astype(np.int32)
        drop = rs.choice(4000, 200, replace=False)
        dest[drop[:100]] = n_rows + rs.randint(0, 50, 100)
        dest[drop[100:]] = -1 - rs.randint(0, 50, 100)
    elif case == "lengths_around_thresholds":
        n_rows = 400
        lens = [1, 2, 3, 31, 32, 33, 34, 63, 64, 65, 127, 128, 129, 1, 1, 2]
        dest = np.concatenate([np.full(n, 10 + 3 * i, np.int32) for i, n in enumerate(lens)])
        rs.shuffle(dest)
    else:
        n_rows, dest = 1000, np.array([123], np.int32)
    n_c = len(dest)
    # 1. the grouping itself: keys ascending, values = the stable order, singleton flags
    ws = torch.zeros(d.apply_workspace_bytes(n_c, n_rows, k), dtype=torch.uint8, device="cuda")
    flags = torch.zeros(n_c, dtype=torch.uint8, device="cuda")
    d.group_dest(cu(dest), n_c, n_rows, ws, flags)
    valid = (dest >= 0) & (dest < n_rows)
    order = np.argsort(np.where(valid, dest, np.iinfo(np.int32).max), kind="stable")[:int(valid.sum())]
    keys, vals = d.apply_workspace_views(ws, n_c)
    nv = int(valid.sum())
    np.testing.assert_array_equal(keys.cpu().numpy()[:nv], dest[order])
    np.testing.assert_array_equal(vals.cpu().numpy()[:nv], order.astype(np.int32))
    cnt = np.bincount(dest[valid], minlength=n_rows)
    np.testing.assert_array_equal(flags.cpu().numpy(), (valid & (cnt[np.where(valid, dest, 0)] == 1)).astype(np.uint8))
    # 2. the apply from that grouping (twice on the same workspace: the control region must come back clean)
    W = rs.randn(n_rows, k).astype(F32)
    contrib = rs.randn(n_c, k).astype(F32)
    lr = F32(0.05)
    Wt = cu(W)
    for rep in range(2):
        d.apply_rows(L.OPT_SGD, Wt, k, None, None, None, 1 + rep, cu(contrib), cu(dest), n_c, (float(lr), 0, 0, 0, 0, 0), ws)
    exp = W.copy()
    bounds = np.flatnonzero(np.diff(dest[order])) + 1
    for rep in range(2):
        for seg in (np.split(order, bounds) if nv else []):
            if len(seg) <= 64:
                g = np.zeros(k, F32)
                for i in seg:
                    g = g + contrib[i]
            else:   # 64-row blocks left to right, then the block sums left to right (the long-segment tree)
                parts = []
                for b0 in range(0, len(seg), 64):
                    p = np.zeros(k, F32)
                    for i in seg[b0:b0 + 64]:
                        p = p + contrib[i]
                    parts.append(p)
                g = np.zeros(k, F32)
                for p in parts:
                    g = g + p
            exp[dest[seg[0]]] = exp[dest[seg[0]]] - lr * g
    np.testing.assert_array_equal(Wt.cpu().numpy(), exp)
