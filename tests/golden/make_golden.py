#!/usr/bin/env python3
"""Generate golden input/output vectors by EXECUTING THE REFERENCE'S OWN hot-path functions.

Runs only in the build container (needs /root/reference).  TensorFlow is absent there, so the
reference is imported with tests/golden/tf_shim ahead of it on sys.path: a numpy stand-in for
the ~40 TF leaf ops its pure functions call (see tf_shim/tensorflow/__init__.py).  The control
flow that executes is the reference's; `tf.random.uniform` returns INJECTED draws.

Usage:  python tests/golden/make_golden.py         (writes tests/golden/*.npz)

Outputs are data only (inputs + expected outputs); nothing of the reference's source travels.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("EMGRAPH_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "tf_shim"))

import numpy as np  # noqa: E402
import tensorflow as tf  # noqa: E402  (the shim)

assert tf.__version__.endswith("numpy-shim")

import emgraph  # noqa: E402,F401
from emgraph.evaluation.protocol import (  # noqa: E402
    create_mappings, generate_corruptions_for_eval, generate_corruptions_for_fit, to_idx)
from emgraph.evaluation.metrics import hits_at_n_score, mr_score, mrr_score, rank_score  # noqa: E402
from emgraph.losses._loss_constants import LOSS_REGISTRY  # noqa: E402
from emgraph.models import ComplEx, DistMult, HolE, TransE  # noqa: E402
from emgraph.regularizers._regularizer_constants import REGULARIZER_REGISTRY  # noqa: E402

F32 = np.float32


def gen_scores():
    out = {}
    cases = []
    for k in (3, 10, 100, 200):
        for n in (1, 7):
            cases.append((k, n))
    out["cases"] = np.array(cases, dtype=np.int64)
    for ci, (k, n) in enumerate(cases):
        rs = np.random.RandomState(1000 + ci)
        scale = 1.0 if k <= 10 else 0.1
        for model_name, cls, kint, kw in (
            ("TransE_L1", TransE, k, {"embedding_model_params": {"norm": 1}}),
            ("TransE_L2", TransE, k, {"embedding_model_params": {"norm": 2}}),
            ("DistMult", DistMult, k, {}),
            ("ComplEx", ComplEx, 2 * k, {}),
            ("HolE", HolE, 2 * k, {}),
        ):
            es, ep, eo = [(rs.randn(n, kint) * scale).astype(F32) for _ in range(3)]
            m = cls(k=k, **kw)
            y = np.asarray(m._fn(es, ep, eo), dtype=F32)
            tag = "%s_c%d" % (model_name, ci)
            out[tag + "_es"], out[tag + "_ep"], out[tag + "_eo"], out[tag + "_y"] = es, ep, eo, y
    return out


def gen_scores_transe_p():
    """TransE._fn with orders of the norm other than 1 / 2 (TransE.py:208-216 hands `norm` to tf.norm as `ord`): the path
    the product serves through its generic kernels (EMG_TRANSE_P), training included since round 4"""
    out = {}
    cases = []
    for k in (3, 10, 100, 200):
        for n in (1, 7):
            for p in (3.0, 1.5, 4.0, float("inf")):
                cases.append((k, n, p))
    out["cases"] = np.array(cases, dtype=np.float64)
    for ci, (k, n, p) in enumerate(cases):
        rs = np.random.RandomState(7000 + ci)
        scale = 1.0 if k <= 10 else 0.1
        es, ep, eo = [(rs.randn(n, k) * scale).astype(F32) for _ in range(3)]
        m = TransE(k=k, embedding_model_params={"norm": p})
        y = np.asarray(m._fn(es, ep, eo), dtype=F32)
        tag = "c%d" % ci
        out[tag + "_es"], out[tag + "_ep"], out[tag + "_eo"], out[tag + "_y"] = es, ep, eo, y
    return out


def gen_losses():
    out = {}
    cases = []
    ci = 0
    for eta in (1, 2, 20):
        for B in (1, 5):
            rs = np.random.RandomState(2000 + ci)
            pos = (rs.randn(B) * 3).astype(F32)
            neg = (rs.randn(B * eta) * 3).astype(F32)
            if B == 5:  # exercise the +-75 clip and the margin kink
                pos[0], neg[0] = 90.0, -80.0
                neg[-1] = 76.0
            out["c%d_pos" % ci], out["c%d_neg" % ci] = pos, neg
            for name in ("pairwise", "nll", "absolute_margin", "self_adversarial", "multiclass_nll"):
                loss_obj = LOSS_REGISTRY[name](eta, {})
                same = LOSS_REGISTRY[name].class_params["require_same_size_pos_neg"]
                pos_in = np.tile(pos, eta) if same else pos  # EmbeddingModel.py:724-729
                val = loss_obj.apply(pos_in, neg)
                out["c%d_%s" % (ci, name)] = np.asarray(val, dtype=F32)
                out["same_size_" + name] = np.array(int(same))
            # non-default hyper-parameters
            out["c%d_pairwise_m2.5" % ci] = np.asarray(
                LOSS_REGISTRY["pairwise"](eta, {"margin": 2.5}).apply(np.tile(pos, eta), neg), dtype=F32)
            out["c%d_self_adversarial_m1_a2" % ci] = np.asarray(
                LOSS_REGISTRY["self_adversarial"](eta, {"margin": 1.0, "alpha": 2.0}).apply(pos, neg), dtype=F32)
            cases.append((eta, B))
            ci += 1
    out["cases"] = np.array(cases, dtype=np.int64)
    # the survey's smoke case
    pos = np.array([0.5, -1.0], F32)
    neg = np.array([0.2, 0.3, -2.0, 1.0], F32)
    for name in ("pairwise", "nll", "absolute_margin", "self_adversarial", "multiclass_nll"):
        same = LOSS_REGISTRY[name].class_params["require_same_size_pos_neg"]
        out["smoke_" + name] = np.asarray(
            LOSS_REGISTRY[name](2, {}).apply(np.tile(pos, 2) if same else pos, neg), dtype=F32)
    return out


def gen_corruptions():
    out = {}
    X = np.array([["a", "x", "b"], ["c", "x", "d"], ["e", "x", "f"], ["b", "y", "h"], ["a", "y", "l"]])
    rel_to_idx, ent_to_idx = create_mappings(X)
    Xi = to_idx(X, ent_to_idx=ent_to_idx, rel_to_idx=rel_to_idx)
    out["toy_X_idx"] = Xi.astype(np.int32)
    out["toy_ent_labels"] = np.array(sorted(ent_to_idx, key=ent_to_idx.get))
    out["toy_rel_labels"] = np.array(sorted(rel_to_idx, key=rel_to_idx.get))
    # --- eval corruptions (protocol.py:448-528)
    all_ent = np.array(list(ent_to_idx.values()), dtype=np.int64)
    x = np.array([Xi[0]], dtype=np.int64)
    for side in ("s,o", "s+o", "s", "o"):
        out["eval_" + side] = np.asarray(generate_corruptions_for_eval(x, all_ent, side))
    sub = np.array([5, 2, 7], dtype=np.int64)
    out["eval_subset_ents"] = sub
    for side in ("s,o", "s", "o"):
        out["eval_subset_" + side] = np.asarray(generate_corruptions_for_eval(np.array([Xi[3]], dtype=np.int64), sub, side))
    # --- fit corruptions (protocol.py:531-659) under injected draws
    cases = []
    ci = 0
    rs = np.random.RandomState(3000)
    Xbig = np.stack([rs.randint(0, 50, 40), rs.randint(0, 4, 40), rs.randint(0, 50, 40)], 1).astype(np.int32)
    out["fit_Xbig"] = Xbig
    ent_list = np.array([3, 9, 11, 40, 41, 42, 7], dtype=np.int32)
    out["fit_entities_list"] = ent_list
    for Xname, Xarr in (("toy", Xi.astype(np.int32)), ("big", Xbig)):
        B = Xarr.shape[0]
        for eta in (1, 3):
            for side in ("s", "o", "s+o", "s,o"):
                for mode in ("size", "list", "batch"):
                    n = B * eta
                    if mode == "size":
                        esz, elist, nchoice = (8 if Xname == "toy" else 50), None, (8 if Xname == "toy" else 50)
                    elif mode == "list":
                        if Xname == "toy":
                            continue
                        esz, elist, nchoice = 0, ent_list, len(ent_list)
                    else:
                        esz, elist = 0, None
                        nchoice = len(np.unique(np.concatenate([Xarr[:, 0], Xarr[:, 2]])))
                    mask = rs.randint(0, 2, n).astype(np.int32)
                    repl = rs.randint(0, nchoice, n).astype(np.int32)
                    if side in ("s+o", "s,o"):
                        tf.random.inject(mask, repl)  # mask draw FIRST (protocol.py:600-619)
                    else:
                        tf.random.inject(repl)
                    y = generate_corruptions_for_fit(Xarr, entities_list=elist, eta=eta, corrupt_side=side,
                                                     entities_size=esz, rnd=0)
                    assert not tf.random._INJECTED
                    tag = "fit_c%d" % ci
                    out[tag + "_mask"], out[tag + "_repl"], out[tag + "_out"] = mask, repl, np.asarray(y, np.int32)
                    cases.append((Xname, eta, side, mode))
                    ci += 1
    out["fit_cases"] = np.array(cases)
    return out


def gen_misc():
    out = {}
    cls = REGULARIZER_REGISTRY["LP"]
    rs = np.random.RandomState(4000)
    w1 = rs.randn(6, 5).astype(F32)
    w2 = rs.randn(3, 5).astype(F32)
    out["lp_w1"], out["lp_w2"] = w1, w2
    for p in (1, 2, 3):
        out["lp_p%d_scalar" % p] = np.asarray(cls({"lambda": 0.01, "p": p}).apply([w1, w2]), dtype=F32)
        out["lp_p%d_list" % p] = np.asarray(cls({"lambda": [0.5, 2.0], "p": p}).apply([w1, w2]), dtype=F32)
    out["lp_default"] = np.asarray(cls({}).apply([w1, w2]), dtype=F32)
    ranks = np.array([1, 12, 6, 2, 1, 40, 3])
    out["metric_ranks"] = ranks
    out["metric_mrr"] = np.array(mrr_score(ranks))
    out["metric_mr"] = np.array(mr_score(ranks))
    for n in (1, 3, 10):
        out["metric_hits%d" % n] = np.array(hits_at_n_score(ranks, n))
    ranks2 = np.array([[1, 2], [3, 1], [10, 20]])
    out["metric_ranks2"] = ranks2
    out["metric2_mrr"] = np.array(mrr_score(ranks2))
    out["metric2_mr"] = np.array(mr_score(ranks2))
    out["metric2_hits1"] = np.array(hits_at_n_score(ranks2, 1))
    out["rank_score"] = np.array(rank_score(np.array([0, 0, 1, 0]), np.array([0.434, 0.65, 0.21, 0.84])))
    return out


def gen_ranks():
    """The rank path, executed from the reference:
      * EmbeddingModel.perform_comparision (EmbeddingModel.py:1989-2033) — the method is called on a model object
        with eval_config['ranking_strategy'] set, for all three strategies, on score vectors with planted ties,
        zeros, negative scores and values inside one comparison quantum (int32(score * 1e5) truncates toward 0);
      * SQLiteAdapter.get_participating_entities (datasets/sqlite_adapter.py:449-508) — a real sqlite3 database
        built by the adapter's own set_data path (numpy_adapter.py:229-247: use_mappings + set_data(.., 'filter',
        mapped_status)), queried per test triple;
      * and, assembled from those two plus the reference's own _fn and generate_corruptions_for_eval, the filtered
        ranks of a small random model following EmbeddingModel.py:1856-1986 line by line (that method itself needs
        tf.data and cannot run under the shim; every numeric piece below is the reference's)."""
    from emgraph.datasets import SQLiteAdapter
    out = {}
    m = TransE(k=3)
    rs = np.random.RandomState(5000)
    cases = []
    explicit = [
        # (corruption scores, positive score): the docstring example of EmbeddingModel.py:2017-2031
        (np.array([0.5, 0.5, 0.3, 0.6, 0.5, 0.5], F32), F32(0.5)),
        # one quantum is 1e-5: everything in (-1e-5, 1e-5) truncates to 0 and TIES with a zero positive
        (np.array([0.0, 4e-6, -4e-6, 9.9e-6, -9.9e-6, 1e-5, -1e-5, 1.1e-5, -1.1e-5], F32), F32(0.0)),
        (np.array([0.0, 4e-6, -4e-6, 9.9e-6, -9.9e-6, 1e-5, -1e-5, 1.1e-5, -1.1e-5], F32), F32(-3e-6)),
        # truncation toward zero of negative scores: -1.234567 -> -123456, not -123457
        (np.array([-1.234567, -1.234561, -1.234571, -1.23457, -1.23456, -1.2345], F32), F32(-1.234565)),
        (np.array([2.5, 2.5, 2.5, 2.5], F32), F32(2.5)),                       # all ties (even count)
        (np.array([2.5, 2.5, 2.5], F32), F32(2.5)),                            # all ties (odd count: ceil)
        (np.array([-7.25], F32), F32(3.0)),
        (np.array([1e3, -1e3, 123.456, 123.4561, 123.45599], F32), F32(123.456)),
    ]
    for corr, pos in explicit:
        cases.append((corr, pos))
    for n in (1, 2, 17, 64, 301):
        corr = (rs.randn(n) * 0.01).astype(F32)
        pos = F32(rs.randn() * 0.01)
        corr[rs.randint(0, n, max(1, n // 5))] = pos          # exact ties
        if n > 4:
            corr[:2] = pos + F32(3e-6)                            # same quantum or not, depending on pos
        cases.append((corr, pos))
    out["cmp_n"] = np.array(len(cases))
    for ci, (corr, pos) in enumerate(cases):
        out["cmp_c%d_corr" % ci], out["cmp_c%d_pos" % ci] = corr, np.array(pos, F32)
        for strat in ("worst", "best", "middle"):
            m.eval_config = {"ranking_strategy": strat}
            out["cmp_c%d_%s" % (ci, strat)] = np.array(int(m.perform_comparision(corr, pos)), np.int64)

    # ---- SQLite filter lookups
    n_ent, n_rel = 40, 3
    F = np.stack([rs.randint(0, n_ent, 400), rs.randint(0, n_rel, 400), rs.randint(0, n_ent, 400)], 1).astype(np.int32)
    F = np.concatenate([F, F[:25]])                               # duplicate rows: SQL DISTINCT / UNION
    T = np.concatenate([F[rs.choice(400, 12, replace=False)],     # test triples present in the filter ...
                        np.stack([rs.randint(0, n_ent, 6), rs.randint(0, n_rel, 6), rs.randint(0, n_ent, 6)], 1),
                        np.array([[39, 2, 39], [0, 0, 0]])]).astype(np.int32)   # ... absent from it, and s == o
    ent_to_idx = {i: i for i in range(n_ent)}
    rel_to_idx = {i: i for i in range(n_rel)}
    ad = SQLiteAdapter()
    ad.use_mappings(rel_to_idx, ent_to_idx)
    ad.set_data(F, "filter", True)                                # mapped_status=True (numpy_adapter.py:245-247)
    objs, subs = [], []
    for x in T:
        po, ps = ad.get_participating_entities(x)
        objs.append(np.sort(np.asarray(po).reshape(-1)).astype(np.int64))
        subs.append(np.sort(np.asarray(ps).reshape(-1)).astype(np.int64))
    ad.cleanup()
    out["flt_filter"], out["flt_test"], out["flt_n_ent"] = F, T, np.array(n_ent)
    out["flt_obj_ptr"] = np.cumsum([0] + [len(a) for a in objs]).astype(np.int64)
    out["flt_obj_idx"] = np.concatenate(objs)
    out["flt_sub_ptr"] = np.cumsum([0] + [len(a) for a in subs]).astype(np.int64)
    out["flt_sub_idx"] = np.concatenate(subs)

    # ---- filtered ranks of a small dyadic model (every fp32 op exact: any summation order gives these bits)
    k = 4
    all_ent = np.arange(n_ent, dtype=np.int64)
    for name, cls, kint in (("TransE", TransE, k), ("DistMult", DistMult, k), ("ComplEx", ComplEx, 2 * k), ("HolE", HolE, 2 * k)):
        E = (rs.randint(-4, 5, (n_ent, kint)) / 4.0).astype(F32)
        R = (rs.randint(-4, 5, (n_rel, kint)) / 4.0).astype(F32)
        mdl = cls(k=k)
        out["rk_%s_E" % name], out["rk_%s_R" % name] = E, R
        fn = lambda x: np.asarray(mdl._fn(E[x[:, 0]], R[x[:, 1]], E[x[:, 2]]), F32)   # noqa: E731 (:1861-1866 lookup + _fn)
        for strat in ("worst", "best", "middle"):
            mdl.eval_config = {"ranking_strategy": strat}
            cmpf = mdl.perform_comparision
            ranks_so, ranks_spo, ranks_raw = [], [], []
            for ti, x in enumerate(T):
                corr = np.asarray(generate_corruptions_for_eval(x[None, :].astype(np.int64), all_ent, "s,o"))  # :1856
                sc = fn(corr.astype(np.int64))
                sp = fn(x[None, :].astype(np.int64))[0]
                obj_sc, sub_sc = sc[:n_ent], sc[n_ent:]                                   # :1883-1892
                io = out["flt_obj_idx"][out["flt_obj_ptr"][ti]:out["flt_obj_ptr"][ti + 1]]
                isub = out["flt_sub_idx"][out["flt_sub_ptr"][ti]:out["flt_sub_ptr"][ti + 1]]
                f_o, f_s = int(cmpf(obj_sc[io], sp)), int(cmpf(sub_sc[isub], sp))        # :1942-1963
                r_s = int(cmpf(sub_sc, sp)) + 1 - f_s                                     # :1967-1979
                r_o = int(cmpf(obj_sc, sp)) + 1 - f_o
                ranks_so.append((r_s, r_o))
                ranks_spo.append(int(cmpf(sc, sp)) + 1 - f_s - f_o)                       # :1981-1986
                ranks_raw.append((int(cmpf(sub_sc, sp)) + 1, int(cmpf(obj_sc, sp)) + 1))
            out["rk_%s_%s_s,o" % (name, strat)] = np.array(ranks_so, np.int64)
            out["rk_%s_%s_s+o" % (name, strat)] = np.array(ranks_spo, np.int64)
            out["rk_%s_%s_raw" % (name, strat)] = np.array(ranks_raw, np.int64)
    return out


def gen_checkpoints():
    """Model files written BY THE REFERENCE's save_model (utils/model_utils.py:22-87): a fitted-looking model object
    of the reference's own class (hyper-parameters through its constructor, parameters and dictionaries set by hand,
    numpy arrays as AmpliGraph-1.x stored them) is pickled by the reference's code.  The files hold plain python /
    numpy data only (dict of str, int, bool, dict, list of ndarray): they load without the reference.
    Beside them: the scores the reference's own _fn gives for a few labelled triples."""
    from emgraph.utils import save_model
    out = {}
    labels_e = np.array(["ent_%02d" % i for i in range(12)])
    labels_r = np.array(["rel_a", "rel_b", "rel_c"])
    rs = np.random.RandomState(6000)
    X = np.stack([labels_e[rs.randint(0, 12, 9)], labels_r[rs.randint(0, 3, 9)], labels_e[rs.randint(0, 12, 9)]], 1)
    out["X"] = X
    for name, cls, kw in (("TransE", TransE, dict(k=5, eta=3, epochs=7, batches_count=2, seed=11, loss="pairwise",
                                                   loss_params={"margin": 2.0}, optimizer="adagrad",
                                                   optimizer_params={"lr": 0.1},
                                                   embedding_model_params={"norm": 2, "corrupt_sides": ["s", "o"]})),
                          ("ComplEx", ComplEx, dict(k=4, eta=2, epochs=3, batches_count=1, seed=5, regularizer="LP",
                                                     regularizer_params={"lambda": 0.01, "p": 3})),
                          ("HolE", HolE, dict(k=6))):
        m = cls(**kw)
        kint = kw["k"] * (1 if name == "TransE" else 2)
        E = (rs.randn(12, kint) * 0.4).astype(F32)
        R = (rs.randn(3, kint) * 0.4).astype(F32)
        m.is_fitted = True
        m.ent_to_idx = {str(l): i for i, l in enumerate(labels_e)}
        m.rel_to_idx = {str(l): i for i, l in enumerate(labels_r)}
        m.trained_model_params = [E, R]
        save_model(m, os.path.join(HERE, "ref_%s.model.pkl" % name))
        Xi = to_idx(X, ent_to_idx=m.ent_to_idx, rel_to_idx=m.rel_to_idx)
        out["scores_" + name] = np.asarray(m._fn(E[Xi[:, 0]], R[Xi[:, 1]], E[Xi[:, 2]]), F32)
        out["E_" + name], out["R_" + name] = E, R
    return out


def gen_model_selection():
    """The call pattern of select_best_model_ranking (evaluation/protocol.py:1317-1703), RECORDED while the
    reference's own routine runs: a subclass of the reference's ComplEx records its constructor / fit arguments and
    the module's evaluate_performance is replaced by a recorder that returns made-up ranks.  What is written is the
    list of calls (JSON) — the contract a drop-in model class and evaluate_performance have to honour."""
    import json

    import emgraph.evaluation.protocol as proto
    calls = []

    def summ(v):
        if isinstance(v, np.ndarray):
            return {"ndarray": list(v.shape)}
        if isinstance(v, dict):
            return {k: summ(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [summ(x) for x in v]
        if isinstance(v, (np.integer, np.floating)):
            return v.item()
        return v if isinstance(v, (int, float, str, bool, type(None))) else repr(type(v))

    class Recording(ComplEx):
        name = "ComplEx"

        def __init__(self, **kw):
            calls.append({"call": "init", "kwargs": summ(kw)})
            super().__init__(**kw)

        def fit(self, *a, **kw):
            calls.append({"call": "fit", "args": summ(list(a)), "kwargs": summ(kw)})
            self.is_fitted = True

    Recording.__init__.__code__  # the routine inspects co_varnames (:1513)

    def fake_eval(X, **kw):
        m = kw["model"]
        calls.append({"call": "evaluate_performance", "args": summ([X]), "kwargs": summ({k: v for k, v in kw.items() if k != "model"}),
                      "model_k": m.all_params["k"]})
        n = len(X)
        base = 1 + (m.all_params["k"] % 7)
        return np.stack([np.arange(n) % 5 + base, np.arange(n) % 3 + base], 1)

    real_eval = proto.evaluate_performance
    proto.evaluate_performance = fake_eval
    try:
        rs = np.random.RandomState(7000)
        mk = lambda n: np.stack([rs.randint(0, 30, n), rs.randint(0, 3, n), rs.randint(0, 30, n)], 1)  # noqa: E731
        Xtr, Xva, Xte = mk(200), mk(20), mk(25)
        grid = {"batches_count": [2], "seed": 0, "epochs": [3], "k": [4, 8], "eta": [2, 4], "loss": ["nll"],
                "loss_params": {}, "embedding_model_params": {}, "regularizer": [None, "LP"],
                "regularizer_params": {"lambda": [1e-3]}, "optimizer": ["adam"], "optimizer_params": {"lr": [0.01]},
                "verbose": False}
        res = proto.select_best_model_ranking(Recording, Xtr, Xva, Xte, dict(grid), use_filter=True,
                                              early_stopping=True, early_stopping_params={"burn_in": 1, "check_interval": 1},
                                              retrain_best_model=True, corrupt_side="s,o", verbose=False)
    finally:
        proto.evaluate_performance = real_eval
    best_model, best_params, best_mrr, ranks_test, test_eval, history = res
    doc = {"param_grid": summ(grid), "shapes": {"train": list(Xtr.shape), "valid": list(Xva.shape), "test": list(Xte.shape)},
           "calls": calls, "best_params": summ(best_params), "best_mrr_train": float(best_mrr),
           "test_evaluation": summ(test_eval), "n_history": len(history),
           "history_keys": sorted(history[0].keys()), "result_keys": sorted(history[0]["results"].keys()),
           "init_varnames": list(ComplEx.__init__.__code__.co_varnames[1:ComplEx.__init__.__code__.co_argcount])}
    path = os.path.join(HERE, "model_selection_calls.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", path, len(calls), "calls")


def main():
    gen_model_selection()
    for name, fn in (("scores", gen_scores), ("scores_transe_p", gen_scores_transe_p), ("losses", gen_losses),
                     ("corruptions", gen_corruptions), ("misc", gen_misc), ("ranks", gen_ranks), ("checkpoints", gen_checkpoints)):
        data = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print("wrote", path, len(data), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
