"""CPU tests of the host logic (no GPU, no compute calls): the C-ABI library loads and exports every
declared symbol; filter CSR == the oracle's per-triple SQLite semantics; rank assembly; C oracle ==
numpy oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import emgraph_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def test_library_loads_and_exports_every_declared_symbol():
    from emgraph_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "emgraph_hip.h")).read()
    declared = set(re.findall(r"\b(emg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert os.path.exists(L.LIB_PATH), "libemgraph_hip.so is not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    handle = L.load()
    assert handle.emg_version() == L.ABI_VERSION
    assert handle.emg_target() == b"gfx950"


def test_prefilter_layout_queries_are_host_arithmetic():
    """the row strides / segment counts the ranking prefilters ask of their callers (include/emgraph_hip.h): pure host
    functions of the sizes, answered without a GPU"""
    from emgraph_amd import _lib as L
    lib = L.load()
    # half-precision MFMA prefilter: instantiated for 4, 7, 8, 10, 13, 16, 19, 22, 25 and (4-wave form) 32, 38, 44, 50 k-steps of 16,
    # rows fetched 64 columns at a time
    for k_cols, ld in ((8, 64), (64, 64), (65, 128), (100, 128), (128, 128), (130, 192), (200, 256), (202, 256), (256, 256),
                       (257, 320), (300, 320), (353, 448), (400, 448), (401, 512), (600, 640), (800, 832), (801, 832)):
        assert lib.emg_eval_prefilter_ld(k_cols) == ld, (k_cols, lib.emg_eval_prefilter_ld(k_cols))
        assert lib.emg_eval_prefilter_ld(k_cols) >= k_cols and lib.emg_eval_prefilter_ld(k_cols) % 64 == 0
    assert lib.emg_eval_prefilter_ld(0) == 0
    # fixed-point (v_sad_u16) prefilter: u16 image rows padded to whole 8-dword k tiles
    for k, ld in ((1, 16), (16, 16), (17, 32), (200, 208), (208, 208)):
        assert lib.emg_eval_sad_ld(k) == ld
    # one pair-buffer segment per wave of the prefilter grids (8 waves per 256 rows x 4096 entities; 4 waves per 128 x 4096)
    assert lib.emg_eval_prefilter_segments(0, 10) == 0 and lib.emg_eval_sad_segments(5, 0) == 0
    assert lib.emg_eval_prefilter_segments(8192, 1_000_000) == 8 * 8 * 32 * 31
    assert lib.emg_eval_prefilter_segments_k(8192, 1_000_000, 400) == 8 * 8 * 32 * 31
    assert lib.emg_eval_prefilter_segments_k(8192, 1_000_000, 800) == 4 * 8 * 64 * 31   # 4 waves x 128 query rows above 400 columns
    assert lib.emg_eval_prefilter_waves(400) == 8 and lib.emg_eval_prefilter_waves(416) == 4 and lib.emg_eval_prefilter_max_cols() == 800
    assert lib.emg_eval_sad_segments(4096, 1_000_000) == 4 * 8 * 32 * 31


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "emgraph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/emg_oracle.c", "").replace("oracle/emgraph_oracle.py", "") \
                    .replace("oracle/emg_oracle.c", "") or f in ("emg_rank.hip", "emg_common.hpp"), \
                    "%s mentions the oracle" % f
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), "%s imports the oracle" % f


def test_no_gpu_means_loud_failure():
    import torch
    from emgraph_amd import device
    from emgraph_amd._lib import EmgError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(EmgError):
        device.require_gpu()


@pytest.mark.parametrize("side", ["s", "o", "s+o", "s,o"])
def test_filter_csr_matches_per_triple_semantics(side):
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import build_filter_csr
    rs = np.random.RandomState(0)
    n_ent, n_rel = 30, 3
    F = np.stack([rs.randint(0, n_ent, 600), rs.randint(0, n_rel, 600), rs.randint(0, n_ent, 600)], 1)
    F = np.concatenate([F, F[:50]])  # duplicates in the filter must not double count (SELECT DISTINCT)
    T = np.concatenate([F[:10], np.stack([rs.randint(0, n_ent, 15), rs.randint(0, n_rel, 15), rs.randint(0, n_ent, 15)], 1)])
    ptr, idx = build_filter_csr(F, T, L.EVAL_SIDE_IDS[side], n_ent)
    n_q = len(T)
    for qi, x in enumerate(T):
        objs, subs = orc.participating_entities(F, x)
        if side == "o":
            np.testing.assert_array_equal(idx[ptr[qi]:ptr[qi + 1]], objs)
        elif side == "s":
            np.testing.assert_array_equal(idx[ptr[qi]:ptr[qi + 1]], subs)
        else:
            np.testing.assert_array_equal(idx[ptr[qi]:ptr[qi + 1]], objs)
            np.testing.assert_array_equal(idx[ptr[n_q + qi]:ptr[n_q + qi + 1]], subs)
    # with a subset: only members survive
    sub = np.array([1, 4, 9, 16, 25])
    ptr2, idx2 = build_filter_csr(F, T, L.EVAL_SIDE_IDS[side], n_ent, entities_subset=sub)
    assert np.all(np.isin(idx2, sub)) and ptr2[-1] == len(idx2)
    # empty filter / empty test
    p0, i0 = build_filter_csr(np.zeros((0, 3), int), T, L.EVAL_SIDE_IDS[side], n_ent)
    assert p0[-1] == len(i0) == (2 * n_q if side in ("s+o", "s,o") else n_q)  # only the self entries
    p1, i1 = build_filter_csr(F, np.zeros((0, 3), int), L.EVAL_SIDE_IDS[side], n_ent)
    assert len(i1) == 0 and p1[-1] == 0


def _numpy_counts(model, E, R, T, side_mode, k, filt=None):
    """(gt, eq, fgt, feq) rows from the LITERAL numpy oracle scores"""
    rows = []
    for obj_side in ([True, False] if side_mode >= 2 else [side_mode == 1]):
        for x in T:
            C = np.arange(E.shape[0])
            corr = orc.generate_corruptions_for_eval(x, C, "o" if obj_side else "s")
            sc = orc.to_cmp_int(orc.score_triples(model, E, R, corr, k=k))
            p = orc.to_cmp_int(orc.score_triples(model, E, R, x[None], k=k))[0]
            fg = fe = 0
            if filt is not None:
                objs, subs = orc.participating_entities(filt, x)
                sel = sc[objs if obj_side else subs]
                fg, fe = int((sel > p).sum()), int((sel == p).sum())
            rows.append((int((sc > p).sum()), int((sc == p).sum()), fg, fe))
    return [np.array(c) for c in zip(*rows)]


@pytest.mark.parametrize("strategy", ["worst", "best", "middle"])
@pytest.mark.parametrize("side", ["s", "o", "s+o", "s,o"])
def test_rank_assembly_matches_oracle(side, strategy):
    from emgraph_amd import _lib as L
    from emgraph_amd.evaluation import ranks_from_counts
    rs = np.random.RandomState(3)
    k, n_ent = 3, 25
    E = (rs.randint(-2, 3, (n_ent, k)) / 2.0).astype(F32)  # many exact ties
    R = (rs.randint(-2, 3, (2, k)) / 2.0).astype(F32)
    T = np.stack([rs.randint(0, n_ent, 12), rs.randint(0, 2, 12), rs.randint(0, n_ent, 12)], 1)
    F = np.concatenate([T, np.stack([rs.randint(0, n_ent, 200), rs.randint(0, 2, 200), rs.randint(0, n_ent, 200)], 1)])
    for filt in (None, F):
        gt, eq, fgt, feq = _numpy_counts("DistMult", E, R, T, L.EVAL_SIDE_IDS[side], k, filt)
        got = ranks_from_counts(gt, eq, fgt, feq, len(T), side, strategy)
        exp = orc.get_ranks("DistMult", E, R, T, corrupt_side=side, strategy=strategy, filter_triples=filt, k=k)
        np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("model", ["TransE_L1", "TransE_L2", "DistMult", "ComplEx", "HolE"])
def test_c_oracle_matches_numpy_oracle(model):
    """validate the C restatement (canonical order) against the literal numpy restatement"""
    rs = np.random.RandomState(12)
    k, n_ent, n_rel = 16, 200, 4
    ki = 2 * k if model in ("ComplEx", "HolE") else k
    mid = orc.MODEL_IDS[model]
    sc = float(F32(2 / k)) if model == "HolE" else 1.0
    E = (rs.randn(n_ent, ki) * 0.4).astype(F32)
    R = (rs.randn(n_rel, ki) * 0.4).astype(F32)
    X = np.stack([rs.randint(0, n_ent, 64), rs.randint(0, n_rel, 64), rs.randint(0, n_ent, 64)], 1).astype(np.int32)
    eta = 4
    codes = co.corrupt_codes(64, eta, 2, n_ent, 1, 2)
    xneg = orc.generate_corruptions_for_fit_philox(X, eta=eta, corrupt_side="s,o", entities_size=n_ent, seed=1, counter=2)
    sp, sn = co.train_forward(mid, E, R, ki, sc, X, eta, codes)
    np.testing.assert_allclose(sp, orc.score_triples(model, E, R, X, k=k), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sn, orc.score_triples(model, E, R, xneg, k=k), rtol=1e-5, atol=1e-6)
    # canonical 1-vs-all scores == literal scores of the eval corruptions (to fp32 noise)
    T = X[:5]
    Q, pos_int = co.build_queries(mid, E, R, ki, sc, T, 3)
    S = co.scores_dense(mid, Q, E, ki, sc)
    for qi, x in enumerate(T):
        corr = orc.generate_corruptions_for_eval(x, np.arange(n_ent), "s,o")
        lit = orc.score_triples(model, E, R, corr, k=k)
        np.testing.assert_allclose(S[qi], lit[:n_ent], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(S[5 + qi], lit[n_ent:], rtol=1e-4, atol=2e-5)
    # on exact (dyadic) data the canonical pipeline reproduces the literal ranks bit for bit
    from emgraph_amd.evaluation import build_filter_csr, ranks_from_counts
    Ed = (rs.randint(-4, 5, (n_ent, ki)) / 4.0).astype(F32)
    Rd = (rs.randint(-4, 5, (n_rel, ki)) / 4.0).astype(F32)
    Fd = np.concatenate([X, np.stack([rs.randint(0, n_ent, 2000), rs.randint(0, n_rel, 2000), rs.randint(0, n_ent, 2000)], 1)])
    Q, pos_int = co.build_queries(mid, Ed, Rd, ki, sc, X[:20], 3)
    gt, eq = co.count(mid, Q, pos_int, Ed, ki, sc)
    ptr, idx = build_filter_csr(Fd, X[:20], 3, n_ent)
    fgt, feq = co.filter_count(mid, Q, pos_int, Ed, 0, ki, sc, ptr, idx)
    for strategy in ("worst", "best", "middle"):
        got = ranks_from_counts(gt, eq, fgt, feq, 20, "s,o", strategy)
        exp = orc.get_ranks(model, Ed, Rd, X[:20], corrupt_side="s,o", strategy=strategy, filter_triples=Fd, k=k)
        np.testing.assert_array_equal(got, exp)


def test_sgd_lr_schedule_reference_goldens():
    """tests/emgraph/models/test_optimizers.py:21,41-43,72-79 (values of update_feed_dict)"""
    from emgraph_amd.training import sgd_learning_rate as lr_at
    assert all(lr_at({"lr": 0.001}, 10, e, b) == 0.001 for e in range(1, 11) for b in range(1, 11))
    fixed = {"lr": 0.001, "decay_lr_rate": 2, "cosine_decay": False, "decay_cycle": 10}
    assert all(lr_at(fixed, 10, e, b) == 0.001 for e in range(1, 11) for b in range(1, 11))
    assert lr_at(fixed, 10, 11, 1) == 0.0005 and lr_at(fixed, 10, 20, 10) == 0.0005 and lr_at(fixed, 10, 21, 1) == 0.00025
    assert lr_at(dict(fixed, end_lr=0.0004), 10, 300, 1) == 0.0004      # never below end_lr; the schedule then stops
    cos = {"lr": 0.001, "end_lr": 0.00001, "decay_lr_rate": 2, "expand_factor": 2, "cosine_decay": True,
           "decay_cycle": 10}
    assert lr_at(cos, 10, 1, 1) == 0.001
    assert lr_at(cos, 10, 6, 1) == 0.000505         # half-way through the first 10-epoch cycle
    assert lr_at(cos, 10, 11, 1) == 0.0005          # restart at half the rate, cycle length doubled
    assert lr_at(cos, 10, 21, 1) == 0.000255
    assert lr_at(cos, 10, 31, 1) == 0.00025         # second restart after 10 + 20 epochs
    rates = [lr_at(cos, 10, e, b) for e in range(1, 31) for b in range(1, 11)]
    assert all(x >= 0.00001 for x in rates) and rates[:100] == sorted(rates[:100], reverse=True)


def test_mappings_and_to_idx_match_oracle():
    from emgraph_amd.evaluation.protocol import create_mappings, create_mappings_and_index, to_idx
    X = np.array([["a", "x", "b"], ["c", "y", "d"], ["b", "x", "a"], ["zz", "y", "a"]])
    r1, e1 = create_mappings(X)
    r2, e2 = orc.create_mappings(X)
    assert r1 == r2 and e1 == e2
    np.testing.assert_array_equal(to_idx(X, e1, r1), orc.to_idx(X, e2, r2))
    r3, e3, Xi = create_mappings_and_index(X)
    assert r3 == r2 and e3 == e2
    np.testing.assert_array_equal(Xi, orc.to_idx(X, e2, r2))
    # the reference's golden (tests/emgraph/evaluation/test_protocol.py:490-496)
    Xg = np.array([["a", "x", "b"], ["c", "y", "d"]])
    rg, eg = create_mappings(Xg)
    np.testing.assert_array_equal(to_idx(Xg, eg, rg), [[0, 0, 1], [2, 1, 3]])
    np.testing.assert_array_equal(to_idx(np.array(["c", "y", "d"]), eg, rg), [[2, 1, 3]])  # 1-D input
    with pytest.raises(ValueError, match="entities"):
        to_idx(np.array([["a", "x", "q"]]), eg, rg)
    with pytest.raises(ValueError, match="relations"):
        to_idx(np.array([["a", "q", "b"]]), eg, rg)
    # integer labels
    Xn = np.array([[5, 1, 7], [7, 2, 9]])
    rn, en = create_mappings(Xn)
    np.testing.assert_array_equal(to_idx(Xn, en, rn), [[0, 0, 1], [1, 1, 2]])


def test_filter_unseen_entities_matches_isin_semantics():
    """protocol.py:1014-1041: keep exactly the triples whose subject AND object are keys of ent_to_idx — string labels,
    dense and sparse integer labels (the cached gather table and the sorted-key search), nothing to remove -> same array"""
    from types import SimpleNamespace
    from emgraph_amd.evaluation.protocol import filter_unseen_entities
    rs = np.random.RandomState(0)
    for labels in (np.array(["e%d" % i for i in range(50)]), np.arange(100, 150), np.arange(0, 50) * 100003):
        model = SimpleNamespace(ent_to_idx={(v.item() if hasattr(v, "item") else v): i for i, v in enumerate(labels)})
        extra = np.array(["zz", "e999"]) if labels.dtype.kind in "US" else np.array([-5, 7, 10 ** 9])
        pool = np.concatenate([labels, extra])
        X = np.stack([rs.choice(pool, 400), rs.choice(labels, 400), rs.choice(pool, 400)], 1)
        seen = np.array(list(model.ent_to_idx.keys()))
        keep = np.isin(X[:, 0], seen) & np.isin(X[:, 2], seen)
        assert 0 < keep.sum() < len(X)
        np.testing.assert_array_equal(filter_unseen_entities(X, model), X[keep])
        assert filter_unseen_entities(X[keep], model) is not None and len(filter_unseen_entities(X[keep], model)) == keep.sum()


def test_metrics_match_reference_goldens(golden):
    from emgraph_amd.evaluation import hits_at_n_score, mr_score, mrr_score, rank_score
    g = golden("misc")
    r = g["metric_ranks"]
    assert mrr_score(r) == g["metric_mrr"] and mr_score(r) == g["metric_mr"]
    for n in (1, 3, 10):
        assert hits_at_n_score(r, n) == g["metric_hits%d" % n]
    r2 = g["metric_ranks2"]
    assert mrr_score(r2) == g["metric2_mrr"] and mr_score(r2) == g["metric2_mr"] and hits_at_n_score(r2, 1) == g["metric2_hits1"]
    assert mrr_score([[1, 2], [3, 1], [10, 20]]) == g["metric2_mrr"]
    # tests/emgraph/evaluation/test_metrics.py:6-39
    assert rank_score(np.array([0, 0, 1, 0]), np.array([0.434, 0.65, 0.21, 0.84])) == 4
    assert mr_score(np.array([0.2, 0.4, 0.6, 0.8])) == 0.5


def test_eval_corruption_layout_matches_reference_golden(golden):
    from emgraph_amd.evaluation import generate_corruptions_for_eval
    g = golden("corruptions")
    x = g["toy_X_idx"][0]
    for side in ("s,o", "s+o", "s", "o"):
        np.testing.assert_array_equal(generate_corruptions_for_eval(x, np.arange(8), side), g["eval_" + side])
    with pytest.raises(ValueError):
        generate_corruptions_for_eval(x, np.arange(8), "x")


# ---- checkpoint interchange and the model-selection call contract (fixtures written by the REFERENCE's own code,
# ---- tests/golden/make_golden.py::gen_checkpoints / gen_model_selection) ---------------------------------------
@pytest.mark.parametrize("name", ["TransE", "ComplEx", "HolE"])
def test_restore_model_reads_files_written_by_the_reference(name, golden, tmp_path):
    """utils/model_utils.py:63-87 layout: a file pickled by the reference's save_model restores into this package's
    class with the same hyper-parameters, dictionaries and parameter arrays; and what this package's save_model
    writes has the same keys and value types, so the reference's restore_model (:142-154) reads it the same way."""
    import pickle

    from emgraph_amd.utils import restore_model, save_model
    path = os.path.join(ROOT, "tests", "golden", "ref_%s.model.pkl" % name)
    with open(path, "rb") as f:
        raw = pickle.load(f)   # plain python / numpy objects only: no class of the reference is needed to load it
    assert sorted(raw) == ["calibration_parameters", "class_name", "ent_to_idx", "hyperparams", "is_calibrated",
                           "is_fitted", "large_graph", "model_params", "rel_to_idx"]
    m = restore_model(path)
    g = golden("checkpoints")
    assert type(m).__name__ == name == raw["class_name"] and m.is_fitted
    assert m.all_params == raw["hyperparams"] and m.ent_to_idx == raw["ent_to_idx"] and m.rel_to_idx == raw["rel_to_idx"]
    np.testing.assert_array_equal(m.trained_model_params[0], g["E_" + name])
    np.testing.assert_array_equal(m.trained_model_params[1], g["R_" + name])
    assert m.k == raw["hyperparams"]["k"] and m.internal_k == g["E_" + name].shape[1]
    np.testing.assert_array_equal(m.get_embeddings(np.array(["ent_03"])), g["E_" + name][3:4])
    if name == "TransE":
        assert m.embedding_model_params["norm"] == 2 and m.loss_params == {"margin": 2.0}
    out = os.path.join(tmp_path, "roundtrip.pkl")
    save_model(m, out)
    with open(out, "rb") as f:
        again = pickle.load(f)
    assert sorted(again) == sorted(raw)
    for key in raw:
        assert type(again[key]) is type(raw[key]), key
    assert again["hyperparams"] == raw["hyperparams"] and again["class_name"] == raw["class_name"]
    for a, b in zip(again["model_params"], raw["model_params"]):
        assert isinstance(a, np.ndarray) and a.dtype == b.dtype
        np.testing.assert_array_equal(a, b)


def test_model_selection_call_contract():
    """select_best_model_ranking (evaluation/protocol.py:1317-1703) is a pure caller of the model class and of
    evaluate_performance.  tests/golden/model_selection_calls.json is the list of calls the reference's routine
    made when it was run (with a recording subclass of its ComplEx) over a 2x2x2 grid with early stopping, a filter
    and retrain_best_model: every one of them must bind to this package's signatures, and the class attributes the
    routine inspects (`name`, `__init__.__code__.co_varnames`, :1513-1525) must be there."""
    import inspect
    import json

    from emgraph_amd.evaluation import evaluate_performance, hits_at_n_score, mr_score, mrr_score
    from emgraph_amd.models import ComplEx
    with open(os.path.join(ROOT, "tests", "golden", "model_selection_calls.json")) as f:
        doc = json.load(f)
    assert ComplEx.name == "ComplEx"
    ours = ComplEx.__init__.__code__.co_varnames[1:ComplEx.__init__.__code__.co_argcount]
    assert list(ours) == doc["init_varnames"]                       # same constructor arguments, same order
    kinds = [c["call"] for c in doc["calls"]]
    assert kinds == ["init", "fit", "evaluate_performance"] * 8 + ["fit", "evaluate_performance"]
    for c in doc["calls"]:
        if c["call"] == "init":
            m = ComplEx(**c["kwargs"])                              # accepted as they are (no GPU needed to construct)
            assert m.get_hyperparameter_dict()["k"] == c["kwargs"]["k"]
        elif c["call"] == "fit":
            b = inspect.signature(ComplEx.fit).bind(None, *c["args"], **c["kwargs"])
            assert list(b.arguments)[1:] == ["X", "early_stopping", "early_stopping_params"]   # passed POSITIONALLY
        else:
            inspect.signature(evaluate_performance).bind(*c["args"], model=None, **c["kwargs"])
    ranks = np.stack([np.arange(25) % 5 + 2, np.arange(25) % 3 + 2], 1)     # what the recorder returned for k = 8
    assert mrr_score(ranks) == pytest.approx(doc["test_evaluation"]["mrr"]) and mr_score(ranks) == doc["test_evaluation"]["mr"]
    assert hits_at_n_score(ranks, n=3) == doc["test_evaluation"]["hits_3"]


@pytest.mark.parametrize("model", ["TransE_L1", "TransE_L2", "DistMult", "ComplEx", "HolE"])
def test_cpu_baseline_agrees_with_the_checker(model):
    """oracle/emg_cpu_fast.c — the OPTIMISED CPU leg bench.py times as cpu_baseline (SIMD reductions, hoisted query
    vectors) — computes the same scores as the order-pinned checker (oracle/emg_oracle.c) up to fp32 reassociation, at
    the benchmark's width (k = 200) and at an odd width"""
    rs = np.random.RandomState(3)
    for k in (200, 37):
        n_ent, n_rel, B, eta = 3000, 7, 257, 5
        ki = 2 * k if model in ("ComplEx", "HolE") else k
        mid = orc.MODEL_IDS[model]
        sc = float(F32(2 / k)) if model == "HolE" else 1.0
        E = (rs.randn(n_ent, ki) * 0.3).astype(F32)
        R = (rs.randn(n_rel, ki) * 0.3).astype(F32)
        X = np.stack([rs.randint(0, n_ent, B), rs.randint(0, n_rel, B), rs.randint(0, n_ent, B)], 1).astype(np.int32)
        codes = co.corrupt_codes(B, eta, 2, n_ent, 5, 9)
        sp, sn = co.train_forward(mid, E, R, ki, sc, X, eta, codes)
        fp, fn = co.fast_train_forward(mid, E, R, ki, sc, X, eta, codes, native=False)
        mag = np.abs(E).max() ** 2 * np.abs(R).max() * ki if not model.startswith("TransE") else np.abs(sp).max()
        np.testing.assert_allclose(fp, sp, rtol=1e-4, atol=1e-5 * mag)
        np.testing.assert_allclose(fn, sn, rtol=1e-4, atol=1e-5 * mag)


def _device_kernels_of_the_library():
    """(symbol, private segment bytes, VGPRs) of every gfx950 kernel in libemgraph_hip.so: the clang offload bundles inside the
    library are split by hand (magic, entry table) and each gfx950 code object's metadata note read with llvm-readelf"""
    import re
    import struct
    import subprocess
    import tempfile
    from emgraph_amd import _lib as L
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not in this image")
    data = open(L.LIB_PATH, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        while True:
            i = data.find(magic, pos)
            if i < 0:
                break
            cnt = struct.unpack_from("<Q", data, i + 24)[0]
            p = i + 32
            for _ in range(cnt):
                off, size, tl = struct.unpack_from("<QQQ", data, p)
                p += 24
                triple = data[p:p + tl].decode()
                p += tl
                if "gfx950" in triple and size > 0:
                    f = os.path.join(tmp, "co_%d.o" % len(out))
                    open(f, "wb").write(data[i + off:i + off + size])
                    notes = subprocess.run([readelf, "--notes", f], capture_output=True, text=True, check=True).stdout
                    syms = re.findall(r"\.symbol:\s+(\S+)", notes)
                    priv = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)
                    vgpr = re.findall(r"\.vgpr_count:\s+(\d+)", notes)
                    assert len(syms) == len(priv) == len(vgpr)
                    out += list(zip(syms, map(int, priv), map(int, vgpr)))
            pos = i + 24
    return out


def test_no_hand_written_kernel_spills_to_scratch():
    """every emg:: kernel of the library has a private segment of 0 bytes (round 4: the eleven window-form scoring kernels — the
    default-optimizer path — spilled 12 .. 84 bytes per lane at their cap of three waves per SIMD; rocPRIM's radix sort, behind
    the wide-key grouping fallback, is not ours and is left out)"""
    ks = _device_kernels_of_the_library()
    ours = [k for k in ks if k[0].startswith("_ZN3emg")]
    assert len(ours) > 300, len(ours)
    spilling = [(s, b, v) for s, b, v in ours if b != 0]
    assert not spilling, spilling
    fused = [k for k in ours if "train_fused_riders_kernel" in k[0]]
    assert len(fused) >= 100 and max(v for _, _, v in fused) <= 512


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """the argument structures cross the C-ABI by pointer: the ctypes mirrors in emgraph_amd/_lib.py must have the size of the
    structs in include/emgraph_hip.h and every field at the header's offset (a field added on one side only would shift everything
    behind it silently).  A C program that includes the header prints sizeof / offsetof; gcc is part of the image."""
    import ctypes as C
    import shutil
    import subprocess
    from emgraph_amd import _lib as L
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pairs = [("emg_backward_args", L.BackwardArgs), ("emg_prepare_args", L.PrepareArgs), ("emg_step_args", L.StepArgs),
             ("emg_apply_args", L.ApplyArgs), ("emg_plan_slot", L.PlanSlot), ("emg_plan_config", L.PlanConfig),
             ("emg_plan_batch", L.PlanBatch)]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "emgraph_hip.h"', "int main(void) {"]
    for cname, cls in pairs:
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for cname, cls in pairs:
        assert got[(cname, "sizeof")] == C.sizeof(cls), (cname, got[(cname, "sizeof")], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_installed_filter_is_remembered_by_content_not_by_identity():
    """evaluate_performance maps the filter triples through the label dictionaries and indexes them on every call (45 of 107 ms at
    1M filter triples); the mapped set and its FilterIndex are remembered by a digest of the array's CONTENT together with the
    dictionaries' identity: an equal array (another object) hits, a changed one or other mappings miss"""
    from types import SimpleNamespace
    from emgraph_amd.datasets import NumpyDatasetAdapter
    from emgraph_amd.evaluation import protocol as P
    rs = np.random.RandomState(0)
    F = np.stack([rs.randint(0, 50, 400), rs.randint(0, 5, 400), rs.randint(0, 50, 400)], 1)
    ent = {i: i for i in range(50)}
    rel = {i: i for i in range(5)}
    model = SimpleNamespace(ent_to_idx=ent, rel_to_idx=rel, set_filter_for_eval=lambda: None)

    def install(arr, m=model):
        ad = NumpyDatasetAdapter()
        ad.use_mappings(m.rel_to_idx, m.ent_to_idx)
        P._install_filter(ad, False, arr, m, True, False)
        return ad
    P._FILTER_CACHE.clear()
    a, b = install(F), install(F.copy())
    assert a.filter_index is b.filter_index and a.filter_adapter is b.filter_adapter
    np.testing.assert_array_equal(a.filter_adapter, F)
    b.cleanup()                                                              # an adapter's cleanup does not break the remembered index
    assert install(F).filter_index is a.filter_index
    G = F.copy()
    G[7, 0] = (G[7, 0] + 1) % 50
    c = install(G)
    assert c.filter_index is not a.filter_index
    np.testing.assert_array_equal(c.filter_adapter, G)
    other = SimpleNamespace(ent_to_idx=dict(ent), rel_to_idx=rel, set_filter_for_eval=lambda: None)   # an equal but different dictionary
    assert install(F, other).filter_index is not a.filter_index
    labels = np.array([["a", "r", "b"], ["b", "r", "a"]], dtype=object)      # object arrays are not digested: never cached
    assert P._array_digest(labels) is None
    # same size, same first / last items, another interior mapping (a refit with another insertion order): a miss, and the mapped
    # filter follows the new ids — the entry holds its dictionaries, so an address reused after a model is gone cannot alias it
    swapped = dict(ent)
    swapped[3], swapped[4] = ent[4], ent[3]
    d = install(F, SimpleNamespace(ent_to_idx=swapped, rel_to_idx=rel, set_filter_for_eval=lambda: None))
    assert d.filter_index is not a.filter_index
    expect = F.copy()
    for col in (0, 2):
        expect[:, col] = np.where(F[:, col] == 3, 4, np.where(F[:, col] == 4, 3, F[:, col]))
    np.testing.assert_array_equal(d.filter_adapter, expect)
    assert all(e[3] is not None and e[4] is not None for e in P._FILTER_CACHE)
    P.clear_filter_cache()
    assert P._FILTER_CACHE == [] and P._LOOKUP_CACHE == []
