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


def main():
    for name, fn in (("scores", gen_scores), ("losses", gen_losses), ("corruptions", gen_corruptions),
                     ("misc", gen_misc)):
        data = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print("wrote", path, len(data), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
