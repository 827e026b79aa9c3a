"""replay one seed of tests/test_api.py::test_fit_random_configurations_match_oracle_training_loop and say WHERE the fitted tables leave
the oracle's: per epoch count, the rows that differ, how many of their elements, and whether the difference looks like a flipped
sign of TransE-L1's gradient (a coordinate whose difference is within rounding of zero).  usage: python tools/dbg_fuzz_seed.py SEED"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_api as TA
from oracle import emgraph_oracle as orc
F32 = np.float32
seed = int(sys.argv[1])
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
if rs.randint(0, 2):
    w = 1.0 / np.arange(1, n_ent + 1); w /= w.sum()
    X = np.stack([rs.choice(n_ent, n, p=w), rs.randint(0, n_rel, n), rs.choice(n_ent, n, p=w)], 1)
else:
    X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
ids = np.unique(np.concatenate([X[:, 0], X[:, 2]]))
remap = np.full(n_ent, -1, np.int64); remap[ids] = np.arange(len(ids))
X = np.stack([remap[X[:, 0]], X[:, 1], remap[X[:, 2]]], 1).astype(np.int64)
rels = np.unique(X[:, 1]); X[:, 1] = np.searchsorted(rels, X[:, 1])
n_ent, n_rel = len(ids), len(rels)
ki = 2 * k if name in ("ComplEx", "HolE") else k
ent0 = (rs.randn(n_ent, ki) * 0.3).astype(F32); rel0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
emp = {"corrupt_side": list(sides) if len(sides) > 1 else sides[0]}
if name == "TransE": emp["norm"] = norm
reg, reg_kw = None, {}
if rs.randint(0, 3) == 0:
    reg = {"lam": float(rs.choice([0.001, 0.01])), "p": int(rs.choice([1, 2, 3]))}
    reg_kw = dict(regularizer="LP", regularizer_params={"lambda": reg["lam"], "p": reg["p"]})
print("config:", (name, norm, k, eta, loss, opt, sides, n_ent, n_rel, n, bc, epochs, lr, reg))
omodel = ("TransE_L%d" % norm) if name == "TransE" else name
for ep in range(1, epochs + 1):
    m = TA._models()[name](k=k, eta=eta, epochs=ep, batches_count=bc, seed=seed, loss=loss, optimizer=opt, optimizer_params={"lr": lr},
                           embedding_model_params=emp, initializer="constant", initializer_params={"entity": ent0, "relation": rel0}, **reg_kw)
    m.fit(X)
    E, R, losses = TA.oracle_fit(omodel, k, X.astype(np.int32), ent0, rel0, eta, ep, bc, seed, loss, None, opt, lr, sides=sides, reg=reg)
    G = m.trained_model_params[0]
    bad = ~np.isclose(G, E, rtol=2e-3, atol=2e-5)
    rows = np.nonzero(bad.any(1))[0]
    print("epochs %d: %d elements off in %d rows; losses got %s oracle %s" % (ep, int(bad.sum()), len(rows), [round(float(x), 5) for x in m.epoch_losses], [round(float(x), 5) for x in losses]))
    for r_ in rows[:12]:
        d = G[r_] - E[r_]
        print("   row %d: %d off, max |diff| %.3e, diff values (first 6 off): %s ; in X as s %d, as o %d" % (
            r_, int(bad[r_].sum()), np.abs(d).max(), np.round(d[bad[r_]][:6], 5).tolist(), int((X[:, 0] == r_).sum()), int((X[:, 2] == r_).sum())))
    bR = ~np.isclose(m.trained_model_params[1], R, rtol=2e-3, atol=2e-5)
    print("   relation table: %d elements off" % int(bR.sum()))
