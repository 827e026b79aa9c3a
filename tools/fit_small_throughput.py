"""fit() end to end at the reference's own configurations (BASELINE configs 1 / 2 / 5 shapes, synthetic): wall time per batch through the
public API against the step time bench.py reports, and the one-off part (label mapping, initialisation, plan / graph capture, snapshot)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd.models import TransE, DistMult, HolE

CFG = {"C1": (TransE, dict(k=100, eta=20, batches_count=64, loss="pairwise"), 38600, 11, 110361),
       "C2": (DistMult, dict(k=200, eta=10, batches_count=100, loss="nll"), 14541, 237, 272115),
       "C5": (HolE, dict(k=200, eta=20, batches_count=100, loss="nll"), 14951, 1345, 483142)}
for name in sys.argv[1:] or ["C1", "C2", "C5"]:
    cls, kw, n_ent, n_rel, n = CFG[name]
    rs = np.random.RandomState(1234)
    X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
    out = {}
    for epochs in (1, 21):
        m = cls(epochs=epochs, seed=0, optimizer="adam", optimizer_params={"lr": 0.0005}, **kw)
        t0 = time.perf_counter()
        m.fit(X)
        out[epochs] = time.perf_counter() - t0
    per_batch = (out[21] - out[1]) / (20 * kw["batches_count"])
    print("%s: fit() 1 epoch %.3f s, 21 epochs %.3f s -> %.4f ms per batch, one-off %.3f s" % (name, out[1], out[21], per_batch * 1e3, out[1] - per_batch * kw["batches_count"]), flush=True)
