"""cProfile of evaluate_performance() on TRAINED-like tables at the C4 shape (tools/eval_throughput.py's second half): the
kernels take ~7 ms of the call, this prints where the rest goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import device as D  # noqa: E402
from emgraph_amd.evaluation import evaluate_performance  # noqa: E402
from emgraph_amd.models import ComplEx  # noqa: E402

rs = np.random.RandomState(0)
n_ent, n_rel, n = 1_000_000, 1000, 64 * 16384
X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
m = ComplEx(k=200, eta=20, epochs=1, batches_count=64, seed=0, loss="nll", optimizer="sgd", optimizer_params={"lr": 0.0005})
m.fit(X)
Xte = X[:4096]
ent, rel = m._device_tables()
T = torch.from_numpy(np.ascontiguousarray(np.stack([np.vectorize(m.ent_to_idx.get)(Xte[:, 0]), np.vectorize(m.rel_to_idx.get)(Xte[:, 1]),
                                                     np.vectorize(m.ent_to_idx.get)(Xte[:, 2])], 1).astype(np.int32))).cuda()
ki = m.internal_k
with torch.no_grad():
    ent.normal_(0.0, 0.1)
    rel.normal_(0.0, 0.1)
    Q, _ = D.eval_build_queries(3, ent, rel, ki, 1.0, T, 1)
    b, o = 0.15, T[:, 2].long()
    eo = ent[o]
    qh = Q[:, :ki] / Q[:, :ki].norm(dim=1, keepdim=True)
    ent[o] = (1 - b * b) ** 0.5 * eo + b * eo.norm(dim=1, keepdim=True) * qh
m._derived_cache = None
for _ in range(3):
    t0 = time.perf_counter()
    evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
    print("evaluate_performance %.4f s" % (time.perf_counter() - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
