"""evaluate_performance() end to end at the C4 shape: ComplEx k=200 on a synthetic |E|~1M graph, 4096 test triples, filter =
the whole graph (1M triples), corrupt_side 's,o'.  Prints wall time of the public call and where the host part goes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd.evaluation import evaluate_performance, mrr_score  # noqa: E402
from emgraph_amd.evaluation.ranking import FilterIndex  # noqa: E402
from emgraph_amd.models import ComplEx  # noqa: E402

rs = np.random.RandomState(0)
n_ent, n_rel, n = 1_000_000, 1000, 64 * 16384
X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
m = ComplEx(k=200, eta=20, epochs=1, batches_count=64, seed=0, loss="nll", optimizer="sgd", optimizer_params={"lr": 0.0005})
m.fit(X)
Xte = X[:4096]
for rep in range(3):
    t0 = time.perf_counter()
    ranks = evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
    dt = time.perf_counter() - t0
    print("evaluate_performance: %.3f s for %d ranks -> %.0f ranks/s (mrr %.4f)" % (dt, ranks.size, ranks.size / dt, mrr_score(ranks)), flush=True)
t0 = time.perf_counter()
F = FilterIndex(np.asarray(X))
print("FilterIndex over %d triples: %.3f s" % (len(X), time.perf_counter() - t0))

# a TRAINED-like model (round 6): the test triples' objects aligned with their object-side query vectors (tools/bench_exact_fast.py's
# "planted" tables) — the exact-fast path decides 99.9 % of the candidates in the half-precision prefilter
import torch  # noqa: E402
from emgraph_amd import device as D  # noqa: E402
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
for rep in range(4):
    t0 = time.perf_counter()
    ranks = evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
    dt = time.perf_counter() - t0
    print("trained-like tables: evaluate_performance %.4f s for %d ranks -> %.0f ranks/s (mean rank %.0f)" % (dt, ranks.size, ranks.size / dt, ranks.mean()), flush=True)
