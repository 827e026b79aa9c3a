"""cProfile of evaluate_performance() at the C4 shape (see tools/eval_throughput.py): where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd.evaluation import evaluate_performance  # noqa: E402
from emgraph_amd import models  # noqa: E402

name = os.environ.get("MODEL", "ComplEx")
rs = np.random.RandomState(0)
n_ent, n_rel, n = 1_000_000, 1000, 64 * 16384
X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
m = getattr(models, name)(k=200, eta=20, epochs=1, batches_count=64, seed=0, loss="nll", optimizer="sgd", optimizer_params={"lr": 0.0005})
m.fit(X)
Xte = X[:4096]
for _ in range(2):
    t0 = time.perf_counter()
    evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
    print("%s evaluate_performance %.3f s" % (name, time.perf_counter() - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
