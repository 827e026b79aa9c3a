"""cProfile of evaluate_performance() end to end at the C4 shape (where do the host milliseconds go?)"""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd.evaluation import evaluate_performance  # noqa: E402
from emgraph_amd.models import ComplEx  # noqa: E402
warnings.simplefilter("ignore")
rs = np.random.RandomState(0)
n_ent, n_rel, n = 1_000_000, 1000, 64 * 16384
X = np.stack([rs.randint(0, n_ent, n), rs.randint(0, n_rel, n), rs.randint(0, n_ent, n)], 1)
m = ComplEx(k=200, eta=20, epochs=1, batches_count=64, seed=0, loss="nll", optimizer="sgd", optimizer_params={"lr": 0.0005})
m.fit(X)
Xte = X[:4096]
for _ in range(2):
    evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
evaluate_performance(Xte, m, filter_triples=X, corrupt_side="s,o")
dt = time.perf_counter() - t0
pr.disable()
print("wall %.4f s" % dt)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
