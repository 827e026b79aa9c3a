"""cProfile of the host side of Trainer.step (C3 shapes, no stage timing)."""
import cProfile, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import _lib as L
from emgraph_amd.training import Trainer

n_ent, n_rel, k_int, B, eta, steps = 1_000_000, 1000, 400, 16384, 20, 300
rs = np.random.RandomState(0)
E = (rs.randn(n_ent, k_int) * 0.01).astype(np.float32); R = (rs.randn(n_rel, k_int) * 0.01).astype(np.float32)
X = np.stack([rs.randint(0, n_ent, 40 * B), rs.randint(0, n_rel, 40 * B), rs.randint(0, n_ent, 40 * B)], 1).astype(np.int32)
tr = Trainer(L.COMPLEX, k_int, 1.0, E, R, eta, loss="nll", optimizer="sgd", optimizer_params={"lr": 5e-4}, batches_count=40)
tr.set_training_set(X, B)
def run(n):
    for i in range(n):
        b = i % 36
        tr.step(b * B, B, 1, b + 1, prefetch=[((b + j) * B, B, 1, b + j + 1) for j in (1, 2)])
run(10); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(steps); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
