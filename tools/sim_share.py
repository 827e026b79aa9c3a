"""One rank's share of the k-sharded N-GPU training step, on ONE GPU (no collective: what the rank computes between
all-reduces).  Weak scaling as bench.py does it: B_global = N x 16384 groups, every rank walks all of them on its
1/N-wide column slab.   python tools/sim_share.py [N ...]     -> ms/step and stage times per N
The all-reduce of the partial scores (B_global x 21 floats) comes on top: N=8 -> 11 MB per step."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from emgraph_amd import parallel  # noqa: E402
from emgraph_amd.training import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("worlds", nargs="*", type=int, default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
w = bench.WORKLOADS["C3"]
k_int = 2 * w["k"]
rs = np.random.RandomState(0)
ent0, rel0 = bench.glorot(rs, w["n_ent"], k_int), bench.glorot(rs, w["n_rel"], k_int)
for N in a.worlds:
    B = w["B"] * N
    nb = 8
    X = bench.make_triples(w, nb * B, 1234)
    e, r = (parallel.shard_columns(ent0, 0, N, True), parallel.shard_columns(rel0, 0, N, True)) if N > 1 else (ent0, rel0)
    tr = Trainer(bench.MODEL_IDS[w["model"]], e.shape[1], 1.0, e, r, w["eta"], loss=w["loss"], optimizer=w["optimizer"],
                 optimizer_params={"lr": 0.0005}, batches_count=nb, seed=0, sharded=N > 1)
    tr.set_training_set(X, B)
    spec = lambda i: ((i % nb) * B, B, i // nb + 1, i % nb + 1)  # noqa: E731

    def run(n, i0):
        for i in range(i0, i0 + n):
            s = spec(i)
            tr.step(s[0], s[1], epoch=s[2], batch=s[3], prefetch=[spec(i + 1), spec(i + 2)])
        return i0 + n

    i = run(20, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i = run(a.steps, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    tr.enable_stage_timing(16)
    i = run(16, i)
    st = {k: round(float(np.mean(v)), 4) for k, v in tr.stage_times_ms().items()}
    print("N=%d  k_local=%d  B_global=%d  ms/step %.3f  -> %.0f M triples/s aggregate if the all-reduce were free  stages %s"
          % (N, e.shape[1], B, dt * 1e3, N and B * 21 / dt / 1e6, st), flush=True)
    del tr
    torch.cuda.empty_cache()
