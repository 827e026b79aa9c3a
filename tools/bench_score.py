"""Gather+score kernels alone on the |E|=1M table (north_star: >= 50 % of HBM peak on the TransE gather+score kernel).

    python tools/bench_score.py [--model TransE|DistMult|ComplEx] [--k 200] [--batch 16384] [--eta 20]

Times emg_train_forward (one positive group = 1+eta scored triples, shared rows read once) and emg_score_triples
(arbitrary triples, 3 rows each) with HIP events on the launch stream; prints algorithmic GB/s (DESIGN.md 4.1 /
SURVEY 8d byte formulas) and the fraction of the 8 TB/s peak.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import _lib as L  # noqa: E402
from emgraph_amd import device as D  # noqa: E402
from emgraph_amd.training import alloc_table  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="TransE")
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--ent", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--eta", type=int, default=20)
    a = ap.parse_args()
    mid = {"TransE": L.TRANSE_L1, "DistMult": L.DISTMULT, "ComplEx": L.COMPLEX}[a.model]
    k_int = 2 * a.k if a.model == "ComplEx" else a.k
    dev = torch.device("cuda")
    rs = np.random.RandomState(0)
    ent = alloc_table(a.ent, k_int, dev, init=(rs.randn(a.ent, k_int) * 0.01).astype(np.float32))
    rel = alloc_table(1000, k_int, dev, init=(rs.randn(1000, k_int) * 0.01).astype(np.float32))
    B, eta = a.batch, a.eta
    nb = 8  # distinct batches so that consecutive launches do not hit the same rows in cache
    pos = [torch.from_numpy(np.stack([rs.randint(0, a.ent, B), rs.randint(0, 1000, B), rs.randint(0, a.ent, B)], 1)
                            .astype(np.int32)).to(dev) for _ in range(nb)]
    codes = [D.corrupt_codes(B, eta, L.SIDE_SO, a.ent, dev, seed=0, counter=i) for i in range(nb)]
    sp = torch.empty(B, dtype=torch.float32, device=dev)
    sn = torch.empty(B * eta, dtype=torch.float32, device=dev)
    it = [0]

    def fwd():
        i = it[0] % nb
        it[0] += 1
        D.train_forward(mid, ent, rel, k_int, 1.0, pos[i], eta, codes[i], scores_pos=sp, scores_neg=sn)
    ms = timed(fwd)
    row = 4 * k_int
    alg = B * (12 + 4 * eta + (3 + eta) * row + 4 * (1 + eta))
    print("%s k_int=%d  emg_train_forward  B=%d eta=%d: %.4f ms  %.1f M triples/s  %.0f GB/s algorithmic = %.1f %% of 8 TB/s"
          % (a.model, k_int, B, eta, ms, B * (1 + eta) / ms / 1e3, alg / ms / 1e6, alg / ms / 1e6 / 80.0))
    n = B * (1 + eta)
    spo = [torch.from_numpy(np.stack([rs.randint(0, a.ent, n), rs.randint(0, 1000, n), rs.randint(0, a.ent, n)], 1)
                            .astype(np.int32)).to(dev) for _ in range(4)]
    out = torch.empty(n, dtype=torch.float32, device=dev)

    def sc():
        i = it[0] % 4
        it[0] += 1
        D.score_triples(mid, ent, rel, k_int, 1.0, spo[i], out=out)
    ms2 = timed(sc)
    alg2 = n * (12 + 3 * row + 4)
    print("%s k_int=%d  emg_score_triples  n=%d: %.4f ms  %.1f M triples/s  %.0f GB/s algorithmic = %.1f %% of 8 TB/s "
          "(relation rows come from a 1000-row table that stays in L2: HBM moves ~2/3 of these bytes)"
          % (a.model, k_int, n, ms2, n / ms2 / 1e3, alg2 / ms2 / 1e6, alg2 / ms2 / 1e6 / 80.0))


if __name__ == "__main__":
    main()
