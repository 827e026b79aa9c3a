#!/usr/bin/env python3
"""precision 0 / 1 / 2 filtered ranking at C4 size (ComplEx k=200, |E|=1M, 's+o'): random positives (mean rank ~ |E|/2, the
worst case for the prefilter band) and planted positives (objects aligned with their query: ranks in the top ~0.1 %)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import device as D
from emgraph_amd.evaluation import FilterIndex, PrefilterTables, rank_triples_device
from emgraph_amd.training import alloc_table

n_ent, n_rel, n_test = 1_000_000, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ki = 2 * k
rs = np.random.RandomState(7)
E = (rs.randn(n_ent, ki) * 0.1).astype(np.float32)
R = (rs.randn(n_rel, ki) * 0.1).astype(np.float32)
X = np.stack([rs.randint(0, n_ent, 200000), rs.randint(0, n_rel, 200000), rs.randint(0, n_ent, 200000)], 1).astype(np.int32)
T = X[:n_test]
F = FilterIndex(X)
dev = torch.device("cuda")


def run(Et, Rt, T, label):
    ef16 = PrefilterTables(Et, ki)
    eb16 = D.to_bf16(Et, ki, ld_dst=D.bf16_ld(ki))
    res = {}
    for prec, kw in ((0, {}), (1, dict(ent_bf16=eb16)), (2, dict(ent_f16=ef16))):
        rank_triples_device(3, Et, Rt, ki, 1.0, T[:192], "s+o", "worst", filter_triples=F, precision=prec, **kw)   # (> 128 query rows: the timed call's kernels)
        torch.cuda.synchronize()
        st = {}
        t0 = time.perf_counter()
        r = rank_triples_device(3, Et, Rt, ki, 1.0, T, "s+o", "worst", filter_triples=F, precision=prec, stats=st, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[prec] = r
        print(label, "precision", prec, "ranks/s %.0f" % (2 * len(T) / dt), "seconds %.4f" % dt, "mean rank %.1f" % r.mean(),
              {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in st.items()})
    assert np.array_equal(res[0], res[2]), "precision 2 != precision 0"
    print(label, "precision 2 == precision 0: OK; bf16 median |d rank| / |E| = %.2e" % np.median(np.abs(res[1] - res[0]) / (2 * n_ent)))


Et, Rt = alloc_table(n_ent, ki, dev, init=E), alloc_table(n_rel, ki, dev, init=R)
run(Et, Rt, T, "random ")
# planted: e_o <- sqrt(1-b^2) e_o + b |e_o| q^ with q the object-side query vector of (s, p): score ~ b |q| |e| ~ 3.3 sigma
Q, _ = D.eval_build_queries(3, Et, Rt, ki, 1.0, torch.from_numpy(T).to(dev), 1)
b = 0.15
o = torch.from_numpy(T[:, 2].astype(np.int64)).to(dev)
eo = Et[o]
qh = Q[:, :ki] / Q[:, :ki].norm(dim=1, keepdim=True)
Et[o] = (1 - b * b) ** 0.5 * eo + b * eo.norm(dim=1, keepdim=True) * qh
run(Et, Rt, T, "planted")
