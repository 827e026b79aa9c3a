"""Kernel-level timing of the half-precision MFMA prefilter (exact-fast mode) at the C4 shape: ComplEx k=200 (k_int=400),
|E| = 1M, 8192 query rows.  Prints ms per launch for random positives (many undecided pairs) and for positives that
rank near the top (few), with torch events on the launch stream."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import device as D  # noqa: E402
from emgraph_amd import _lib as L  # noqa: E402
from emgraph_amd.evaluation import ranking as RK  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
n_ent, k_int, nq = 1_000_000, 400, 4096
E = torch.randn((n_ent, k_int), generator=g, device=dev) * 0.1
R = torch.randn((1000, k_int), generator=g, device=dev) * 0.1
rs = np.random.RandomState(0)
T = torch.from_numpy(np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 1000, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)).to(dev)
Q, pos_int = D.eval_build_queries(L.COMPLEX, E, R, k_int, 1.0, T, L.EVAL_SPO)
ld = D.prefilter_ld(k_int)
Eh = D.to_f16(E, k_int, ld_dst=ld)
Qh = D.to_f16(Q, k_int, ld_dst=ld)
bounds = RK.table_norm_bounds(E, Eh, k_int)
band = RK.prefilter_band(Q, Qh, k_int, bounds)
n_rows = Q.shape[0]
n_seg = D.eval_prefilter_segments(n_rows, n_ent)
pairs, pcount = RK._pair_buffer(dev, n_seg)
for label, p in (("random positives", pos_int), ("top positives", torch.full_like(pos_int, 300000))):
    best = 1e9
    for _ in range(4):
        cnt = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        D.eval_prefilter_f16(L.COMPLEX, Qh, p, band, Eh, 0, k_int, 1.0, cnt, pairs, pcount)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-18s %.3f ms  pairs %d  overflow %d  checksum %d" % (label, best, int(pcount[:n_seg].sum()), int(pcount[n_seg]), int(cnt.sum())))
