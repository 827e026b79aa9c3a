import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import device as D, _lib as L
from emgraph_amd.evaluation import rank_triples_device
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
n_ent, k, nq = 1_000_000, 200, 512
E = torch.randn((n_ent, k), generator=g, device=dev) * 0.1
R = torch.randn((1000, k), generator=g, device=dev) * 0.1
rs = np.random.RandomState(0)
T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 1000, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)
for mid, name in ((L.TRANSE_L1, "L1"), (L.TRANSE_L2, "L2")):
    st = {}
    r0 = rank_triples_device(mid, E, R, k, 1.0, T, "s+o", "worst", stats=st)
    torch.cuda.synchronize()
    st = {}
    t0 = time.perf_counter()
    r = rank_triples_device(mid, E, R, k, 1.0, T, "s+o", "worst", stats=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(name, "ranks/s %.0f" % (2 * nq / dt), "kernel_ms %.3f" % st["count_ms"], "checksum", int(r.sum()))
