"""TransE 1-vs-all ranking at |E| = 1M, k = 200: exact f32 kernel vs the fixed-point (v_sad_u16) exact-fast mode."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import device as D, _lib as L
from emgraph_amd.evaluation import L2Tables, SadTables, rank_triples_device
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
n_ent, k = 1_000_000, 200
nq = int(os.environ.get("NQ", "512"))
E = torch.randn((n_ent, k), generator=g, device=dev) * 0.1
R = torch.randn((1000, k), generator=g, device=dev) * 0.1
rs = np.random.RandomState(0)
T = np.stack([rs.randint(0, n_ent, nq), rs.randint(0, 1000, nq), rs.randint(0, n_ent, nq)], 1).astype(np.int32)


def run(mid, **kw):
    rank_triples_device(mid, E, R, k, 1.0, T, "s+o", "worst", **kw)
    torch.cuda.synchronize()
    st = {}
    t0 = time.perf_counter()
    r = rank_triples_device(mid, E, R, k, 1.0, T, "s+o", "worst", stats=st, **kw)
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0, st


for mid, name in ((L.TRANSE_L1, "L1"), (L.TRANSE_L2, "L2")):
    r, dt, st = run(mid)
    print(name, "exact   ranks/s %.0f" % (2 * nq / dt), "kernel_ms %.3f" % st["count_ms"], "checksum", int(r.sum()))
    if mid == L.TRANSE_L1:
        t0 = time.perf_counter()
        tabs = SadTables(E, R, k)
        torch.cuda.synchronize()
        print("   image of the table: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
        r2, dt2, st2 = run(mid, precision=2, ent_f16=tabs)
        print(name, "sad     ranks/s %.0f" % (2 * nq / dt2), "kernel_ms %.3f" % st2["count_ms"], "checksum", int(r2.sum()),
              "equal", bool(np.array_equal(r, r2)), "pairs", st2.get("pairs"), "frac %.5f" % (st2.get("pairs", 0) / (2.0 * nq * n_ent)),
              "fallback", st2.get("fallback"))
    else:
        t0 = time.perf_counter()
        tabs = L2Tables(E, k)
        tabs.bounds(0, n_ent)
        torch.cuda.synchronize()
        print("   augmented half rows of the table: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
        r2, dt2, st2 = run(mid, precision=2, ent_f16=tabs)
        print(name, "mfma    ranks/s %.0f" % (2 * nq / dt2), "kernel_ms %.3f" % st2["count_ms"], "checksum", int(r2.sum()),
              "equal", bool(np.array_equal(r, r2)), "pairs", st2.get("pairs"), "frac %.5f" % (st2.get("pairs", 0) / (2.0 * nq * n_ent)),
              "fallback", st2.get("fallback"))
