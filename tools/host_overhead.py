"""Host-side cost of the calls one training step makes (enqueue time only, C3 shapes)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import _lib as L, device as D
from emgraph_amd.training import Trainer

n_ent, n_rel, k_int, B, eta = 1_000_000, 1000, 400, 16384, 20
rs = np.random.RandomState(0)
E = (rs.randn(n_ent, k_int) * 0.01).astype(np.float32); R = (rs.randn(n_rel, k_int) * 0.01).astype(np.float32)
X = np.stack([rs.randint(0, n_ent, 40 * B), rs.randint(0, n_rel, 40 * B), rs.randint(0, n_ent, 40 * B)], 1).astype(np.int32)
tr = Trainer(L.COMPLEX, k_int, 1.0, E, R, eta, loss="nll", optimizer="sgd", optimizer_params={"lr": 5e-4}, batches_count=40)
tr.set_training_set(X, B)
for i in range(5):
    tr.step(i * B, B, 1, i + 1)
torch.cuda.synchronize()

def timeit(name, fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print("%-28s %7.1f us" % (name, dt * 1e6))

sl = tr.slots[0]
pos = tr.X[:B]
codes = sl["codes"][:B * eta]
n_ce = (2 + eta) * B
timeit("prepare_batch call", lambda: D.prepare_batch(pos, eta, tr.sides, n_ent, codes, sl["dest_ent"][:n_ce], sl["dest_rel"][:B],
                                                     n_ent, n_rel, sl["ws_ent"], sl["ws_rel"], seed=0, counter0=0,
                                                     single_flags=sl["single"][:n_ce]), 50)
ce, cr = tr.contrib_ent[:n_ce], tr.contrib_rel[:B]
hyper = tr._hyper(5e-4) if tr.step_count else None
tr.step_count = max(tr.step_count, 1)
hyper = tr._hyper(5e-4)
timeit("train_backward_ex call", lambda: D.train_backward_ex(L.COMPLEX, tr.ent, tr.rel, k_int, 1.0, pos, eta, codes, ce, cr,
                                                             fused_loss=tr.loss_id, margin=1.0, loss_accum=tr.loss_accum[0:1],
                                                             single_ent=sl["single"][:n_ce], opt_id=tr.opt_id, step=1, hyper=hyper,
                                                             ent_state0=None, ent_state1=None, tag_ent=tr.tag_ent), 50)
timeit("apply_grouped(ent) call", lambda: D.apply_grouped(tr.opt_id, tr.ent, k_int, None, None, tr.tag_ent, 1, ce, n_ce, True, hyper, sl["ws_ent"]), 50)
timeit("apply_grouped(rel) call", lambda: D.apply_grouped(tr.opt_id, tr.rel, k_int, None, None, tr.tag_rel, 1, cr, B, False, hyper, sl["ws_rel"]), 50)
ev = torch.cuda.Event()
st = torch.cuda.Stream()
timeit("event.record", lambda: ev.record())
timeit("stream.wait_event", lambda: st.wait_event(ev))
def ctx():
    with torch.cuda.stream(st):
        pass
timeit("with torch.cuda.stream", ctx)
timeit("tensor slice X[a:b]", lambda: tr.X[5:5 + B])
timeit("current_stream()", lambda: torch.cuda.current_stream())
i = [5]
def full():
    tr.step((i[0] % 30) * B, B, 1, i[0] % 30 + 1, prefetch=[(((i[0] + j) % 30) * B, B, 1, (i[0] + j) % 30 + 1) for j in (1, 2)])
    i[0] += 1
timeit("Trainer.step (whole)", full, 100)
