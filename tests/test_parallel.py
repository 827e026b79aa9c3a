"""Multi-process tests of the N>1 path.

CPU (gloo, world_size 2): the slab / range arithmetic and the collectives of emgraph_amd.parallel, with the
oracle standing in for the kernels (test-only), must reproduce the unsharded result.
GPU: the real k-sharded training + range-sharded evaluation (HIP kernels) with two ranks sharing cuda:0
over gloo must match a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

from tests import _dist_workers as W  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, tmp_path, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=W.run, args=(fn, r, world, port, str(tmp_path)) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join()
            raise AssertionError("distributed worker hung")
    for r in range(world):
        fail = os.path.join(tmp_path, "fail_%d" % r)
        assert not os.path.exists(fail), open(fail).read()
        assert os.path.exists(os.path.join(tmp_path, "ok_%d" % r)), "rank %d did not finish" % r


def test_slab_and_range_arithmetic():
    from emgraph_amd import parallel
    for k, world in ((200, 8), (100, 3), (7, 4), (5, 8)):
        sl = parallel.column_slabs(k, world)
        assert sl[0][0] == 0 and sl[-1][1] == k and all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in sl) - min(b - a for a, b in sl) <= 1
    rs = np.random.RandomState(0)
    for cplx in (False, True):
        T = rs.randn(6, 20).astype(np.float32)
        k = 10 if cplx else 20
        slabs = [parallel.shard_columns(T, r, 3, cplx) for r in range(3)]
        assert all(s.shape[1] % 4 == 0 for s in slabs)
        np.testing.assert_array_equal(parallel.unshard_columns(slabs, k, cplx), T)
        ts = [parallel.shard_columns(torch.from_numpy(T), r, 3, cplx).numpy() for r in range(3)]
        for a, b in zip(slabs, ts):
            np.testing.assert_array_equal(a, b)
    for n, world in ((1000000, 8), (10, 3), (2, 4)):
        rr = [parallel.entity_range(n, r, world) for r in range(world)]
        assert rr[0][0] == 0 and rr[-1][1] == n and all(rr[i][1] == rr[i + 1][0] for i in range(world - 1))


def test_kshard_partial_scores_allreduce_gloo(tmp_path):
    _spawn(W.kshard_scores_worker, 2, tmp_path)


def test_range_sharded_eval_counts_allreduce_gloo(tmp_path):
    _spawn(W.eval_counts_worker, 2, tmp_path)


@pytest.mark.parametrize("world", [2, 3])
def test_batch_sharded_exchange_is_rank_count_independent_gloo(tmp_path, world):
    _spawn(W.batch_exchange_worker, world, tmp_path)
    Ws = [np.load(os.path.join(tmp_path, "W_%d.npy" % r)) for r in range(world)]
    for w in Ws[1:]:
        np.testing.assert_array_equal(w, Ws[0])


@pytest.mark.parametrize("world", [2, 3, 5])
def test_row_exchange_device_resident_form_is_rank_count_independent_gloo(tmp_path, world):
    """parallel.RowExchange (round 4: metadata phase + data phase, fixed-capacity all-gather, no torch.unique / argsort in the
    product path): same result as one process, for 2, 3 and 5 ranks (5 ranks over 23 positives and 37 rows: uneven everything)"""
    _spawn(W.row_exchange_worker, world, tmp_path)
    Ws = [np.load(os.path.join(tmp_path, "W_%d.npy" % r)) for r in range(world)]
    for w in Ws[1:]:
        np.testing.assert_array_equal(w, Ws[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name,loss,opt,reg", [("ComplEx", "nll", "sgd", None), ("TransE_L2", "pairwise", "adagrad", None),
                                               ("HolE", "multiclass_nll", "momentum", None), ("DistMult", "nll", "adam_lazy", None),
                                               ("ComplEx", "nll", "adam", None),                       # the reference's default optimizer
                                               ("DistMult", "pairwise", "adagrad", {"lambda": 1e-3, "p": 2}),   # LP + a stateful optimizer
                                               ("TransE_L2", "nll", "adam", {"lambda": 1e-3, "p": 3})])
def test_batch_sharded_fit_two_ranks_equals_single_process(tmp_path, name, loss, opt, reg):
    """north_star's training split: batch rows split over the ranks, gradient rows sent to the owner of their
    destination, which SUMS them in global slot order; the sums are all-gathered and every replica applies the optimizer
    (replicated state) — so the reference's default optimizer (Keras Adam: every row decays every step) and the LP
    regulariser with any optimizer run under this split.  Negatives are drawn by GLOBAL row index, so two ranks must
    reproduce the single-process run with the same kernels (contribution path, no in-place singletons) BIT for bit; the
    default single-process plan (in-place singleton updates) differs from it by fp32 rounding only."""
    from emgraph_amd import models
    _spawn(W.batch_sharded_fit_worker, 2, tmp_path, name, loss, opt, reg)
    res = [np.load(os.path.join(tmp_path, "res_%d.npz" % r)) for r in range(2)]
    for key in ("E", "R", "pred"):
        np.testing.assert_array_equal(res[0][key], res[1][key])          # replicas stay identical
    assert int(res[0]["xgmi"]) > 0
    rs = np.random.RandomState(5)
    n_ent, n_rel = 80, 4
    X = np.stack([rs.randint(0, n_ent, 900), rs.randint(0, n_rel, 900), rs.randint(0, n_ent, 900)], 1)
    X[:n_ent, 0] = np.arange(n_ent)
    X[:n_rel, 1] = np.arange(n_rel)
    kw = dict(k=10, eta=3, epochs=2, batches_count=3, seed=3, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05})
    if reg is not None:
        kw.update(regularizer="LP", regularizer_params=reg)
    emp = {"norm": 2} if name == "TransE_L2" else {}
    cls = models.TransE if name == "TransE_L2" else getattr(models, name)
    import emgraph_amd.training as T
    plain = T.Trainer.__init__
    try:   # single process, every gradient row through the contribution buffer (what the sharded step does)
        def no_inplace(self, *a, **k_):
            k_["inplace"] = False
            plain(self, *a, **k_)
        T.Trainer.__init__ = no_inplace
        m0 = cls(embedding_model_params=emp, **kw)
        m0.fit(X[:803])
    finally:
        T.Trainer.__init__ = plain
    np.testing.assert_array_equal(res[0]["E"], m0.trained_model_params[0])
    np.testing.assert_array_equal(res[0]["R"], m0.trained_model_params[1])
    np.testing.assert_allclose(res[0]["losses"], m0.epoch_losses, rtol=1e-12)   # (data term: a sum over the ranks' double accumulators)
    if opt == "adam" or reg is not None:
        return   # (Keras Adam / LP: the in-place plan is a different update order for near-zero gradients; bitwise check above)
    m1 = cls(embedding_model_params=emp, **kw)
    m1.fit(X[:803])
    np.testing.assert_allclose(res[0]["E"], m1.trained_model_params[0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(res[0]["R"], m1.trained_model_params[1], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,loss,opt", [("ComplEx", "nll", "sgd"), ("TransE_L2", "pairwise", "adagrad"), ("HolE", "multiclass_nll", "momentum")])
def test_batch_sharded_fit_with_the_state_sharded_by_owner_equals_the_replicated_form(tmp_path, name, loss, opt):
    """SURVEY 8e's ZeRO-1 form of the batch plan (`shard_state`): every rank holds the optimizer state of ITS id range only, the
    owner applies the optimizer to the summed gradients of its rows and the UPDATED rows are all-gathered (emg_scatter_rows writes
    them into every replica).  Same sums, same rule, applied once: tables, predictions and losses must equal the replicated form's
    bit for bit (and thereby one GPU's); the state arrays hold half the rows; the optimizer touches about half as many rows."""
    rep, own = tmp_path / "rep", tmp_path / "own"
    rep.mkdir(); own.mkdir()
    _spawn(W.batch_sharded_fit_worker, 2, rep, name, loss, opt, None, False)
    _spawn(W.batch_sharded_fit_worker, 2, own, name, loss, opt, None, True)
    a = [np.load(os.path.join(rep, "res_%d.npz" % r)) for r in range(2)]
    b = [np.load(os.path.join(own, "res_%d.npz" % r)) for r in range(2)]
    for key in ("E", "R", "pred", "losses"):
        np.testing.assert_array_equal(b[0][key], b[1][key])     # replicas stay identical
        np.testing.assert_array_equal(b[0][key], a[0][key])     # and equal the replicated form
    assert int(b[0]["xgmi"]) == int(a[0]["xgmi"])               # the same bytes on the links
    if opt != "sgd":
        assert int(a[0]["state_rows"]) == 80 and int(b[0]["state_rows"]) == 40 and int(b[1]["state_rows"]) == 40
    assert int(b[0]["opt_rows"]) + int(b[1]["opt_rows"]) == int(a[0]["opt_rows"]) == int(a[1]["opt_rows"])


@pytest.mark.gpu
def test_shard_state_refuses_what_moves_every_row():
    """Keras Adam decays every row every step and a folded LP regulariser moves every row: their state cannot live at the owner"""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    E, R = np.zeros((8, 4), np.float32), np.zeros((2, 4), np.float32)
    for kw in (dict(optimizer="adam", sharded="batch"), dict(optimizer="adagrad", sharded=False),
               dict(optimizer="adagrad", sharded="batch", regularizer="LP", regularizer_params={"lambda": 1e-3, "p": 2})):
        with pytest.raises(ValueError, match="shard_state"):
            Trainer(L.DISTMULT, 4, 1.0, E, R, 2, shard_state=True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name,loss,opt", [("ComplEx", "nll", "adam"), ("TransE_L2", "pairwise", "sgd"),
                                           ("HolE", "multiclass_nll", "adagrad"), ("DistMult", "nll", "momentum")])
def test_sharded_fit_and_eval_two_ranks_match_single_process(tmp_path, name, loss, opt):
    from emgraph_amd import models
    from emgraph_amd.evaluation import evaluate_performance
    _spawn(W.sharded_fit_worker, 2, tmp_path, name, loss, opt)
    res = [np.load(os.path.join(tmp_path, "res_%d.npz" % r)) for r in range(2)]
    for key in ("E", "R", "ranks", "ranks_sub", "pred"):
        np.testing.assert_array_equal(res[0][key], res[1][key])  # both ranks end with the same full model
    for r in res:   # predict(sharded=True) (a collective over contiguous ranges) == the rank-local default
        np.testing.assert_array_equal(r["pred"], r["pred_local"])
    if "ranks_bf16" in res[0]:  # bf16 mode: range-sharded counters == single-rank counters, on both ranks
        for r in res:
            np.testing.assert_array_equal(r["ranks_bf16"], r["ranks_bf16_single"])
    # single-process reference run (same data / seed as the worker)
    rs = np.random.RandomState(5)
    n_ent, n_rel = 80, 4
    X = np.stack([rs.randint(0, n_ent, 900), rs.randint(0, n_rel, 900), rs.randint(0, n_ent, 900)], 1)
    X[:n_ent, 0] = np.arange(n_ent)
    X[:n_rel, 1] = np.arange(n_rel)
    kw = dict(k=10, eta=3, epochs=2, batches_count=3, seed=3, loss=loss, optimizer=opt, optimizer_params={"lr": 0.05})
    m = models.TransE(embedding_model_params={"norm": 2}, **kw) if name == "TransE_L2" else getattr(models, name)(**kw)
    m.fit(X[:800])
    # k-slab partial sums change the fp32 summation order only
    np.testing.assert_allclose(res[0]["E"], m.trained_model_params[0], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(res[0]["R"], m.trained_model_params[1], rtol=1e-3, atol=1e-5)
    # range-sharded evaluation of the SAME parameters is exact (integer counters): restore rank 0's tables
    m.trained_model_params = [res[0]["E"], res[0]["R"]]
    m._dev = None
    np.testing.assert_array_equal(evaluate_performance(X[800:], m, filter_triples=X, corrupt_side="s,o"), res[0]["ranks"])
    np.testing.assert_array_equal(evaluate_performance(X[800:840], m, filter_triples=X, corrupt_side="s+o",
                                                       entities_subset=list(range(0, 80, 3))), res[0]["ranks_sub"])


@pytest.mark.gpu
def test_sharded_exact_fast_ranking_survives_pair_buffer_overflow(tmp_path):
    """the default evaluation precision under world > 1 (it raised where the single-GPU path falls back, and a rank
    raising alone left the others in a collective): overflow on every rank -> exact kernel locally -> all-reduce"""
    _spawn(W.sharded_overflow_worker, 2, tmp_path)
    res = [np.load(os.path.join(tmp_path, "res_%d.npz" % r)) for r in range(2)]
    for r in res:
        assert int(r["fallback"]) >= 1, "the test is meant to overflow the pair buffer"
        np.testing.assert_array_equal(r["got"], r["want"])
        np.testing.assert_array_equal(r["auto"], r["want"])
        np.testing.assert_array_equal(r["ties"], r["want"])
        assert int(r["ties_taken"]) == 1 and int(r["ties_fallback"]) == 0, "the ties form of the prefilter decides this table on every rank"


@pytest.mark.gpu
def test_rccl_single_rank_collectives_run(tmp_path):
    """RCCL itself (backend "nccl" on ROCm) on this box: a one-rank process group is all a single GPU allows — two
    ranks cannot share a device under RCCL, which is why the two-rank GPU tests above stage their collectives through
    gloo.  Creates the communicator on cuda:0 and runs the three collectives the sharded paths use (all_reduce of
    scores / counters, all_gather of slabs, all_to_all_single of gradient rows) in a subprocess of its own."""
    import subprocess
    import sys
    code = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
dist.all_reduce(t)
g = [torch.empty_like(t)]
dist.all_gather(g, t)
o = torch.empty_like(t)
dist.all_to_all_single(o, t)
torch.cuda.synchronize()
assert torch.equal(g[0], t) and torch.equal(o, t) and float(t[12345]) == 12345.0
dist.destroy_process_group()
print("rccl ok")
''' % _free_port()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("sharding", ["both"])
def test_bench_runs_at_two_ranks_and_prints_both_plans(sharding):
    """`bench.py --gpus 2` as the driver starts it for a scaling run (here: both ranks on ONE device over gloo,
    EMG_BENCH_ONE_DEVICE=1): the N > 1 branch must start, time both training plans (k-slabs + score all-reduce; batch rows +
    gradient-row exchange — north_star's split) and end with ONE parseable stdout line carrying n_gpus, both plans and the bytes
    each rank puts on the links per step.  No scaling number is asserted: one device cannot give one."""
    import json
    import subprocess
    env = dict(os.environ, EMG_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-eval", "--no-cpu",
           "--no-ceilings", "--sustained-seconds", "0", "--sharding", sharding]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    # (gloo announces its connections on stdout; RCCL does not — the contract's line is the LAST one, and the only JSON one)
    assert len([ln for ln in lines if ln.startswith("{")]) == 1 and lines[-1].startswith("{") and len(lines[-1]) < 4096, out.stdout[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    want = {"k", "batch"} if sharding == "both" else {sharding}
    assert set(d["plans"]) == want and d["plan"] in want
    for name in want:
        pl = d["plans"][name]
        assert pl["value"] > 0 and pl["ms_per_step"] > 0 and pl["xgmi_bytes_per_step_per_rank"] > 0
    if "batch" in want:   # rows a rank's optimizer updates: all of the global batch's with replicated state, its own range's with sharded state
        rows = d["plans"]["batch"]["optimizer_rows_per_step_per_rank"]
        assert 0 < rows["owner_sharded_state"] < rows["replicated_state"]
    assert d["value"] == max(pl["value"] for pl in d["plans"].values())
    assert d["xgmi_bytes_per_step_per_rank"] == d["plans"][d["plan"]]["xgmi_bytes_per_step_per_rank"]
    assert d["config"]["global_batch"] == 2 * d["config"]["B_per_gpu"]
