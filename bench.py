#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its own configuration, measured on MI355X.

metric : "positive+negative triples scored/sec at k=200, eta=20; filtered ranks/sec"
workload (N=1 default) : C3 = ComplEx k=200 (k_int=400) eta=20 on synthetic |E|=1M |R|=1k
          (BASELINE.json configs[2]; the config the metric's k=200/eta=20 is quoted on and which
          fits one GPU).  A "step" = one full pass of the hot path over one batch of B positives:
          Philox corruptions -> fused gather+score of B*(1+eta) triples -> NLL loss + dL/dscore
          -> fused backward -> deterministic row-sparse SGD apply (the reference forces SGD for
          |E| > 5e5, EmbeddingModel.py:1267-1274).  value = B*(1+eta)*steps*N / time.
          The second half of the metric (filtered ranks/sec, config C4) is reported in "eval".

Contract: W untimed warm-up steps, exactly K timed steps bracketed by barrier + synchronize,
max over ranks, rank 0 prints ONE JSON line.  Inputs are resident in HBM before timing starts.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3     # f32-input MFMA == f32 vector peak
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA (not the 2:1-sparse headline)

WORKLOADS = {
    # name: model, k, eta, n_ent, n_rel, B, loss, optimizer
    "C3": dict(model="ComplEx", k=200, eta=20, n_ent=1_000_000, n_rel=1000, B=16384, loss="nll", optimizer="sgd",
               desc="ComplEx k=200 (k_int=400) eta=20, synthetic |E|=1M |R|=1k, B=16384/GPU, NLL, SGD"),
    "C3p": dict(model="TransE", k=200, eta=20, n_ent=1_000_000, n_rel=1000, B=16384, loss="nll", optimizer="sgd",
                desc="TransE-L1 k=200 eta=20 on the |E|=1M table (HBM-roofline target run)"),
    "C2": dict(model="DistMult", k=200, eta=10, n_ent=14541, n_rel=237, B=2722, loss="nll", optimizer="adam",
               desc="DistMult k=200 eta=10 NLL, FB15k-237-shaped, B=2722"),
    "C1": dict(model="TransE", k=100, eta=20, n_ent=38600, n_rel=11, B=1725, loss="pairwise", optimizer="adam",
               desc="TransE-L1 k=100 eta=20 pairwise, WN11-shaped, B=1725"),
}
MODEL_IDS = {"TransE": 0, "TransE_L2": 1, "DistMult": 2, "ComplEx": 3, "HolE": 4}


def glorot(rs, rows, cols):
    lim = math.sqrt(6.0 / (rows + cols))
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)


def algorithmic_bytes(stage, B, eta, k_int, n_unique_ent=None, n_unique_rel=None, n_single=0):
    """ALGORITHMIC HBM bytes of one launch (DESIGN.md 'bytes per unit'); int32 ids, fp32 rows."""
    row = 4 * k_int
    if stage == "forward":   # spo + codes + (3+eta) rows + (1+eta) scores
        return B * (12 + 4 * eta + (3 + eta) * row + 4 * (1 + eta))
    if stage == "backward":  # spo + codes + g + (3+eta) rows read + (3+eta) rows written
        return B * (12 + 4 * eta + 4 * (1 + eta) + (3 + eta) * row + (3 + eta) * row)
    if stage == "fused":     # spo + codes + singleton flags + (3+eta) rows read + (3+eta) rows written
        # (a singleton row is written in place, any other row to the contribution buffer: one row each)
        return B * (12 + 4 * eta + (2 + eta) + (3 + eta) * row + (3 + eta) * row)
    if stage == "apply_ent":  # non-singleton contribution rows read once + RMW of each such destination
        n_ns = (2 + eta) * B - n_single
        return n_ns * row + 2 * (n_unique_ent - n_single) * row + (2 + eta) * B * 8
    if stage == "apply_rel":
        return B * row + 2 * n_unique_rel * row + B * 8
    return 0


def pmc_traffic(stage, args, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r1_d_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    doubled per MI355X_MICROARCH.md).  Only valid for the configuration it was collected on; else null."""
    if stage != "fused" or args.workload != "C3" or args.batch or world != 1 or args.no_inplace or args.no_fused:
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r1_d_pmc_traffic.json")))
        k = [v for n, v in d["kernels"].items() if "train_backward_kernel<3, 4, 1, 64, true, 1>" in n]
        return k[0]["hbm_bytes_per_launch"] if k else None
    except (OSError, ValueError, KeyError):
        return None


def run_train(args, rank, world):
    import torch

    from emgraph_amd import device as D
    from emgraph_amd.training import Trainer

    w = WORKLOADS[args.workload]
    cplx = w["model"] in ("ComplEx", "HolE")
    k_int = 2 * w["k"] if cplx else w["k"]
    scale = float(np.float32(2 / w["k"])) if w["model"] == "HolE" else 1.0
    from emgraph_amd import parallel
    # weak scaling: the per-GPU batch is fixed, the GLOBAL batch grows with N.  With k-sharding every rank
    # walks all B_global groups but only its 1/N column slab of each row => per-GPU bytes stay constant.
    B0, eta = (args.batch or w["B"]), w["eta"]
    real_world = world
    sim = max(1, args.simulate_ranks)   # profiling aid: ONE GPU runs rank 0's share of an N-rank job (no collective)
    if sim > 1:
        world, rank = sim, 0
    B = B0 * world
    steps, warm = args.steps, args.warmup
    rs = np.random.RandomState(0)  # init seed 0 (constants.py:52)
    ent0 = glorot(rs, w["n_ent"], k_int)
    rel0 = glorot(rs, w["n_rel"], k_int)
    drs = np.random.RandomState(1234)  # the SAME triples on every rank
    n_tr = (steps + warm) * B
    X = np.stack([drs.randint(0, w["n_ent"], n_tr), drs.randint(0, w["n_rel"], n_tr),
                  drs.randint(0, w["n_ent"], n_tr)], 1).astype(np.int32)
    k_full = k_int
    ent_l, rel_l = ent0, rel0
    if world > 1:
        ent_l = parallel.shard_columns(ent0, rank, world, cplx)
        rel_l = parallel.shard_columns(rel0, rank, world, cplx)
        k_int = ent_l.shape[1]
    tr = Trainer(MODEL_IDS[w["model"]], k_int, scale, ent_l, rel_l, eta, loss=w["loss"], optimizer=w["optimizer"],
                 optimizer_params={"lr": 0.0005}, batches_count=steps + warm, seed=0, fused=not args.no_fused,
                 inplace=not args.no_inplace, pipeline=not args.no_pipeline, sharded=world > 1)
    tr.set_training_set(X, B)
    nxt = lambda i: [((j) * B, B, 1, j + 1) for j in (i + 1, i + 2, i + 3) if j < warm + steps]  # noqa: E731  (batches ahead)
    for i in range(warm):
        tr.step(i * B, B, epoch=1, batch=i + 1, prefetch=nxt(i))
    torch.cuda.synchronize()
    if real_world > 1:
        import torch.distributed as dist
        dist.barrier()
    tr.enable_stage_timing()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        tr.step(i * B, B, epoch=1, batch=i + 1, prefetch=nxt(i))
    t_issue = time.perf_counter() - t0  # host time to enqueue the timed steps (host-bound if ~= dt)
    torch.cuda.synchronize()
    if real_world > 1:
        import torch.distributed as dist
        dist.barrier()
    dt = time.perf_counter() - t0
    if real_world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = tr.read_loss()
    assert math.isfinite(loss), "loss is not finite"
    stage_ms = {k: float(np.mean(v)) for k, v in tr.stage_times_ms().items()}
    # unique touched rows of the last timed batch (for the apply kernel's algorithmic bytes)
    n_ce = (2 + eta) * B
    sl = tr.slots[0]  # any slot: both hold a full prepared batch of the same shape
    n_ue = int(torch.unique(sl["dest_ent"][:n_ce]).numel())
    n_ur = int(torch.unique(sl["dest_rel"][:B]).numel())
    n_single = int(sl["single"][:n_ce].sum().item()) if tr.inplace else 0
    stages = {}
    for name, ms in stage_ms.items():
        ab = algorithmic_bytes(name, B, eta, k_int, n_ue, n_ur, n_single)
        stages[name] = {"ms": round(ms, 4), "alg_bytes": ab, "GBps": round(ab / (ms * 1e-3) / 1e9, 1) if ab else None}
    stages["_batch"] = {"unique_ent_rows": n_ue, "singleton_slots": n_single, "unique_rel_rows": n_ur}
    dom = max((s for s in stages if stages[s].get("alg_bytes")), key=lambda s: stages[s]["ms"])
    ach = stages[dom]["GBps"]
    # on-box streaming-copy rate (1 GiB read + 1 GiB written per copy): the practical HBM ceiling next to the
    # 8 TB/s datasheet peak the fraction below is quoted against (SURVEY 8d asks for both denominators)
    src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ce0.record()
    for _ in range(5):
        dst.copy_(src)
    ce1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 4 / (ce0.elapsed_time(ce1) * 1e-3) / 1e9
    del src, dst
    roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "copy_rate_measured": round(copy_gbs, 1),
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args, world),
                "alg_bytes_per_launch": stages[dom]["alg_bytes"], "avg_launch_ms": stages[dom]["ms"]}
    return dict(dt=dt, t_issue=t_issue, fused=tr.fused, sim=sim, B=B, B0=B0, eta=eta, k_int=k_full, k_local=k_int, stages=stages, roofline=roofline, loss=loss,
                w=w, tr=tr, ent0=ent0, rel0=rel0, X=X, scale=scale)


def run_eval(res, args):
    """C4: filtered 1-vs-all ranks/sec on the same tables ('s+o', worst), exact f32 MFMA path."""
    import torch

    from emgraph_amd import parallel
    from emgraph_amd.evaluation import FilterIndex, rank_triples_device
    from emgraph_amd.training import alloc_table
    w = res["w"]
    rank, world = parallel.rank_world()
    n_test = args.eval_triples
    rs = np.random.RandomState(99)
    T = res["X"][rs.choice(len(res["X"]), n_test, replace=False)]
    F = FilterIndex(res["X"])  # one-off index of the filter triples (not timed: built once per evaluation run)
    mid = MODEL_IDS[w["model"]]
    # trained-scale tables (N(0, 0.1)): the Glorot start values are ~1e-3, whose scores all truncate to the
    # same int32(score*1e5) — legal but unrepresentative of a ranking workload
    ers = np.random.RandomState(7)
    dev = torch.device("cuda")
    ent = alloc_table(w["n_ent"], res["k_int"], dev, init=(ers.randn(w["n_ent"], res["k_int"]) * 0.1).astype(np.float32))
    rel = alloc_table(w["n_rel"], res["k_int"], dev, init=(ers.randn(w["n_rel"], res["k_int"]) * 0.1).astype(np.float32))
    shard = (rank, world) if world > 1 else None
    rank_triples_device(mid, ent, rel, res["k_int"], res["scale"], T[:64], "s+o", "worst", filter_triples=F, shard=shard)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    st = {}
    t0 = time.perf_counter()
    ranks = rank_triples_device(mid, ent, rel, res["k_int"], res["scale"], T, "s+o", "worst", filter_triples=F, shard=shard,
                                stats=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_ranks = 2 * n_test  # one rank = one (test triple, side)
    flops = 2.0 * res["k_int"] * w["n_ent"] * n_ranks
    cplx = w["model"] in ("ComplEx", "HolE", "DistMult")
    out = {"metric": "filtered ranks/sec", "value": round(n_ranks / dt, 1), "unit": "ranks/s", "test_triples": n_test,
           "corrupt_side": "s+o", "precision": "f32 (exact, v_mfma_f32_32x32x2_f32)" if cplx else "f32 VALU",
           "seconds": round(dt, 4), "mean_rank": float(np.mean(ranks))}
    if cplx:
        kflops = flops / world  # each rank's count kernels cover its candidate range
        kt = st["count_ms"] * 1e-3
        out["roofline"] = {"bound": "mfma", "kernel": "count_mfma_pipe_kernel", "achieved": round(kflops / kt / 1e12, 2),
                           "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(kflops / kt / 1e12 / MFMA_F32_PEAK_TF, 4),
                           "launches": st["count_launches"], "kernel_ms": round(st["count_ms"], 3),
                           "end_to_end_TFLOPs": round(flops / dt / 1e12, 2),
                           "note": "kernel time from HIP events around the count launches; end-to-end adds query build, "
                                   "filter CSR + H2D, filter kernel, D2H; f32-input MFMA peak"}
        # bf16 MFMA throughput mode (statistical rank agreement, see emg_rank_bf16.hip); the bf16 copy of the
        # table is made once per evaluation run, like the filter index
        from emgraph_amd import device as D
        eb = D.to_bf16(ent, res["k_int"], ld_dst=D.bf16_ld(res["k_int"]))
        kw = dict(filter_triples=F, shard=shard, precision=1, ent_bf16=eb)
        rank_triples_device(mid, ent, rel, res["k_int"], res["scale"], T[:64], "s+o", "worst", **kw)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        stb = {}
        t0 = time.perf_counter()
        rb = rank_triples_device(mid, ent, rel, res["k_int"], res["scale"], T, "s+o", "worst", stats=stb, **kw)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        ktb = stb["count_ms"] * 1e-3
        out["bf16"] = {"value": round(n_ranks / dtb, 1), "unit": "ranks/s", "seconds": round(dtb, 4),
                       "precision": "bf16 operands, f32 accumulate (v_mfma_f32_32x32x16_bf16)",
                       "median_rel_rank_error_vs_exact": float(np.median(np.abs(rb - ranks) / (2.0 * w["n_ent"]))),
                       "roofline": {"bound": "mfma", "kernel": "count_mfma_bf16_v3_kernel",
                                    "achieved": round(kflops / ktb / 1e12, 2), "peak": MFMA_BF16_PEAK_TF,
                                    "unit": "TFLOP/s", "frac": round(kflops / ktb / 1e12 / MFMA_BF16_PEAK_TF, 4),
                                    "launches": stb["count_launches"], "kernel_ms": round(stb["count_ms"], 3),
                                    "end_to_end_TFLOPs": round(flops / dtb / 1e12, 2)}}
    return out


def cpu_baseline(res, args):
    """The oracle's fused C port (OpenMP, all host cores) on a bounded sample of the same workload:
    forward scoring of B*(1+eta) triples per batch.  A reported baseline, not the target."""
    from oracle import c_oracle as co
    w = res["w"]
    B, eta, k_int = res["B"], res["eta"], res["k_int"]
    mid = MODEL_IDS[w["model"]]
    X = res["X"]
    nb = max(1, min(args.cpu_batches, len(X) // B))
    co.lib()
    codes = [co.corrupt_codes(B, eta, 2, w["n_ent"], 0, i) for i in range(nb)]
    co.train_forward(mid, res["ent0"], res["rel0"], k_int, res["scale"], X[:256], eta, codes[0][:256 * eta])  # warm
    t0 = time.perf_counter()
    done = 0
    while True:  # cycle over the sample's batches until ~args.cpu_seconds of CPU work has been timed
        for i in range(nb):
            co.train_forward(mid, res["ent0"], res["rel0"], k_int, res["scale"], X[i * B:(i + 1) * B], eta, codes[i])
        done += nb
        if time.perf_counter() - t0 >= args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    out = {"value": round(done * B * (1 + eta) / dt, 1), "unit": "triples scored/s", "cores": co.num_threads(),
           "kind": "port", "sample": "%d passes over %d batches of B=%d (%d triples scored), fused C/OpenMP "
           "gather+score (forward only), same tables" % (done // nb, nb, B, done * B * (1 + eta)),
           "seconds": round(dt, 3)}
    # second figure: the literal op-by-op numpy restatement (3 materialised gathers + elementwise passes, one
    # thread) — the closest stand-in for the reference's TF-eager-CPU graph; bounded to ~2e5 triples
    from oracle import emgraph_oracle as orc
    Bs = min(B, 8192)
    xp = X[:Bs]
    xn = orc.generate_corruptions_for_fit_philox(xp, eta=eta, corrupt_side="s,o", entities_size=w["n_ent"], seed=0, counter=0)
    kk = w["k"]
    t0 = time.perf_counter()
    orc.score_triples(w["model"], res["ent0"], res["rel0"], xp, k=kk)
    orc.score_triples(w["model"], res["ent0"], res["rel0"], xn, k=kk)
    dtn = time.perf_counter() - t0
    out["numpy_unfused_1thread"] = {"value": round(Bs * (1 + eta) / dtn, 1), "unit": "triples scored/s",
                                    "sample": "%d triples scored" % (Bs * (1 + eta)), "seconds": round(dtn, 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--eval-triples", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work timed for cpu_baseline")
    ap.add_argument("--simulate-ranks", type=int, default=1,
                    help="profiling aid: run rank 0's share (column slab, N-fold batch) of an N-rank job on one GPU; "
                         "the all-reduce is a no-op, results are not a valid bench line")
    ap.add_argument("--cpu-batches", type=int, default=4)
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="A/B: separate forward / loss / backward kernels")
    ap.add_argument("--no-inplace", action="store_true", help="A/B: every gradient row through the contribution buffer")
    ap.add_argument("--no-pipeline", action="store_true", help="A/B: batch preparation on the compute stream")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # EMG_BENCH_ONE_DEVICE: smoke-testing the N>1 code path on a 1-GPU box (all ranks on cuda:0, gloo)
    torch.cuda.set_device(0 if os.environ.get("EMG_BENCH_ONE_DEVICE") else local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if os.environ.get("EMG_BENCH_ONE_DEVICE") else "nccl")
    res = run_train(args, rank, world)
    n = world
    w = res["w"]
    triples = res["B"] * (1 + res["eta"]) * args.steps  # B is the GLOBAL batch
    line = {
        "metric": "positive+negative triples scored/sec at k=200, eta=20; filtered ranks/sec",
        "value": round(triples / res["dt"], 1), "unit": "triples scored/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(res["dt"] / args.steps * 1e3, 4), "host_issue_ms_per_step": round(res["t_issue"] / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload + ": " + w["desc"], "step": ("corrupt+group | fused score+loss+grad (+in-place singleton SGD) | segmented apply" if res["fused"] else
                            "corrupt+group | partial scores | all-reduce | loss | backward (+in-place singleton SGD) | segmented apply"),
                   "B_per_gpu": res["B0"], "global_batch": res["B"], "eta": res["eta"], "k_int": res["k_int"],
                   "k_int_per_gpu": res["k_local"], "n_ent": w["n_ent"], "n_rel": w["n_rel"],
                   "parallelism": ("k-sharded x%d: all-reduce of partial scores only" % n) if n > 1 else "single"},
        "roofline": res["roofline"], "stages": res["stages"], "loss_sum": res["loss"],
    }
    if res["sim"] > 1:
        line["simulated_ranks"] = res["sim"]
        line["note"] = "PROFILING AID, not a bench line: one GPU ran rank 0's share of a %d-rank job without the collective" % res["sim"]
    if not args.no_eval and res["sim"] == 1:
        ev = run_eval(res, args)  # every rank takes part (range-sharded candidates + counter all-reduce)
        line["eval"] = ev
    if rank == 0:
        if not args.no_cpu and world == 1:  # CPU baseline: rank 0, N=1 only
            line["cpu_baseline"] = cpu_baseline(res, args)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
