#!/usr/bin/env python3
"""step / stage times of a small-batch workload as eta, B and the optimizer vary (where does a 60 us step go?)
usage: python tools/sweep_small.py [C2|C1|C5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

base = sys.argv[1] if len(sys.argv) > 1 else "C2"
args = argparse.Namespace(no_fused=False, no_inplace=False, no_pipeline=False)
w0 = bench.WORKLOADS[base]


def run(label, **over):
    w = dict(w0); w.update(over)
    bench.WORKLOADS["X"] = w
    r = bench.StepRunner("X", args, 0, 1)
    r.run(40)
    dt, issue = r.timed(300)
    st = r.stages(16)
    ms = {k: v["ms"] for k, v in st.items() if isinstance(v, dict) and "ms" in v}
    print("%-28s ms/step %.4f  inplace %d  stages %s  %s" % (label, dt / 300 * 1e3, int(r.tr.inplace), ms, st.get("_batch")), flush=True)
    r.close()


run(base)
for eta in (1, 2, 5, 10, 20, 40):
    run("eta=%d" % eta, eta=eta)
for B in (w0["B"] // 4, w0["B"] // 2, w0["B"] * 2, w0["B"] * 4):
    run("B=%d" % B, B=B)
run("sgd", optimizer="sgd")
run("adagrad", optimizer="adagrad")
