"""What does the batch-preparation chain (Philox codes, radix sort, flags; side streams) cost the compute kernels it
runs beside?  Steps the same three batches over and over with EMG_PLAN_KEEP=1 (prepared slots stay valid: no
preparation after the first round) and compares with the normal rolling schedule.  GPU box: python tools/prep_contention.py"""
import os
import subprocess
import sys
import time

if len(sys.argv) == 1:
    for keep in ("0", "1", "0", "1"):
        env = dict(os.environ)
        if keep == "1":
            env["EMG_PLAN_KEEP"] = "1"
        subprocess.run([sys.executable, __file__, keep], env=env, check=True)
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402

keep = sys.argv[1] == "1"
args = argparse.Namespace(no_fused=False, no_inplace=False, no_pipeline=False)
r = bench.StepRunner("C3", args, 0, 1)
specs = [r.spec(i) for i in range(3)] if keep else None


def run(n):
    for i in range(n):
        if keep:
            s = specs[i % 3]
            r.tr.step(s[0], s[1], epoch=s[2], batch=s[3], prefetch=[specs[(i + 1) % 3], specs[(i + 2) % 3]])
        else:
            r.run(1)


run(60)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(1500)
torch.cuda.synchronize()
print("keep=%d  ms/step %.4f" % (keep, (time.perf_counter() - t0) / 1500 * 1e3))
