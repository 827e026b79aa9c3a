"""Kernel-level timing of the bf16 1-vs-all count kernels at the headline shape (|E|=1M, k_int=400).

    python tools/bench_bf16_count.py [--rows 8192] [--ent 1000000] [--k 400] [--reps 3] [--v1]

Times emg_eval_count_bf16 with torch.cuda events on the current stream (the kernels launch on it), prints
ms and TFLOP/s (2 * rows * ent * k_int per launch), and checks that the query-stationary kernel and the
tile kernel (forced by a candidate list) produce identical counters.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emgraph_amd import device as D  # noqa: E402
from emgraph_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--ent", type=int, default=1000000)
    ap.add_argument("--k", type=int, default=400)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--v1", action="store_true", help="also time the v1 tile kernel")
    ap.add_argument("--f32", action="store_true", help="time the exact f32 MFMA count kernel instead")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    if a.f32:
        ld4 = (a.k + 3) // 4 * 4
        E = torch.zeros((a.ent, ld4), dtype=torch.float32, device=dev)
        E[:, :a.k] = torch.randn((a.ent, a.k), generator=g, device=dev) * 0.1
        Q = torch.zeros((a.rows, ld4), dtype=torch.float32, device=dev)
        Q[:, :a.k] = torch.randn((a.rows, a.k), generator=g, device=dev) * 0.1
        pos = torch.randint(-200000, 200000, (a.rows,), generator=g, device=dev, dtype=torch.int32)
        cnt = torch.zeros((2, a.rows), dtype=torch.int32, device=dev)
        best = 1e9
        for _ in range(a.reps):
            cnt.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            D.eval_count(L.COMPLEX, Q, pos, E, a.k, 1.0, cnt[0], cnt[1])
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("f32 exact count: %.3f ms  %.1f TFLOP/s  (checksum %d)" % (best, 2.0 * a.rows * a.ent * a.k / best * 1e-9,
                                                                         int(cnt.sum().item())))
        return
    ld = D.bf16_ld(a.k)
    E = torch.zeros((a.ent, ld), dtype=torch.bfloat16, device=dev)
    E[:, :a.k] = (torch.randn((a.ent, a.k), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    Q = torch.zeros((a.rows, ld), dtype=torch.bfloat16, device=dev)
    Q[:, :a.k] = (torch.randn((a.rows, a.k), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    pos = torch.randint(-200000, 200000, (a.rows,), generator=g, device=dev, dtype=torch.int32)
    selfe = torch.zeros(a.rows, dtype=torch.int32, device=dev)
    flop = 2.0 * a.rows * a.ent * a.k

    def run(cand, need=0):
        cnt = torch.zeros((2, a.rows), dtype=torch.int32, device=dev)
        best = 1e9
        for _ in range(a.reps):
            cnt.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            D.eval_count_bf16(L.COMPLEX, Q, pos, selfe, E, a.k, 1.0, cnt[0], cnt[1], cand=cand, need=need)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best, cnt.cpu().numpy()

    ms2, c2 = run(None)
    print("both counters (v3 register-stationary kernel; EMG_BF16_V4=2: v4): %.3f ms  %.1f TFLOP/s" % (ms2, flop / ms2 * 1e-9))
    ms3, c3 = run(None, need=1)
    print("one counter ('worst'; v4 by default, EMG_BF16_V4=0: v3): %.3f ms  %.1f TFLOP/s  (#(>=) equals gt+eq: %s)"
          % (ms3, flop / ms3 * 1e-9, bool(np.array_equal(c3[0], c2[0] + c2[1]))))
    if a.v1:
        ms1, c1 = run(torch.arange(a.ent, dtype=torch.int32, device=dev))
        print("v1 tile kernel     : %.3f ms  %.1f TFLOP/s" % (ms1, flop / ms1 * 1e-9))
        print("counters identical :", bool(np.array_equal(c1, c2)))


if __name__ == "__main__":
    main()
