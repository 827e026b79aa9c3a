#!/usr/bin/env python3
"""per-wave wall-clock stamps of the apply kernel's (or, with a second argument `fused`, the fused kernel's) last launch
(library variant built with -DEMG_TRACE: tools/build_variant.sh trace emg_apply "-DEMG_TRACE=1" emg_fused_m3 "-DEMG_TRACE=1"):
usage: EMGRAPH_HIP_LIB=.../libemgraph_hip_trace.so python tools/trace_waves.py [C2] [fused]"""
import argparse, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from emgraph_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
os.environ["EMG_GRAPH"] = "0"
args = argparse.Namespace(no_fused=False, no_inplace=False, no_pipeline=False)
r = bench.StepRunner(name, args, 0, 1)
r.run(20); r.sync()
lib = ctypes.CDLL(L.LIB_PATH)
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"
(lib.emg_trace_clear_fused if fused else lib.emg_trace_clear)()
r.run(1); r.sync()
buf = np.zeros(4 * 65536, np.uint64)
assert (lib.emg_trace_read_fused if fused else lib.emg_trace_read)(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.size)) == 0
t = buf.reshape(-1, 4).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
start, mid, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (np.maximum(t[:, 1], t[:, 2]) - t0) * 0.01   # us (100 MHz clock)
print("waves", len(t), "first start 0, last start %.2f us, last end %.2f us" % (start.max(), end.max()))
print("wave lifetime us: median %.2f  p90 %.2f  max %.2f   (table 0 part: median %.2f)" % (
    np.median(end - start), np.percentile(end - start, 90), (end - start).max(), np.median(mid - start)))
busy = np.zeros(int(end.max() / 2) + 2)   # waves alive per 2 us bin
for a, b in zip(start, end):
    busy[int(a / 2):int(b / 2) + 1] += 1
if not fused and (t[:, 3] > 0).any():
    it = (t[:, 3] - t[:, 0]) * 0.01
    print("entity table: items (segments) median %.2f us  p90 %.2f;  untouched rows after them median %.2f us  p90 %.2f" % (
        np.median(it), np.percentile(it, 90), np.median(mid - start - it), np.percentile(mid - start - it, 90)))
print("waves alive per 2 us:", busy.astype(int).tolist())
h, edges = np.histogram(start, bins=12)
print("starts per bin:", list(zip(np.round(edges[:-1], 1), h)))
h, edges = np.histogram(end, bins=12)
print("ends per bin:  ", list(zip(np.round(edges[:-1], 1), h)))
late = np.argsort(-end)[:8]
print("latest waves (index, start, end):", [(int(i), round(start[i], 2), round(end[i], 2)) for i in late])
