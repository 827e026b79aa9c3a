#!/usr/bin/env bash
# Kernel timeline of the C3 step: per-kernel durations and the idle gaps between consecutive kernels of a step.
# GPU box: bash tools/trace_gaps.sh
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_gaps; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --quick --no-eval --no-cpu --steps 300 --warmup 30 "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# steps = intervals between consecutive train_backward launches; take the middle 200
fb = [i for i, r in enumerate(rows) if "train_backward" in r[2]]
fb = fb[len(fb) // 4: 3 * len(fb) // 4]
period = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(fb, fb[1:])]
print("steps analysed %d   mean period %.1f us  (min %.1f max %.1f)" % (len(period), sum(period) / len(period), min(period), max(period)))
agg = collections.defaultdict(list)
for a, b in zip(fb, fb[1:]):
    t0 = rows[a][0]
    for r in rows[a:b]:
        agg[r[2] + " q" + r[3]].append(((r[0] - t0) / 1e3, (r[1] - t0) / 1e3))
print("%-60s %6s %9s %9s %9s" % ("kernel (queue)", "n/step", "start us", "end us", "dur us"))
for k, v in sorted(agg.items(), key=lambda kv: sum(x[0] for x in kv[1]) / len(kv[1])):
    n = len(v) / len(period)
    print("%-60s %6.2f %9.1f %9.1f %9.1f" % (k, n, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), sum(x[1] - x[0] for x in v) / len(v)))
PY
rm -rf $OUT/*/
