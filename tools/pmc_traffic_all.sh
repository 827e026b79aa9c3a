#!/usr/bin/env bash
# HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, counters + kernel-trace only) of every library kernel of
# several bench workloads.  usage on the GPU box: bash tools/pmc_traffic_all.sh OUT.json [workloads...]   (default: C3 C3a C3g C1 C2 C5)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
DST="${1:-gpurun_out/pmc_traffic_all.json}"; shift
WLS="${*:-C3 C3a C3g C1 C2 C5}"
OUT=gpurun_out/pmc_all; rm -rf $OUT; mkdir -p $OUT
for wl in $WLS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${wl}_$c -o p -- python3 bench.py --workload $wl --quick --steps 6 --warmup 2 --no-cpu --no-eval --no-ceilings > $OUT/${wl}_$c.log 2>&1 || echo "pass failed: $wl $c"
  done
done
python3 - "$OUT" "$DST" $WLS <<'PY'
import csv, glob, json, sys, collections, os
sys.path.insert(0, os.getcwd())
out, dst, wls = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for wl in wls:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, wl, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "emg::" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][c].append(float(r["Counter_Value"]))
        for f in glob.glob("%s/%s_%s/**/*kernel_trace.csv" % (out, wl, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "emg::" in r["Kernel_Name"]:
                    dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ks = {}
    for k, v in agg.items():
        f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        d = dur.get(k, [])
        hb = (2 * f + w) * 1024
        ks[k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "dispatches": len(v["FETCH_SIZE"]), "hbm_bytes_per_launch": int(hb),
                 "avg_us_under_pmc": round(sum(d) / max(1, len(d)), 1), "GBps_under_pmc": round(hb / max(1e-9, sum(d) / max(1, len(d)) * 1e-6) / 1e9, 1)}
    res[wl] = ks
json.dump({"source_hash": __import__("emgraph_amd._lib", fromlist=["x"]).load().emg_source_hash().decode(), "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python3 bench.py --workload W --quick "
           "--steps 6 --warmup 2 --no-cpu --no-eval --no-ceilings` per workload W.  KB per dispatch averaged over the dispatches of a kernel "
           "(a kernel launched for two tables — the catch-up — averages both).  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is "
           "doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced 16 B/lane read).", "workloads": res}, open(dst, "w"), indent=1)
for wl in wls:
    for k, v in sorted(res[wl].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]:
        print(wl, k[:80], v)
PY
rm -rf $OUT
