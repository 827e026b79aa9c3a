#!/usr/bin/env bash
# HBM bytes per launch of the training-step kernels: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), counters only.
# usage (on the GPU box): bash tools/pmc_traffic.sh <out.json>
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic; mkdir -p $OUT
CMD="python3 bench.py --quick --steps 6 --warmup 2 --no-cpu --no-eval --no-pipeline --no-ceilings"
CMD2="python3 tools/bench_score.py"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- $CMD > $OUT/$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${c}_score -o p -- $CMD2 > $OUT/${c}_score.log 2>&1
done
python3 - "$OUT" "${1:-gpurun_out/pmc_traffic.json}" "$CMD" "$CMD2" <<'PY'
import csv, glob, json, sys, collections, os
sys.path.insert(0, os.getcwd())
out, dst, cmd, cmd2 = sys.argv[1:5]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s*/**/*counter_collection.csv" % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "emg::" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "_score" in f:   # bench_score.py launches the same kernel in two shapes: tell them apart by grid size
                    name += " [emg_train_forward: 16384 groups x 21 triples]" if int(r["Grid_Size"]) == 16384 * 64 else " [emg_score_triples: 344064 triples]"
                agg[name][c].append(float(r["Counter_Value"]))
res = {}
for k, v in agg.items():
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
    w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    res[k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "dispatches": len(v["FETCH_SIZE"]),
              "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump({"source_hash": __import__("emgraph_amd._lib", fromlist=["x"]).load().emg_source_hash().decode(), "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `%s` (C3, B=16384) and of "
           "`%s` (gather+score kernels alone on the 1M-entity table).  Values are KB per dispatch averaged over dispatches. "
           "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a "
           "wide coalesced 16 B/lane read); calibration kernel: prepare_ids_kernel reads 3*4*B = 196,608 B of triples and "
           "writes codes + destination ids = 4*B*eta + 4*(2+eta)*B + 4*B = 2,818,048 B." % (cmd, cmd2), "kernels": res},
          open(dst, "w"), indent=1)
for k, v in sorted(res.items()):
    print(k, v)
PY
