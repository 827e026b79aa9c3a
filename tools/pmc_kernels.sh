#!/usr/bin/env bash
# PMC passes (counters + kernel-trace only, one set per run) over the kernels of one bench workload whose names match a regex.
# usage on the GPU box: PMC_FILTER='apply_segments|deferred_catchup|train_fused' bash tools/pmc_kernels.sh TAG [bench args]
#   -> gpurun_out/TAG_pmc.txt (per kernel: launches, avg us under PMC, every counter's average per launch)
TAG="$1"; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --quick --steps 6 --warmup 2 --no-cpu --no-eval --no-ceilings $*"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - "$OUT" "${PMC_FILTER:-emg::}" > gpurun_out/${TAG}_pmc.txt <<'PY'
import csv, glob, re, sys, collections
out, flt = sys.argv[1], re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
short = lambda kn: kn.split("(")[0].replace("void ", "")[:90]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = short(r["Kernel_Name"])
        if flt.search(kn):
            agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = short(r["Kernel_Name"])
        if kn in agg:
            dur[kn].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    d = dur[k]
    print(k, "launches", len(d), "avg us under PMC %.1f" % (sum(d) / max(1, len(d))))
    for c, v in sorted(agg[k].items()):
        print("   %-36s avg %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
    a = agg[k]
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        f, w = sum(a["FETCH_SIZE"]) / len(a["FETCH_SIZE"]), sum(a["WRITE_SIZE"]) / len(a["WRITE_SIZE"])
        print("   hbm_bytes_per_launch (2*FETCH+WRITE)*1024 = %.4g" % ((2 * f + w) * 1024))
PY
rm -rf $OUT/p*/
cat gpurun_out/${TAG}_pmc.txt | head -150
