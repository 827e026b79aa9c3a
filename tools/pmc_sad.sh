#!/usr/bin/env bash
# PMC passes over the TransE ranking kernels (count_sad_kernel, count_transe_big_kernel, rescore): separate passes,
# counters only (no tracing domains besides kernel-trace).  usage (GPU box): bash tools/pmc_sad.sh
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_sad; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  NQ=1024 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/bench_transe_eval.py > $OUT/p$i.log 2>&1 < /dev/null
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "count_sad" in name or "count_transe_big" in name or "rescore_pairs" in name:
            key = (name.split("(")[0].replace("void emg::", "").replace("emg::", "")[:44], r["Counter_Name"])
            agg[key] += float(r["Counter_Value"]); n[key] += 1
    for k, v in sorted(agg.items()):
        print("%-46s %-24s %.5g per launch (%d launches)" % (k[0], k[1], v / n[k], n[k]))
PY
done | tee $OUT/summary.txt
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
