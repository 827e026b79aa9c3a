#!/usr/bin/env bash
# PMC passes over the exact-fast mode's prefilter kernels (bitmap form) at C4 size: separate passes, counters only (kernel-trace for names)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_prefilter; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/bench_prefilter.py "$@" > $OUT/p$i.log 2>&1
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "count_mfma_bf16" in r["Kernel_Name"] or "prefilter_compact" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"]); n[k] += 1
    for k, v in sorted(agg.items()):
        print("%-46s %-30s %.5g per launch (%d launches)" % (k[0], k[1], v / n[k], n[k]))
PY
done
