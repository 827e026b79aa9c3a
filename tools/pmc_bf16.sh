#!/usr/bin/env bash
# PMC passes over the bf16 count kernels (separate passes; counters only, no tracing domains besides kernel-trace)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bf16; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/bench_bf16_count.py --reps 1 "$@" > $OUT/p$i.log 2>&1
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "count_mfma_bf16" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        print("%-42s %-28s %.4g" % (k[0], k[1], v))
PY
done
