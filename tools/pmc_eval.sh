#!/usr/bin/env bash
# PMC passes over the evaluation section of bench.py (exact f32 + bf16 count kernels)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_eval; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/bench_bf16_count.py --f32 --reps 1 > $OUT/p$i.log 2>&1
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "count_" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])
            agg[k] = max(agg[k], float(r["Counter_Value"]))   # the largest launch (the timed 8192-row pass)
    for k, v in sorted(agg.items()):
        print("%-46s %-28s %.4g" % (k[0], k[1], v))
PY
done
