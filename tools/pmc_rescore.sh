#!/usr/bin/env bash
# PMC passes over the exact-fast evaluation (f16 prefilter + rescore_pairs_kernel): where do the re-scored rows come from?
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_rescore; mkdir -p $OUT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/bench_exact_fast.py 4096 > $OUT/p$i.log 2>&1
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rescore" in r["Kernel_Name"] or "v3_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-46s %-24s %s" % (k[0], k[1], " ".join("%.4g" % x for x in v[-4:])))
PY
done
