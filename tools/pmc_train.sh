#!/usr/bin/env bash
# SQ / TCC counters of the training-step kernels (separate PMC passes, counters + kernel-trace only).
# usage (on the GPU box): bash tools/pmc_train.sh [kernel-name-substring ...]   (default: apply_rows train_backward)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_train; mkdir -p $OUT
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu --no-eval --no-pipeline ${EMG_PMC_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
pats = sys.argv[2:] or ["apply_rows", "train_backward", "group_", "prepare_"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(p in kn for p in pats):
            agg[kn[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[kn[:70]] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"))
for k in sorted(agg):
    print(k, "vgpr/agpr/sgpr/grid/wg/lds =", meta[k])
    for c, v in sorted(agg[k].items()):
        print("   %-24s avg %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
