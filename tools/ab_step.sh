#!/usr/bin/env bash
# A/B of the C3 / C3b / C2 step under environment switches: bash tools/ab_step.sh "LABEL:ENV=VAL ENV2=VAL2" ...
# (BENCH_ARGS=--no-pipeline among the switches adds bench.py arguments; the numbers come from bench_detail.json)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%:*}"; envs="${spec#*:}"
  for wl in ${AB_WORKLOADS:-C3 C3b C2}; do
    steps=300; [[ $wl == C3b ]] && steps=60
    env $envs bash -c 'python3 bench.py $BENCH_ARGS "$@"' _ --workload $wl --no-cpu --no-eval --no-others --sustained-seconds 0 --no-ceilings --steps $steps --warmup 20 >/dev/null 2>&1
    out=$(cat bench_detail.json)
    python3 - "$label" "$wl" "$out" <<'PY'
import json, sys
label, wl, out = sys.argv[1:4]
try:
    d = json.loads(out)
    st = {k: v["ms"] for k, v in d["stages"].items() if isinstance(v, dict) and "ms" in v}
    print("%-14s %-4s ms/step %.4f  host %.4f  stages %s" % (label, wl, d["ms_per_step"], d["host_issue_ms_per_step"], st))
except Exception as e:
    print(label, wl, "FAILED", e, out[:200])
PY
  done
done
