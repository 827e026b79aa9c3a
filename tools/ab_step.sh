#!/usr/bin/env bash
# A/B of the C3 / C3b / C2 step under environment switches: bash tools/ab_step.sh "LABEL:ENV=VAL ENV2=VAL2" ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%:*}"; envs="${spec#*:}"
  for wl in ${AB_WORKLOADS:-C3 C3b C2}; do
    steps=300; [[ $wl == C3b ]] && steps=60
    out=$(env $envs python3 bench.py --workload $wl --no-cpu --no-eval --no-others --sustained-seconds 0 --no-ceilings --steps $steps --warmup 20 2>/dev/null | tail -1)
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
