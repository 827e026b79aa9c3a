#!/usr/bin/env bash
# A/B of the evaluation kernels (C4: 1M entities, ComplEx k = 200) under environment switches: bash tools/ab_eval.sh "LABEL:ENV=VAL ..." ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%:*}"; envs="${spec#*:}"
  env $envs python3 bench.py --quick --no-cpu --no-ceilings --steps 20 --warmup 5 >/dev/null 2>gpurun_out/ab_eval.err
  python3 - "$label" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
e = d.get("eval", {})
def g(o, *ks):
    for k in ks:
        o = o.get(k, {}) if isinstance(o, dict) else {}
    return o
print(sys.argv[1], json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if isinstance(vv, (int, float, str, bool))}) for k, v in e.items()})[:1500])
PY
done
