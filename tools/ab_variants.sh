#!/usr/bin/env bash
# A/B of library variants built by tools/build_variant.sh on one box: tools/ab_variants.sh "v1 v2 ..." [bench args]
vars="$1"; shift
for rep in 1 2; do
for v in $vars; do
  export EMGRAPH_HIP_LIB=$PWD/emgraph_amd/lib/variants/libemgraph_hip_$v.so
  python bench.py --quick --no-eval --no-cpu --steps 600 --warmup 30 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v})"
done
done
