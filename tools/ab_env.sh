#!/usr/bin/env bash
# helper for A/B runs on one box:  source tools/ab_env.sh; run LABEL [bench args];  ENV=... run LABEL2 ...
run() { python bench.py --quick --no-eval --no-cpu --steps 600 --warmup 30 "${@:2}" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v})"; }
