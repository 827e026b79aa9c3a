run() { python bench.py --no-eval --no-cpu --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v})"; }
run base; run base
EMG_NO_LONG=1 run nolong; EMG_NO_LONG=1 run nolong
EMG_APPLY_DEEP=1 run deep; EMG_APPLY_DEEP=0 run lean
EMG_NO_LONG=1 EMG_APPLY_DEEP=1 run nolong_deep
EMG_NO_LONG=1 EMG_APPLY_DEEP=0 run nolong_lean
