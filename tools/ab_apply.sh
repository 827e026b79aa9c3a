for w in 0 4 8 16 32 64; do
  if [ $w = 0 ]; then unset EMG_APPLY_WIN; else export EMG_APPLY_WIN=$w; fi
  python bench.py --no-eval --no-cpu --steps 100 --warmup 10 --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('win=$w nopipe', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v})"
done
unset EMG_APPLY_WIN
python bench.py --no-eval --no-cpu --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v})"
