for w in "$@"; do python bench.py --quick --no-eval --no-cpu --steps 300 --warmup 20 --workload $w 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value']/1e6,1), d['ms_per_step'], d['host_issue_ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v}, d['stages'].get('_batch'))"; done
