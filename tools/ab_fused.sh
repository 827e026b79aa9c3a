for v in base occ4 keep occ4keep; do
  if [ $v = base ]; then unset EMGRAPH_HIP_LIB; else export EMGRAPH_HIP_LIB=$PWD/emgraph_amd/lib/variants/libemgraph_hip_$v.so; fi
  for i in 1 2; do python bench.py --no-eval --no-cpu --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items() if 'ms' in v})"; done
done
