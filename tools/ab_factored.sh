#!/usr/bin/env bash
# A/B on one box: factored entity contributions on / off (EMG_FACTORED), C3 and the secondary workloads
for f in 1 0 1 0; do
  echo "== C3 EMG_FACTORED=$f"
  EMG_FACTORED=$f python bench.py --quick --no-eval --no-cpu --steps 1000 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); st=d.get('stages') or {}
print(d['ms_per_step'], d['value'], {k:(v.get('ms') if isinstance(v,dict) else v) for k,v in st.items() if k!='_batch'})"
done
for f in 1 0; do
  for w in C2 C5 C3z C3b; do
    echo "== $w EMG_FACTORED=$f"
    EMG_FACTORED=$f python bench.py --workload $w --quick --no-eval --no-cpu --steps 500 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); st=d.get('stages') or {}
print(d['ms_per_step'], d['value'], {k:(v.get('ms') if isinstance(v,dict) else v) for k,v in st.items() if k!='_batch'})"
  done
done
