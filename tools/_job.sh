python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
source tools/ab_env.sh
for w in C3 C1 C2 C5; do run "$w" --workload $w; done
