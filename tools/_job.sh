source tools/ab_env.sh
for w in C1 C2 C5; do
  run "$w" --workload $w
  run "$w" --workload $w
done
EMG_WIDE_GROUPS=0 run "C1 narrow" --workload C1
EMG_WIDE_GROUPS=0 run "C1 narrow" --workload C1
EMG_WIDE_GROUPS=0 run "C3p narrow" --workload C3p
run "C3p" --workload C3p
python -m pytest tests/test_graph_step.py tests/test_config_widths.py tests/test_api.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3
