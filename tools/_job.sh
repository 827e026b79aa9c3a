source tools/ab_env.sh
for w in C1 C2 C5 C3a C3g; do
run "$w fix" --workload $w
EMG_APPLY_FIX=0 run "$w nofix" --workload $w
run "$w fix" --workload $w
EMG_APPLY_FIX=0 run "$w nofix" --workload $w
done
python -m pytest tests/test_graph_step.py tests/test_config_widths.py tests/test_api.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3
