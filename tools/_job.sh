source tools/ab_env.sh
V=emgraph_amd/lib/variants
for i in 1 2; do
for w in C1 C3p; do
run "$w main" --workload $w
EMGRAPH_HIP_LIB=$V/libemgraph_hip_sgnold.so run "$w sgnold" --workload $w
done
done
python -m pytest tests/test_graph_step.py tests/test_config_widths.py tests/test_api.py tests/test_hip_kernels.py tests/test_full_size.py -x -q -m gpu 2>&1 | tail -3
