source tools/ab_env.sh
V=emgraph_amd/lib/variants
python -m pytest tests/test_graph_step.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
run "C3 main(prep4)" --workload C3
EMGRAPH_HIP_LIB=$V/libemgraph_hip_prep1.so run "C3 prep1" --workload C3
done
for w in C3a C1 C2 C5; do
run "$w main(prep4)" --workload $w
EMGRAPH_HIP_LIB=$V/libemgraph_hip_prep1.so run "$w prep1" --workload $w
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
