source tools/ab_env.sh
V=emgraph_amd/lib/variants
for i in 1 2 3; do
run "C3a main" --workload C3a
EMGRAPH_HIP_LIB=$V/libemgraph_hip_fastloss.so run "C3a fastloss" --workload C3a
done
