cd $GRAFT_REPO_ROOT
python -m pytest tests -q -x -m gpu 2>&1 | tail -15
AB_WORKLOADS="C3a C3g C1" bash tools/ab_step.sh "ieee:EMG_X=0" "fast:EMGRAPH_HIP_LIB=$GRAFT_REPO_ROOT/emgraph_amd/lib/variants/libemgraph_hip_fastrecip.so" "ieee2:EMG_X=0" > gpurun_out/r4_e_ab.txt 2>&1
cat gpurun_out/r4_e_ab.txt
