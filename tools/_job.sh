cd $GRAFT_REPO_ROOT
python -m pytest tests/test_config_widths.py -q -x -k "replayed or inplace_choice or deferred" 2>&1 | tail -6
python -m pytest tests/test_full_size.py -q -x -k "c3_exact" 2>&1 | tail -5
AB_WORKLOADS="C3a" bash tools/ab_step.sh "w3:EMG_X=0" "w2:EMGRAPH_HIP_LIB=$GRAFT_REPO_ROOT/emgraph_amd/lib/variants/libemgraph_hip_ip6w2.so" "w3b:EMG_X=0" "w2b:EMGRAPH_HIP_LIB=$GRAFT_REPO_ROOT/emgraph_amd/lib/variants/libemgraph_hip_ip6w2.so" > gpurun_out/r4_i_ab.txt 2>&1
AB_WORKLOADS="C1 C2 C5" bash tools/ab_step.sh "dflt:EMG_X=0" "inpl:EMG_INPLACE=1" "dflt2:EMG_X=0" "inpl2:EMG_INPLACE=1" >> gpurun_out/r4_i_ab.txt 2>&1
cat gpurun_out/r4_i_ab.txt
