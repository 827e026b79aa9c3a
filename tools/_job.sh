cd $GRAFT_REPO_ROOT
python -m pytest tests/test_config_widths.py tests/test_graph_step.py -q -x 2>&1 | tail -8
python -m pytest tests/test_api.py tests/test_hip_kernels.py -q -x 2>&1 | tail -5
AB_WORKLOADS="C3p C1 C3" bash tools/ab_step.sh "packed:EMG_X=0" "full:EMG_FACTORED=0" "packed2:EMG_X=0" "full2:EMG_FACTORED=0" > gpurun_out/r4_q_ab.txt 2>&1
cat gpurun_out/r4_q_ab.txt
