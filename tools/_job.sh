cd $GRAFT_REPO_ROOT
python -m pytest tests/test_config_widths.py tests/test_graph_step.py -q -x 2>&1 | tail -6
python -m pytest tests/test_full_size.py -q -x -k "c3_exact" 2>&1 | tail -5
AB_WORKLOADS="C3a C3g C3mo" bash tools/ab_step.sh "new:EMG_X=0" "new2:EMG_X=0" > gpurun_out/r4_k_ab.txt 2>&1
cat gpurun_out/r4_k_ab.txt
