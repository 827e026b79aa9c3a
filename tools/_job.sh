cd $GRAFT_REPO_ROOT
python -m pytest tests/test_graph_step.py tests/test_config_widths.py -q 2>&1 | tail -4
AB_WORKLOADS="C1 C3 C3p" bash tools/ab_step.sh "now:EMG_X=0" > gpurun_out/r4_r_ab.txt 2>&1
cat gpurun_out/r4_r_ab.txt
