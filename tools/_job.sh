cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -6
AB_WORKLOADS="C3a C3g C3" bash tools/ab_step.sh "new:EMG_X=0" "new2:EMG_X=0" > gpurun_out/r4_j_ab.txt 2>&1
cat gpurun_out/r4_j_ab.txt
