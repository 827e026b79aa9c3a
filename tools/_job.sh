cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -q -x -k "transe_any_norm" 2>&1 | tail -5
python -m pytest tests/test_api.py -q -x -k "any_norm" 2>&1 | tail -5
python -m pytest tests/test_config_widths.py -q -x -k "falls_back" 2>&1 | tail -5
AB_WORKLOADS="C3 C3a C3g C1 C2 C5" bash tools/ab_step.sh "base:EMG_X=0" > gpurun_out/r4_a_ab.txt 2>&1
cat gpurun_out/r4_a_ab.txt
PMC_FILTER='apply_segments|deferred_catchup|train_fused' bash tools/pmc_kernels.sh r4_a_c3a --workload C3a > /dev/null 2>&1
bash tools/prof_quick.sh r4_a_c3a --workload C3a --steps 50 --warmup 10
