cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -q -k "one_call" 2>&1 | tail -3
EMG_INPLACE=1 EMG_ADAM_DEFERRED=1 EMG_FUZZ_SEEDS=400 python -m pytest tests/test_api.py -q -x -k "random_configurations" 2>&1 | tail -15
EMG_INPLACE=1 EMG_FUZZ_SEEDS=200 python -m pytest tests/test_api.py -q -x -k "random_configurations" 2>&1 | tail -8
