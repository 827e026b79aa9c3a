python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "any_norm" 2>&1 | tail -30
