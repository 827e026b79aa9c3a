python tools/sweep_small.py C1 2>&1 | grep -v amdgpu.ids | head -8
echo "--- deep"
EMG_DEEP_B=100000 python tools/sweep_small.py C1 2>&1 | grep -v amdgpu.ids | head -8
