cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -q -k "block_tree or apply_rows" 2>&1 | tail -4
bash tools/profile_round4.sh r4_p
