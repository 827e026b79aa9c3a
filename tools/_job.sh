python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/profile_round4.sh r4_z
