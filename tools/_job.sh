# scratch job script for `gpurun -- 'bash tools/_job.sh'` (the round's last content: the full GPU suite + the evidence run)
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/profile_round4.sh r4_z
