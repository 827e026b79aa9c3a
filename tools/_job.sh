python -m pytest tests/test_library_switches.py -x -q -m gpu 2>&1 | tail -5
