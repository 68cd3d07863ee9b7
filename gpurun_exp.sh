python -m pytest tests/test_guard.py -m gpu -q -x 2>&1 | tail -3
