python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "rows_adam" 2>&1 | tail -8
