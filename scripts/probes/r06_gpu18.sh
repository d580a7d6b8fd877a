for i in 1 2 3; do python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "pose_chain or rows_adam" 2>&1 | tail -3; done
