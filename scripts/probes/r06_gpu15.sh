python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "mask_rank or birth_order or densif or graph" 2>&1 | tail -5
python -m pytest tests -x -q -m gpu -k "densif or deform or birth or rigid or split or prune" 2>&1 | tail -5
