for i in 1 2 3; do
RDG_EARLY_ROWS_ADAM_SPLIT=0 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split 0', b['ms_per_step'], b['value'])"
RDG_EARLY_ROWS_ADAM_SPLIT=1 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split 1', b['ms_per_step'], b['value'])"
done
python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "rows_adam or pose_chain" 2>&1 | tail -3
