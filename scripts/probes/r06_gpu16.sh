for i in 1 2; do
RDG_EAGER_POSE_FORK=0 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose fork 0', b['ms_per_step'], b['value'])"
RDG_EAGER_POSE_FORK=1 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose fork 1', b['ms_per_step'], b['value'])"
done
python -m pytest tests -x -q -m gpu -k "pose or rows_adam or train_step" 2>&1 | tail -4
