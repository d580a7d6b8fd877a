python - <<'PY'
import torch
from rodygs_amd import trainstep as TS
st = TS._low_priority_stream(torch.device("cuda:0"))
print(type(st).__name__, "priority", st.priority)
PY
python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "rows_adam or pose_chain" 2>&1 | tail -1
python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['value'])"
