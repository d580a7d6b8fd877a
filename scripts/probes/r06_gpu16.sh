python - <<'PY'
import ctypes, torch
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
a, b = ctypes.c_int(0), ctypes.c_int(0)
print("priority range rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)), "least", a.value, "greatest", b.value)
PY
for i in 1 2; do
RDG_EARLY_ROWS_ADAM=0 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early 0          ', b['ms_per_step'], b['value'])"
RDG_EARLY_ROWS_ADAM=1 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early 1 low prio ', b['ms_per_step'], b['value'])"
RDG_SIDE_STREAM_PRIORITY=0 RDG_EARLY_ROWS_ADAM=1 python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early 1 prio 0   ', b['ms_per_step'], b['value'])"
done
