for i in 1 2; do
for B in 16384 256 512 1024 2048; do
RDG_ADAM_BLOCKS=$B python bench.py --no-sub-records --no-live-pmc --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adam blocks $B', round(b['ms_per_step'],4), round(b['value'],1), 'adam', b['stage_ms'].get('adam'), 'mlp_bwd', b['stage_ms'].get('mlp_bwd'))"
done
done
