mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "fused_reference or graphed_iteration" 2>&1 | tail -30
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -x -q -k "fused_adam or reference_iteration" 2>&1 | tail -5
for mode in reference-v1 reference; do
  for pts in 1000000 200000; do
    python bench.py --iteration $mode --points $pts --steps 60 --settle 20 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', $pts, 'ms/sub-step', round(j['ms_per_sub_step'],4), 'it/s', round(j['value'],1))"
  done
done
for pts in 1000000 200000; do
python bench.py --iteration reference --graph --points $pts --steps 60 --settle 20 --warmup 3 2>gpurun_out/r06/ri_graph.err | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reference+graph', $pts, 'ms/sub-step', round(j['ms_per_sub_step'],4), 'it/s', round(j['value'],1))"
done
tail -5 gpurun_out/r06/ri_graph.err
