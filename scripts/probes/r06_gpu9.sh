mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k "inplace or one_captured_graph" 2>&1 | tail -5
for rep in 1 2; do
for pts in 100000; do
  python bench.py --loop 600 --points $pts 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=j['loop']; print('eager          ', $pts, 'fps %.1f' % j['value'], 'ratio %.4f' % L['sustained_over_steady'], 'steady %.1f' % L['steady_state_fps_of_the_window'], 'densify ms %.2f' % L['densify_ms_mean'])"
  python bench.py --loop 600 --points $pts --fixed-capacity 1.2 --graph 2>gpurun_out/r06/fixed_graph.err | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=j['loop']; print('fixed graph    ', $pts, 'fps %.1f' % j['value'], 'ratio %.4f' % L['sustained_over_steady'], 'steady %.1f' % L['steady_state_fps_of_the_window'], 'captures', j['config']['graph_captures'], 'densify ms %.2f' % L['densify_ms_mean'], [round(s['graph_capture_ms'],1) for s in L['segments']])"
done
done
python bench.py --loop 600 --points 100000 --fixed-capacity 1.2 --graph --loop-profile 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=j['loop']; print(json.dumps(L['densifications'][1], indent=0)[:800]); print(L['segments'][0])"
