set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "preprocess_and_binning_bit_exact or committed_rasterizer_fixture or config1_forward or ragged or culled_and_degenerate or non_finite or deterministic_backward_is_bit or skewed_scene or split_compositing or full_size_sampled or full_frame_parity or sharded_step_matches or row_order or owner_stage" 2>&1 | tail -25 > gpurun_out/r06/t1.log
cat gpurun_out/r06/t1.log
for i in 1 2; do
RDG_CULL=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_cull1_$i.json 2> gpurun_out/r06/bench_cull1_$i.err
RDG_CULL=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_cull0_$i.json 2> gpurun_out/r06/bench_cull0_$i.err
done
tail -c 600 gpurun_out/r06/bench_cull1_1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/bench_cull*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["value"], j.get("roofline",{}).get("achieved"), {k:v for k,v in j.get("config",{}).items() if k in ("D","instances","mean_instances")})
    except Exception as e: print(f, "ERR", e)
PY
