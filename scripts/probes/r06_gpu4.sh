mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma_mlp or deformation_field_matches or mlp_and_pose_grad_sinks or graph_replay_of_the_train_step or train_step_runs or deterministic_train_step or nan_filled or sharded_step_matches" 2>&1 | tail -15
for i in 1 2; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_mlpf_$i.json 2>/dev/null
RDG_MLP_UNFUSED=1 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench_mlpu_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/bench_mlp*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["ms_per_step"],4), round(j["value"],1), {k:round(v,4) for k,v in j["stage_ms"].items() if k.startswith("mlp")})
PY
