mkdir -p gpurun_out/r06
T0=$(date +%s)
python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
T1=$(date +%s)
echo "bench default wall: $((T1-T0)) s"
tail -3 gpurun_out/r06/bench_default.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r06/bench_default.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["config"]["D"], j["config"]["D_composited"])
print(json.dumps(j["roofline"], indent=0)[:1800])
print(json.dumps(j.get("hbm_traffic_per_kernel_live"), indent=0))
print(j["sub_records"]["seconds_total"])
PY
