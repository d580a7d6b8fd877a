O=gpurun_out/prof_r06c
mkdir -p $O
T0=$(date +%s)
python bench.py > $O/r06_bench.json 2> $O/bench.err
T1=$(date +%s)
echo "bench default wall: $((T1-T0)) s"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/prof_r06c/r06_bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["traffic_over_algorithmic"], j["roofline"]["traffic_source"][:60])
print(json.dumps(j["sub_records"]["loop_100k"]))
print(j["sub_records"]["seconds_total"], j["parity_check"]["bench_frame"]["ok"], j["parity_check"]["c2"]["ok"])
PY
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -4
