O=gpurun_out/prof_r06b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k "inplace or one_captured_graph" 2>&1 | tail -3
python bench.py --loop 2500 --points 100000 --fixed-capacity 1.3 --graph > $O/r06_loop_100k_fixed_graph_long_bench.json 2> $O/long.err
tail -3 $O/long.err
python bench.py --loop 2500 --fixed-capacity 1.3 --graph > $O/r06_loop_fixed_graph_long_bench.json 2> $O/long1m.err
tail -3 $O/long1m.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/prof_r06b/*long*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); L=j["loop"]
        print(f.split("/")[-1], "fps %.1f" % j["value"], "ratio %.4f" % L["sustained_over_steady"], "steady %.1f" % L["steady_state_fps_of_the_window"], "overflows", L["capacity_overflows"], "P", L["P_trajectory"][0], L["P_trajectory"][-1], "captures", j["config"].get("graph_captures"), "regrown", j["config"].get("capacity_regrown"), "rows", j["config"].get("rows"), "dens ms %.2f" % L["densify_ms_mean"], [round(d["ms"],1) for d in L["densifications"]])
    except Exception as e: print(f, "ERR", e)
PY
