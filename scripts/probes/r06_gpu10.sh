O=gpurun_out/prof_r06b
mkdir -p $O
python bench.py --loop 600 --points 100000 > $O/r06_loop_100k_bench_b.json 2>/dev/null
python bench.py --loop 600 --points 100000 --graph > $O/r06_loop_100k_graph_bench_b.json 2>/dev/null
python bench.py --loop 600 --points 100000 --fixed-capacity 1.2 > $O/r06_loop_100k_fixed_bench.json 2>/dev/null
python bench.py --loop 600 --points 100000 --fixed-capacity 1.2 --graph > $O/r06_loop_100k_fixed_graph_bench.json 2>/dev/null
python bench.py --loop 2500 --points 100000 --fixed-capacity 1.6 --graph > $O/r06_loop_100k_fixed_graph_long_bench.json 2>/dev/null
python bench.py --loop 2500 --points 100000 > $O/r06_loop_100k_long_bench.json 2>/dev/null
python bench.py --loop 600 --fixed-capacity 1.2 > $O/r06_loop_fixed_bench.json 2>/dev/null
python bench.py --loop 600 --fixed-capacity 1.2 --graph > $O/r06_loop_fixed_graph_bench.json 2>/dev/null
python bench.py --loop 600 > $O/r06_loop_bench_b.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/prof_r06b/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); L=j["loop"]
        print(f.split("/")[-1], "fps %.1f" % j["value"], "ratio %.4f" % L["sustained_over_steady"], "steady %.1f" % L["steady_state_fps_of_the_window"], "overflows", L["capacity_overflows"], "P", L["P_trajectory"][0], L["P_trajectory"][-1], "captures", j["config"].get("graph_captures"), "rows", j["config"].get("rows"), "dens ms %.2f" % L["densify_ms_mean"])
    except Exception as e: print(f, "ERR", e)
PY
