"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean of every counter over the dispatches.
usage: python scripts/pmc_summary.py <dir with *_counter_collection.csv> [out.json [P W H [scene]]]
"_meta" stamps the file with bench.kernel_source_hash() and the workload (bench.py reads the VALU-issue figures of the
compositing kernels from it and drops them when the kernel sources have changed)."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"dispatches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
out["_meta"] = {"kernel_source_hash": kernel_source_hash(),
                "workload": {"points": int(sys.argv[3]), "width": int(sys.argv[4]), "height": int(sys.argv[5]),
                             "scene": sys.argv[6] if len(sys.argv) > 6 else "uniform"} if len(sys.argv) > 5 else None,
                "counters": "mean per dispatch, summed over the chip's SQs (rocprofv3 --pmc, own pass)"}
s = json.dumps(out, indent=1, sort_keys=True)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s)
print(s[:6000])
