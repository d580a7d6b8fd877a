"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean of every counter over the dispatches.
usage: python scripts/pmc_summary.py <dir with *_counter_collection.csv> [out.json]"""
import csv, glob, json, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"dispatches": max(v[1] for v in cs.values())} for k, cs in acc.items()}
s = json.dumps(out, indent=1, sort_keys=True)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s)
print(s[:6000])
