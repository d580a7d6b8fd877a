#!/bin/bash
# Where do the binning count pass's returning integer atomics (and the compositing backward's float atomics) execute?
# Lists the TCC atomic counters this rocprofv3 offers, collects up to four of them (own --pmc pass, no trace domain) over
# a short bench run and prints their per-dispatch means for the kernels concerned.   usage: scripts/atomic_probe.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/atomic_probe
rm -rf $O /tmp/atomic_pmc; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -o -E "\bTCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*\b" $O/avail.txt | sort -u > $O/atomic_counters.txt
cat $O/atomic_counters.txt
C=$(grep -E "_sum$" $O/atomic_counters.txt | grep -v -E "\[|LEVEL" | head -4 | tr '\n' ' ')
[ -z "$C" ] && C=$(head -2 $O/atomic_counters.txt | tr '\n' ' ')
echo "collecting: $C TCC_REQ_sum"
rocprofv3 --pmc $C TCC_REQ_sum --output-format csv -d /tmp/atomic_pmc -- python3 bench.py --steps 4 --warmup 2 --settle 2 --no-cpu-baseline > /dev/null 2> $O/pmc.err
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/atomic_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    if any(s in k for s in ("bucket", "render_bwd", "tile_scan", "adam_multi", "preprocess_bwd")):
        print(k, {c: round(v[0] / v[1]) for c, v in cs.items()}, "dispatches", max(v[1] for v in cs.values()))
PY
tail -3 $O/pmc.err
