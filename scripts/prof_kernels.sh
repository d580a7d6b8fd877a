#!/bin/bash
# rocprofv3 kernel table of one bench command on the GPU box:  scripts/prof_kernels.sh <tag> [rows] -- <bench args ...>
# writes gpurun_out/<tag>_kernel_stats.csv and prints its top rows (per launch and per step).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
ROWS=24
if [ "$1" != "--" ]; then ROWS=$1; shift; fi
shift
O=/tmp/prof_$TAG
rm -rf $O; mkdir -p $O gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu-baseline "$@" > $O/bench.json 2> $O/err.txt
F=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ]; then tail -5 $O/err.txt; exit 1; fi
cp $F gpurun_out/${TAG}_kernel_stats.csv
python3 - "$F" "$ROWS" $O/bench.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
try:
    b = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]); print("bench: %.4f ms/step" % b["ms_per_step"])
except Exception as e:
    print("no bench line", e)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.1f ms" % (tot / 1e6))
for r in rows[:int(sys.argv[2])]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "%8.2f ms total" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
PY
