#!/bin/bash
# rocprofv3 kernel table of any python script on the GPU box:  scripts/prof_py.sh <tag> <rows> <script.py> [args ...]
# writes gpurun_out/<tag>_kernel_stats.csv and prints its top rows.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; ROWS=$2; shift 2
O=/tmp/prof_$TAG
rm -rf $O; mkdir -p $O gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 "$@" > $O/out.txt 2> $O/err.txt
cat $O/out.txt
F=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ]; then tail -5 $O/err.txt; exit 1; fi
cp $F gpurun_out/${TAG}_kernel_stats.csv
python3 - "$F" "$ROWS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "%8.2f ms total" % (float(r["TotalDurationNs"]) / 1e6))
PY
