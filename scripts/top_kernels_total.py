"""Top kernels of a rocprofv3 kernel_stats.csv by TOTAL time: name, calls, total ms, avg us."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%8.2f ms" % (float(r["TotalDurationNs"]) / 1e6), "%8.1f us" % (float(r["AverageNs"]) / 1e3))
print("total kernel ms", tot / 1e6)
