"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
