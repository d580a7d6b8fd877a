"""Print the kernel sequence of the LAST full train step from a rocprofv3 --kernel-trace CSV (kernel name, us)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with rdg_adam_multi_kernel followed by torch small-Adam kernels; take the span between the last two adam launches
idx = [i for i, r in enumerate(rows) if "rdg_adam_multi_kernel" in r["Kernel_Name"]]
# bench.py: 5 warm-up steps, 20 timed steps (only the dominant kernel bracketed by events), 5 steps with every stage
# bracketed (each bracket costs ~10 us of stream gap): show a step of the TIMED region unless told otherwise
which = int(sys.argv[2]) if len(sys.argv) > 2 else (19 if len(idx) >= 26 else len(idx) - 2)
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:110]}")
    prev_end = e
    tot += e - s
print("span us", (int(rows[b]["End_Timestamp"]) - t0) / 1e3, "busy us", tot / 1e3)
