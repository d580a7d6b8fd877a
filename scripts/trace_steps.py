"""Step periods of a rocprofv3 --kernel-trace CSV of bench.py: for every train step (delimited by rdg_adam_multi_kernel)
the period, the busy time and the largest idle gaps with the kernels around them -- where the GPU waits for the host."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rdg_adam_multi_kernel" in r["Kernel_Name"]]
gaps_by_pair = {}
for n in range(len(idx) - 1):
    a, b = idx[n], idx[n + 1]
    t0 = int(rows[a]["End_Timestamp"])
    prev_end, busy, big = t0, 0, []
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = s - prev_end
        if g > 3000:
            big.append((g / 1e3, r["Kernel_Name"][:40]))
            key = r["Kernel_Name"][:40]
            gaps_by_pair.setdefault(key, []).append(g / 1e3)
        busy += e - s
        prev_end = max(prev_end, e)
    period = (int(rows[b]["End_Timestamp"]) - t0) / 1e3
    print(f"step {n:3d}: period {period:8.1f} us  busy {busy / 1e3:8.1f}  idle {period - busy / 1e3:7.1f}  "
          + "  ".join(f"{g:.0f}us before {k}" for g, k in sorted(big, reverse=True)[:4]))
print("gaps > 3 us by the kernel that follows them (count, mean us):")
for k, v in sorted(gaps_by_pair.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:42s} {len(v):4d}  {sum(v) / len(v):8.1f}")
