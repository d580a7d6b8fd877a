"""Build profiles/*_pmc_hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs as
MI355X_MICROARCH.md prescribes).  usage: python scripts/pmc_hbm_traffic.py <fetch_dir> <write_dir> <out.json> P W H [scene]
The file is stamped with bench.kernel_source_hash(): bench.py reports the traffic figure only while the dominant
kernel's sources are the ones the counters were taken with."""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return re.sub(r"<.*>", "", n).strip()


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (two separate passes) --output-format csv -- python "
                  "bench.py --steps 4 --warmup 2 --no-cpu-baseline",
       "workload": {"points": int(sys.argv[4]), "width": int(sys.argv[5]), "height": int(sys.argv[6]),
                    "scene": sys.argv[7] if len(sys.argv) > 7 else "uniform"},
       "kernel_source_hash": kernel_source_hash(),
       "units": "bytes per launch; FETCH_SIZE/WRITE_SIZE are KB counters; hbm_bytes_corrected = 2*FETCH_SIZE*1024 + "
                "WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of wide reads, MI355X_MICROARCH.md HBM section; float "
                "atomics are counted in WRITE_SIZE)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("rdg_"):
        continue
    f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
    out["kernels"][k] = {"launches": max(nf, nw), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                         "hbm_bytes_corrected": 2 * f * 1024 + w * 1024, "hbm_bytes_uncorrected": (f + w) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"])[:12]:
    print(f"{k:40s} {v['hbm_bytes_corrected'] / 1e6:9.1f} MB  (fetch {v['FETCH_SIZE_KB'] / 1e3:8.1f} MB x2, write {v['WRITE_SIZE_KB'] / 1e3:8.1f} MB)")
