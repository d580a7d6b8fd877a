"""One line per bench JSON under a profiles directory: ms per step, value, and the fields the DESIGN §5 table quotes."""
import glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "profiles"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
for f in sorted(glob.glob(os.path.join(d, tag + "_*.json"))):
    try:
        txt = open(f).read().strip().splitlines()
        b = json.loads(txt[-1]) if txt else {}
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), "unreadable", e)
        continue
    if not isinstance(b, dict) or "ms_per_step" not in b:
        print(os.path.basename(f), {k: b[k] for k in list(b)[:6]} if isinstance(b, dict) else type(b))
        continue
    extra = {k: b[k] for k in ("sustained_over_steady", "steady_ms_per_step", "capacity_overflows", "graph_captures", "densifications",
                               "capacity_regrown", "ms_per_sub_step", "psnr_delta_db") if k in b}
    cfg = b.get("config", {})
    for k in ("P_final", "points_final", "D", "D_composited"):
        if k in cfg:
            extra[k] = cfg[k]
    print(f"{os.path.basename(f):52s} {b['ms_per_step']:.4f} ms  {b['value']:.1f} {b.get('unit','')}  {extra}")
