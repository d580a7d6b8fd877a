"""Prints the stage timers (and the step time) of bench lines given on the command line -- a compact A/B view."""
import json
import sys

KEYS = ("preprocess", "scan_dup", "sort", "render_fwd", "render_bwd", "preprocess_bwd", "deform_fwd", "deform_bwd", "adam",
        "loss_fwd", "loss_bwd", "mlp_fwd", "mlp_bwd")
for path in sys.argv[1:]:
    try:
        line = [ln for ln in open(path) if ln.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:                                   # noqa: BLE001
        print(f"{path}: unreadable ({e})")
        continue
    st = d.get("stage_ms", {})
    print(f"{path}: {d['ms_per_step']:.4f} ms/step  {d['value']:.1f} fps  " +
          " ".join(f"{k}={st.get(k, 0) * 1e3:.0f}" for k in KEYS))
