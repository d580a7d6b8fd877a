"""Randomised parity sweep on the GPU box: HIP path vs oracle on N random small scenes (random size, ragged image sizes,
SH degree, background, pose, scale modifier, gradient gates) with the per-column bar of the test-suite
(tests/test_gpu_parity.py::check_pair).  Prints one line per case and a summary; exit code 1 on any failure."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402
from oracle import rasterizer_oracle as O  # noqa: E402,F401
from sweep_cases import sweep_case  # noqa: E402

T.FLIP_ENTRIES = int(os.environ.get("RDG_SWEEP_FLIP_ENTRIES", "4"))     # see tests/test_gpu_parity.py::FLIP_ENTRIES
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = flips = 0
for c in range(n_cases):
    sc, deg, bg, kw = sweep_case(seed0, c)
    P, W, H, deg_max = sc["means3D"].shape[0], sc["W"], sc["H"], int(round(sc["shs"].shape[1] ** 0.5)) - 1
    tag = f"case {c:3d}: P={P:5d} {W}x{H} deg {deg}/{deg_max} {kw}"
    res = None
    try:
        res = T.run_pair(sc, deg, bg, **kw)
        T.check_pair(res, T.NAMES)
        print("ok  ", tag)
    except Exception as e:                                            # noqa: BLE001
        # a pixel whose blend / stop decision differs between the two implementations (alpha on the 1/255 boundary,
        # T on the 1e-4 one) shows in the per-pixel contributor count: such a case is the discontinuity, not an error
        n_flip = -1
        if res is not None:
            fT_h, nc_h = res[2][6]
            fT_o, nc_o = res[5][5]["final_T"], res[5][5]["n_contrib"]
            # a decision flipped on the LAST splat of a pixel changes its contributor count, one in the middle of the list
            # changes the pixel's final transmittance by that splat's alpha (>= 1/255) and nothing else
            n_flip = int(((nc_h.cpu() != nc_o) | ((fT_h.cpu().double() - fT_o.double()).abs() > 1e-3 * fT_o.double().abs())).sum())
        if n_flip > 0:
            flips += 1
            print("flip", tag, f"\n      {n_flip} pixel(s) with another contributor count / transmittance;", str(e)[:300])
        else:
            bad += 1
            print("FAIL", tag, "\n     ", str(e)[:400])
print(f"{n_cases - bad - flips} of {n_cases} cases within the per-column bar, {flips} more differ by a flipped pixel decision "
      f"(contributor count or final transmittance of a pixel differs), {bad} fail")
sys.exit(1 if bad else 0)
