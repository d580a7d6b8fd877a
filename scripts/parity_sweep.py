"""Randomised parity sweep on the GPU box: HIP path vs oracle on N random small scenes (random size, ragged image sizes,
SH degree, background, pose, scale modifier, gradient gates) with the per-column bar of the test-suite
(tests/test_gpu_parity.py::check_pair).  Prints one line per case and a summary; exit code 1 on any failure.
  usage: parity_sweep.py <cases> <seed0> [<first case index>]      RDG_SWEEP_PROFILE=aniso: pancakes and needles"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402
from oracle import rasterizer_oracle as O  # noqa: E402,F401
from sweep_cases import sweep_case, sweep_case_aniso  # noqa: E402

T.FLIP_ENTRIES = int(os.environ.get("RDG_SWEEP_FLIP_ENTRIES", "4"))     # see tests/test_gpu_parity.py::FLIP_ENTRIES


PROFILE = os.environ.get("RDG_SWEEP_PROFILE", "")        # "aniso": pancakes and needles (tests/sweep_cases.py::sweep_case_aniso)
import resolution  # noqa: E402  (tests/resolution.py: the arbiters a miss of the bar is taken to)


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0            # first case index (to re-run one case of a sweep)
bad = flips = 0
counts = {"f64": 0, "geom": 0, "f32": 0, "f32s": 0, "cond": 0}
for c in range(first, first + n_cases):
    sc, deg, bg, kw = (sweep_case_aniso if PROFILE == "aniso" else sweep_case)(seed0, c)
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
            # only a miss of the 1e-4 bar (an image, a gradient) can be a matter of float32 resolution; anything that must be exact
            # (radii, contributor counts beyond the allowance, the per-pixel state) is a failure whatever the float64 oracle says
            exact = res is None or not str(e).startswith(("d_", "color", "depth", "normal", "alpha", "final_T"))
            verdict, txt = ("fail", "") if exact else resolution.classify(sc, deg, bg, kw, res)
            what = {"f64": "outside the bar against the float32 oracle, inside it against the oracle run in float64:",
                    "geom": "inside the bar against the float64 oracle evaluated at the float32 geometry (the bit-exact per-Gaussian "
                            "forward both implementations share):",
                    "f32": "a column at float32 resolution: outside the bar against the float64 oracle, within "
                           f"{resolution.F32_FACTOR:g}x the float32 oracle's own distance from it:",
                    "f32s": "a scene no float32 evaluation resolves: outside the bar against the float64 oracle, within "
                            f"{resolution.F32_FACTOR:g}x the float32 oracle's LARGEST distance over the scene's gradient columns:",
                    "cond": "a column float32 INPUTS do not determine: the float64 oracle's own gradient moves by more than HIP's "
                            "distance when the inputs are perturbed by 2^-21 (four ulps):"}
            if verdict == "fail":
                bad += 1
                print("FAIL", tag, "\n     ", str(e)[:400], "\n     ", txt)
            else:
                counts[verdict] += 1
                print({"f64": "or64", "geom": "geom", "f32": "or32", "f32s": "o32s", "cond": "cond"}[verdict], tag, "\n     ", what[verdict],
                      str(e)[:300], "\n     ", txt)
inside = n_cases - bad - flips - sum(counts.values())
print(f"{inside} of {n_cases} cases within the per-column bar, {flips} more differ by a flipped pixel decision (contributor count or "
      f"final transmittance of a pixel differs), {counts['f64']} more are inside the bar against the oracle run in float64 where the "
      f"float32 oracle is not, {counts['geom']} more are inside it against the float64 oracle at the float32 geometry, {counts['f32']} "
      f"more have a column at float32 resolution (HIP outside the bar against float64, within {resolution.F32_FACTOR:g}x the float32 "
      f"oracle's own distance from it), {counts['f32s']} more are scenes no float32 evaluation resolves (within "
      f"{resolution.F32_FACTOR:g}x the float32 oracle's largest distance), {counts['cond']} more a column float32 inputs do not determine, {bad} fail")
sys.exit(1 if bad else 0)
