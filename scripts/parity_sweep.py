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

import resolution  # noqa: E402  (tests/resolution.py: the arbiters a miss of the bar is taken to)
import sweep_run  # noqa: E402  (tests/sweep_run.py: one case, shared with the frozen-rules -m gpu test)

sweep_run.SWEEP_FLIP_ENTRIES = int(os.environ.get("RDG_SWEEP_FLIP_ENTRIES", str(sweep_run.SWEEP_FLIP_ENTRIES)))
PROFILE = os.environ.get("RDG_SWEEP_PROFILE", "")        # "aniso": pancakes and needles (tests/sweep_cases.py::sweep_case_aniso)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0            # first case index (to re-run one case of a sweep)
bad = flips = 0
counts = {"f64": 0, "geom": 0, "f32": 0, "f32s": 0, "cond": 0}
print(f"rules {sweep_run.rules_hash()}  profile {PROFILE or 'regular'}  seed0 {seed0}")
for c in range(first, first + n_cases):
    verdict, tag, txt = sweep_run.run_case(PROFILE, seed0, c)
    if verdict == "ok":
        print("ok  ", tag)
    elif verdict == "flip":
        flips += 1
        print("flip", tag, "\n     ", txt)
    elif verdict == "fail":
        bad += 1
        print("FAIL", tag, "\n     ", txt)
    else:
        counts[verdict] += 1
        print({"f64": "or64", "geom": "geom", "f32": "or32", "f32s": "o32s", "cond": "cond"}[verdict], tag, "\n     ", txt)
inside = n_cases - bad - flips - sum(counts.values())
print(f"{inside} of {n_cases} cases within the per-column bar, {flips} more differ by a flipped pixel decision (contributor count or "
      f"final transmittance of a pixel differs), {counts['f64']} more are inside the bar against the oracle run in float64 where the "
      f"float32 oracle is not, {counts['geom']} more are inside it against the float64 oracle at the float32 geometry, {counts['f32']} "
      f"more have a column at float32 resolution (HIP outside the bar against float64, within {resolution.F32_FACTOR:g}x the float32 "
      f"oracle's own distance from it), {counts['f32s']} more are scenes no float32 evaluation resolves (within "
      f"{resolution.F32_FACTOR:g}x the float32 oracle's largest distance), {counts['cond']} more a column float32 inputs do not determine, {bad} fail")
sys.exit(1 if bad else 0)
