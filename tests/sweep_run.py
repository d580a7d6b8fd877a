"""One case of the randomised parity sweep (scripts/parity_sweep.py, tests/test_gpu_round6.py): a random small scene through the
HIP path and the oracle, held to check_pair's per-column bar; a miss is taken to the arbiters of tests/resolution.py.
verdict: "ok" | "flip" (a witnessed pixel-decision flip explains the miss) | "f64" | "geom" | "f32" | "f32s" | "cond" | "fail"."""
import hashlib
import inspect
import os

import resolution
import test_gpu_parity as T
from sweep_cases import sweep_case, sweep_case_aniso

HERE = os.path.dirname(os.path.abspath(__file__))
SWEEP_FLIP_ENTRIES = 4          # entries of a column one flipped pixel may move in a sweep scene (see T.FLIP_ENTRIES)


def rules_hash() -> str:
    """sha256 over everything that decides a sweep verdict: the arbiters (resolution.py), the comparison rules
    (oracle/parity.py, rel_ok / check_pair / run_pair of the parity tests), the case generators and this file.  The frozen-rules
    test holds a copy: a later edit of any rule shows up as a change of that constant in the same commit."""
    h = hashlib.sha256()
    for path in (os.path.join(HERE, "resolution.py"), os.path.join(HERE, "sweep_cases.py"), os.path.join(HERE, "sweep_run.py"),
                 os.path.join(os.path.dirname(HERE), "oracle", "parity.py")):
        with open(path, "rb") as f:
            h.update(f.read())
    for fn in (T.rel_ok, T.check_pair, T.run_pair):
        h.update(inspect.getsource(fn).encode())
    h.update(repr((T.TOL, T.OUTLIER_FRAC, T.OUTLIER_CAP, SWEEP_FLIP_ENTRIES, resolution.TOL, resolution.F32_FACTOR,
                   resolution.COND_REL)).encode())
    return h.hexdigest()[:16]


def run_case(profile: str, seed0: int, c: int):
    """-> (verdict, tag, text)"""
    sc, deg, bg, kw = (sweep_case_aniso if profile == "aniso" else sweep_case)(seed0, c)
    P, W, H, deg_max = sc["means3D"].shape[0], sc["W"], sc["H"], int(round(sc["shs"].shape[1] ** 0.5)) - 1
    tag = f"case {c:3d}: P={P:5d} {W}x{H} deg {deg}/{deg_max} {kw}"
    keep = T.FLIP_ENTRIES
    T.FLIP_ENTRIES = SWEEP_FLIP_ENTRIES
    res = None
    try:
        res = T.run_pair(sc, deg, bg, **kw)
        T.check_pair(res, T.NAMES)
        return "ok", tag, ""
    except Exception as e:                                            # noqa: BLE001
        # a pixel whose blend / stop decision differs between the two implementations (alpha on the 1/255 boundary, T on the
        # 1e-4 one) shows in the per-pixel contributor count / final transmittance: such a case is the discontinuity, not an error
        n_flip = -1
        if res is not None:
            fT_h, nc_h = res[2][6]
            fT_o, nc_o = res[5][5]["final_T"], res[5][5]["n_contrib"]
            n_flip = int(T.flipped_pixels(fT_h, nc_h, fT_o, nc_o))
        if n_flip > 0:
            return "flip", tag, f"{n_flip} pixel(s) with another contributor count / transmittance; {str(e)[:300]}"
        # only a miss of the 1e-4 bar (an image, a gradient) can be a matter of float32 resolution; anything that must be exact
        # (radii, contributor counts beyond the allowance, the per-pixel state) is a failure whatever the float64 oracle says
        exact = res is None or not str(e).startswith(("d_", "color", "depth", "normal", "alpha", "final_T"))
        if exact:
            return "fail", tag, str(e)[:400]
        verdict, txt = resolution.classify(sc, deg, bg, kw, res)
        return verdict, tag, str(e)[:300] + " || " + txt
    finally:
        T.FLIP_ENTRIES = keep
