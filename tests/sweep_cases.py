"""The random small scenes of scripts/parity_sweep.py as a function of (seed0, case index), so that a case the sweep
flagged can be named in a test and taken apart by the debug scripts."""
import random

from oracle import rasterizer_oracle as O


def orbit_view(deg_y, deg_x, t):
    import test_gpu_parity as T
    return T.orbit_view(deg_y, deg_x, t)


def sweep_case(seed0: int, c: int):
    """(scene dict, active SH degree, background, run_pair keyword arguments) of case ``c`` of the sweep seeded ``seed0``."""
    rng = random.Random(seed0 + c)
    P = rng.choice([1, 7, 63, 64, 65, 200, 777, 1500, 3000, 5000])
    W = rng.choice([16, 33, 100, 128, 250, 320, 401])
    H = rng.choice([16, 17, 96, 128, 200, 240, 333])
    deg_max = rng.choice([0, 1, 2, 3])
    deg = rng.randint(0, deg_max)
    bg = tuple(rng.random() for _ in range(3))
    sc = O.synthetic_scene(P, W, H, deg_max, seed=seed0 + c)
    sc["viewmatrix"] = orbit_view(rng.uniform(-20, 20), rng.uniform(-15, 15),
                                  (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 1.5)))
    if rng.random() < 0.3:
        sc["scales"] = sc["scales"] * rng.uniform(1.5, 5.0)          # long lists, overflow retries
    kw = dict(cov_grad=rng.random() < 0.8, sh_grad=rng.random() < 0.8, scale_modifier=rng.choice([1.0, 1.0, 0.7, 1.3]),
              seed=seed0 + c, normal_loss=rng.choice([0.0, 0.0, 0.5]), depth_loss=rng.choice([0.1, 0.1, 0.0]))
    return sc, deg, bg, kw
