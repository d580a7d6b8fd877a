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


def sweep_case_aniso(seed0: int, c: int):
    """The same case with the shapes a TRAINED scene is made of: surface-aligned pancakes (one axis 10-300x thinner than the
    other two) and needles (two thin axes) instead of the generator's mildly anisotropic blobs (its axes differ by exp(0.5 N)).
    Derivatives with respect to a thin axis are small differences of large terms in any formulation: this profile is where a
    hand-derived backward and autograd part ways first (finding 3).  Own random stream: the draws of
    sweep_case are untouched."""
    import torch
    sc, deg, bg, kw = sweep_case(seed0, c)
    rng = random.Random(7919 * (seed0 + c) + 13)
    g = torch.Generator().manual_seed(seed0 + c + 555)
    P = sc["scales"].shape[0]
    kind = torch.rand(P, generator=g)
    thin = torch.pow(10.0, -(1.0 + 1.5 * torch.rand(P, generator=g)))            # 0.1 ... 0.003
    axis = torch.randint(0, 3, (P,), generator=g)
    f = torch.ones(P, 3)
    pancake = kind < 0.6                                                           # 60 % pancakes, 25 % needles, 15 % blobs
    needle = (kind >= 0.6) & (kind < 0.85)
    f[pancake, axis[pancake]] = thin[pancake]
    f[needle] = thin[needle, None].expand(-1, 3).clone()
    f[needle, axis[needle]] = 1.0
    sc["scales"] = sc["scales"] * f * rng.choice([1.0, 1.0, 2.0])                 # (some a little larger: the thin ones stay visible)
    return sc, deg, bg, kw
