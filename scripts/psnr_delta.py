"""PSNR-delta experiment (BASELINE.json metric, second half): optimise the SAME small scene with the SAME torch code
(activations, 0.8 L1 + 0.2 D-SSIM, torch.optim.Adam eps 1e-15, reference learning rates) once through the HIP
rasterizer (GPU) and once through the CPU oracle rasterizer, and compare the PSNR of the two results.

The reference CUDA rasterizer cannot run here; the oracle is the normative restatement of it, so this
measures "training through rodygs_amd lands where training through the specification lands".

    python scripts/psnr_delta.py --steps 60 --out profiles/r01_psnr_delta.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

from oracle import rasterizer_oracle as O                                  # noqa: E402  (checker side of the experiment)
from rodygs_amd.synthetic import synthetic_scene
from rodygs_amd.checkpoint import psnr                                     # noqa: E402
from rodygs_amd.losses import photometric_loss                             # noqa: E402  (torch restatement, both runs)


def make_params(sc, dev):
    op = sc["opacities"].clamp(1e-4, 1 - 1e-4)
    raw = {"xyz": sc["means3D"], "shs": sc["shs"], "scaling": torch.log(sc["scales"]), "rotation": sc["rotations"],
           "opacity": torch.log(op / (1 - op))}
    return {k: v.clone().to(dev).requires_grad_(True) for k, v in raw.items()}


def optimiser(p, spatial=5.0):
    return torch.optim.Adam([{"params": [p["xyz"]], "lr": 0.00016 * spatial}, {"params": [p["shs"]], "lr": 0.0025},
                             {"params": [p["opacity"]], "lr": 0.05}, {"params": [p["scaling"]], "lr": 0.005},
                             {"params": [p["rotation"]], "lr": 0.001}], eps=1e-15)


def activated(p):
    return dict(means3D=p["xyz"], shs=p["shs"], opacities=torch.sigmoid(p["opacity"]), scales=torch.exp(p["scaling"]),
                rotations=torch.nn.functional.normalize(p["rotation"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--points", type=int, default=1000)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    W = H = a.size
    sc = synthetic_scene(a.points, W, H, 3, seed=3)
    tgt = synthetic_scene(a.points, W, H, 3, seed=4)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 3)
    with torch.no_grad():
        gt = O.rasterize(tgt["means3D"], torch.zeros(a.points, 3), tgt["opacities"], tgt["viewmatrix"], st, shs=tgt["shs"],
                         scales=tgt["scales"], rotations=tgt["rotations"])[0].clamp(0, 1)

    def render_oracle(p):
        act = activated(p)
        return O.rasterize(act["means3D"], torch.zeros(a.points, 3), act["opacities"], sc["viewmatrix"], st, shs=act["shs"],
                           scales=act["scales"], rotations=act["rotations"])[0]

    from rodygs_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda")
    rs = GaussianRasterizationSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3, device=dev), 1.0,
                                       sc["projmatrix"].to(dev), 3, False, False, True, True)
    vm = sc["viewmatrix"].to(dev)

    def render_hip(p):
        act = activated(p)
        m2 = torch.zeros(a.points, 3, device=dev, requires_grad=True)
        return GaussianRasterizer(rs)(means3D=act["means3D"], means2D=m2, shs=act["shs"], opacities=act["opacities"],
                                      scales=act["scales"], rotations=act["rotations"], viewmatrix=vm)[0]

    res = {}
    for name, render, d in (("hip", render_hip, dev), ("oracle", render_oracle, torch.device("cpu"))):
        p = make_params(sc, d)
        opt = optimiser(p)
        g = gt.to(d)
        t0 = time.perf_counter()
        with torch.no_grad():
            first = float(psnr(g, render(p)))
        for _ in range(a.steps):
            opt.zero_grad(set_to_none=True)
            photometric_loss(render(p), g, 0.2).backward()
            opt.step()
        with torch.no_grad():
            last = float(psnr(g, render(p)))
        res[name] = {"psnr_start_db": first, "psnr_end_db": last, "seconds": time.perf_counter() - t0}
        print(name, res[name], flush=True)
    res["delta_db"] = res["hip"]["psnr_end_db"] - res["oracle"]["psnr_end_db"]
    res["config"] = {"points": a.points, "size": a.size, "steps": a.steps, "sh_degree": 3,
                     "loss": "0.8 L1 + 0.2 D-SSIM", "optimizer": "torch Adam eps 1e-15, reference learning rates"}
    print(json.dumps(res))
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
