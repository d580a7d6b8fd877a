"""Generates the rasterizer fixture G4 from the ORACLE: tests/golden/rasterizer_golden_<scene>.npz.

    python tests/golden/make_rasterizer_golden.py

The reference holds no source, test or golden vector for its rasterizer (un-vendored submodule,
/root/reference/.gitmodules:1-4; call site src/trainer/renderer.py:87-101), so nothing reference-derived can pin
that stage.  What CAN be pinned is the spec itself: this file freezes what oracle/rasterizer_oracle.py produced, so
that a later joint drift of the oracle and the kernels (both edited the same way, live comparison still green)
shows up against committed numbers.  Two scenes:
  c1      BASELINE.json configs[0]: 1 k Gaussians, 256x256, SH degree 0 (stored [P,16,3] as the reference does),
          identity pose, black background;
  skewed  rodygs_amd.synthetic.skewed_scene: one tile with > 8192 instances, two with 1.5-3 k, depth ties, orbit
          pose, coloured background, SH degree 1 -- the long-list paths of binning and compositing.
Stored per scene: every input, and radii, tiles_touched, D, unsorted / sorted keys + values, tile ranges,
n_contrib, final_T, colour / depth / normal / alpha, and the gradient of `fixture_loss` w.r.t. every input
including viewmatrix and means2D -- all under the REFERENCE's tile-rectangle rule (OracleSettings.cull = False) --, plus
(`cull_*`, round 6) the integers that change under the tight rectangles of RdgRasterSettings.cull = 1: tiles_touched, D,
sorted keys / values, ranges and n_contrib (a list position).  Arrays only (no source text)."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

INPUTS = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
CULL_KEYS = ("tiles_touched", "num_rendered", "keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted", "ranges",
             "n_contrib")


def fixture_weights(H, W):
    """Loss weights as a closed form of the pixel coordinates (no RNG to drift): colour, depth, alpha."""
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    wc = torch.stack([0.6 + 0.4 * torch.sin(0.37 * x + 0.11 * y + c) for c in (0.0, 1.0, 2.0)])
    wd = (0.5 + 0.5 * torch.cos(0.05 * x - 0.23 * y)).unsqueeze(0)
    wa = (0.5 + 0.5 * torch.sin(0.19 * x + 0.07 * y + 0.5)).unsqueeze(0)
    return wc, wd, wa


def fixture_loss(color, depth, alpha):
    H, W = color.shape[1:]
    wc, wd, wa = (t.to(color.device) for t in fixture_weights(H, W))
    return (color * wc).sum() + 0.1 * (depth * wd).sum() + (alpha * wa).sum()


def orbit(deg_y, deg_x, t):
    ay, ax = math.radians(deg_y), math.radians(deg_x)
    Ry = torch.tensor([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    w2c = torch.eye(4)
    w2c[:3, :3] = Rx @ Ry
    w2c[:3, 3] = torch.tensor(t)
    return w2c.t().contiguous()


def scenes():
    from rodygs_amd.synthetic import skewed_scene, synthetic_scene
    c1 = synthetic_scene(1000, 256, 256, 3, seed=2)
    sk = skewed_scene(256, 192, [(3, 4, 9500), (9, 2, 3000), (12, 9, 1500)], background=1500, sh_degree_max=1,
                      seed=31, equal_depth_every=7)
    sk["viewmatrix"] = orbit(1.5, -1.0, (0.05, -0.03, 0.1))
    return {"c1": (c1, 0, (0.0, 0.0, 0.0)), "skewed": (sk, 1, (0.1, 0.2, 0.3))}


def run_oracle(inp, deg, bg, H, W, tanx, tany, proj, cull=False):
    """inp: dict of float32 CPU tensors.  Returns (arrays dict) of everything the fixture stores."""
    from oracle import rasterizer_oracle as O
    ins = {k: inp[k].clone().requires_grad_(True) for k in INPUTS}
    P = ins["means3D"].shape[0]
    m2 = torch.zeros(P, 3, requires_grad=True)
    st = O.OracleSettings(H, W, tanx, tany, torch.tensor(bg), 1.0, proj, deg, cull=cull)
    color, depth, normal, alpha, radii, aux = O.rasterize(ins["means3D"], m2, ins["opacities"], ins["viewmatrix"], st,
                                                          shs=ins["shs"], scales=ins["scales"], rotations=ins["rotations"])
    fixture_loss(color, depth, alpha).backward()
    b, g = aux["binning"], aux["geom"]
    out = dict(radii=radii.numpy(), tiles_touched=g["tiles_touched"].numpy(), num_rendered=np.int64(b["num_rendered"]),
               keys_unsorted=b["keys_unsorted"], vals_unsorted=b["vals_unsorted"], keys_sorted=b["keys_sorted"],
               vals_sorted=b["vals_sorted"], ranges=b["ranges"], n_contrib=aux["n_contrib"].numpy(),
               final_T=aux["final_T"].detach().numpy(), color=color.detach().numpy(), depth=depth.detach().numpy(),
               normal=normal.detach().numpy(), alpha=alpha.detach().numpy(), grad_means2D=m2.grad.numpy())
    for k in INPUTS:
        out["grad_" + k] = ins[k].grad.numpy()
    return out


def main():
    torch.set_num_threads(1)       # one summation order, whatever box regenerates the files
    for name, (sc, deg, bg) in scenes().items():
        inp = {k: sc[k].to(torch.float32).contiguous() for k in INPUTS}
        out = run_oracle(inp, deg, bg, sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], sc["projmatrix"])
        store = {"in_" + k: v.numpy() for k, v in inp.items()}
        store.update(in_projmatrix=sc["projmatrix"].numpy(), in_tanfovx=np.float64(sc["tanfovx"]),
                     in_tanfovy=np.float64(sc["tanfovy"]), in_W=np.int64(sc["W"]), in_H=np.int64(sc["H"]),
                     in_sh_degree=np.int64(deg), in_bg=np.asarray(bg, dtype=np.float32))
        store.update(out)
        culled = run_oracle(inp, deg, bg, sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], sc["projmatrix"], cull=True)
        store.update({"cull_" + k: culled[k] for k in CULL_KEYS})
        path = os.path.join(OUT, f"rasterizer_golden_{name}.npz")
        np.savez_compressed(path, **store)
        r = out["ranges"].astype(np.int64)
        print(f"{name}: P={inp['means3D'].shape[0]} D={int(out['num_rendered'])} max tile list="
              f"{int((r[:, 1] - r[:, 0]).max())} max n_contrib={int(out['n_contrib'].max())} -> "
              f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
