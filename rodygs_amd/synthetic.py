"""Seeded synthetic inputs for the hot path: the clouds and cameras bench.py, the scripts and the
tests render.  Host-side, torch CPU tensors; nothing here touches the GPU or the oracle.

  * ``synthetic_scene``  -- the survey's generator: uniform in the frustum slab z in [2, 20], projected sigma chosen so
    that D (tile instances) is comparable to a densified scene (1 M @ 1080p: ~3.5 tiles per Gaussian).
  * ``skewed_scene``     -- the opposite of uniform: clusters of low-opacity Gaussians that all project into a few chosen
    tiles, so that single tiles hold thousands to hundreds of thousands of instances (real RoDyGS scenes densify every
    100 iterations, /root/reference/configs/train/train_kubric_mrig.yaml:168-173, and are nothing like uniform).
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814   # /root/reference/src/utils/sh_utils.py:24


def projection_matrix(znear, zfar, fovx, fovy):
    """Restatement of /root/reference/src/utils/graphic_utils.py:43-63 (pinned by tests/golden)."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (2 * right)
    Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def _camera(W, H, fovx_deg):
    fovx = math.radians(fovx_deg)
    focal = W / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(H / (2 * focal))
    return fovx, fovy, focal, math.tan(fovx / 2), math.tan(fovy / 2)


def synthetic_scene(P: int, W: int, H: int, sh_degree_max: int = 3, seed: int = 777, fovx_deg: float = 50.0,
                    device="cpu", variant: str = "uniform"):
    """Seeded synthetic cloud + camera of SURVEY.md §8d (uniform in the frustum slab z in [2,20]).

    ``variant`` (same draws, same camera; not the headline workload -- two regimes the survey's generator never enters):
      * "sheets": surface-like -- the Gaussians lie on eight fronto-parallel sheets (z = 4, 6, ... 18, +-1 %) with
        opacities 0.6-0.99 and twice the projected sigma, so a pixel saturates (T < 1e-4) within the first sheets and the
        compositing stops early: most of every tile list is never walked;
      * "dense": three times the projected sigma (D ~ 9x: >= 15 M tile instances at 1 M Gaussians / 1080p) with the
        survey's opacities -- a heavily densified scene."""
    if variant not in ("uniform", "sheets", "dense"):
        raise ValueError(f"unknown scene variant {variant!r}")
    g = torch.Generator().manual_seed(seed)
    fovx, fovy, focal, tanx, tany = _camera(W, H, fovx_deg)
    z = 2.0 + 18.0 * torch.rand(P, generator=g)
    if variant == "sheets":
        z = (4.0 + 2.0 * torch.floor((z - 2.0) / 18.0 * 8.0).clamp(0, 7)) * (1.0 + 0.01 * (2 * torch.rand(P, generator=g) - 1))
    x = (2 * torch.rand(P, generator=g) - 1) * 1.1 * z * tanx
    y = (2 * torch.rand(P, generator=g) - 1) * 1.1 * z * tany
    xyz = torch.stack([x, y, z], dim=1)
    sigma_px = min(max(2.0 * (1e6 / P) ** (1.0 / 3.0) * (W / 1920.0), 0.7), 8.0)
    sigma_px *= {"uniform": 1.0, "sheets": 2.0, "dense": 3.0}[variant]
    scales = (sigma_px * z / focal).unsqueeze(1) * torch.exp(0.5 * torch.randn(P, 3, generator=g))
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g))
    if variant == "sheets":
        opac = 0.6 + 0.39 * torch.rand(P, 1, generator=g)
    K = (sh_degree_max + 1) ** 2
    shs = torch.zeros(P, K, 3)
    shs[:, 0] = (torch.rand(P, 3, generator=g) - 0.5) / SH_C0
    if K > 1:
        shs[:, 1:] = 0.05 * torch.randn(P, K - 1, 3, generator=g)
    view = torch.eye(4)  # identity pose looking down +z
    proj = projection_matrix(0.01, 100.0, fovx, fovy)
    scene = dict(means3D=xyz, scales=scales, rotations=q, opacities=opac, shs=shs,
                 viewmatrix=view.t().contiguous(), projmatrix=proj.t().contiguous(),
                 tanfovx=tanx, tanfovy=tany, W=W, H=H, fovx=fovx, fovy=fovy)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in scene.items()}


def skewed_scene(W: int, H: int, clusters, background: int = 2000, sh_degree_max: int = 3, seed: int = 31,
                 fovx_deg: float = 50.0, sigma_px: float = 1.6, opacity=(0.006, 0.03), equal_depth_every: int = 0):
    """A cloud whose tile occupancy is as uneven as it gets.  ``clusters`` = [(tile_x, tile_y, count), ...]: `count`
    Gaussians whose centres project uniformly into the inner 10x10 pixels of that 16x16 tile with a projected sigma
    of ~``sigma_px`` px (radius <= 6 px: they stay inside the tile or reach one neighbour), at depths spread over
    [2, 20], with opacities in ``opacity`` -- low enough that compositing runs thousands of splats deep before a pixel
    saturates, high enough (>= 1/255 at the centre) that every splat can blend.  ``background`` further Gaussians come
    from ``synthetic_scene``.  ``equal_depth_every`` > 0 puts every k-th cluster Gaussian at the centre of its
    predecessor (the same view-space depth under any pose = ties in the sort key: the order must then fall back to the
    Gaussian index).  Identity camera, same conventions as
    ``synthetic_scene``; the Gaussians are shuffled so that a cluster is not a contiguous index range."""
    g = torch.Generator().manual_seed(seed)
    fovx, fovy, focal, tanx, tany = _camera(W, H, fovx_deg)
    base = synthetic_scene(max(background, 1), W, H, sh_degree_max, seed=seed + 1, fovx_deg=fovx_deg)
    xyz_l, sc_l, op_l = [], [], []
    for (tx, ty, count) in clusters:
        px = tx * 16 + 3.0 + 10.0 * torch.rand(count, generator=g)
        py = ty * 16 + 3.0 + 10.0 * torch.rand(count, generator=g)
        z = 2.0 + 18.0 * torch.rand(count, generator=g)
        if equal_depth_every > 0:
            idx = torch.arange(equal_depth_every, count, equal_depth_every)
            px[idx], py[idx], z[idx] = px[idx - 1], py[idx - 1], z[idx - 1]
        # pixel centre -> view space: px = ((ndc + 1) W - 1) / 2, ndc = x / (z tanx)
        x = ((2.0 * px + 1.0) / W - 1.0) * z * tanx
        y = ((2.0 * py + 1.0) / H - 1.0) * z * tany
        xyz_l.append(torch.stack([x, y, z], dim=1))
        s = (sigma_px * z / focal).unsqueeze(1) * torch.exp(0.15 * torch.randn(count, 3, generator=g))
        sc_l.append(s)
        op_l.append(opacity[0] + (opacity[1] - opacity[0]) * torch.rand(count, 1, generator=g))
    n_cl = sum(c for _, _, c in clusters)
    K = (sh_degree_max + 1) ** 2
    q = torch.randn(n_cl, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    shs = torch.zeros(n_cl, K, 3)
    shs[:, 0] = (torch.rand(n_cl, 3, generator=g) - 0.5) / SH_C0
    if K > 1:
        shs[:, 1:] = 0.05 * torch.randn(n_cl, K - 1, 3, generator=g)
    cat = dict(means3D=torch.cat(xyz_l + [base["means3D"][:background]]),
               scales=torch.cat(sc_l + [base["scales"][:background]]),
               rotations=torch.cat([q, base["rotations"][:background]]),
               opacities=torch.cat(op_l + [base["opacities"][:background]]),
               shs=torch.cat([shs, base["shs"][:background]]))
    perm = torch.randperm(cat["means3D"].shape[0], generator=g)
    scene = {k: v[perm].contiguous() for k, v in cat.items()}
    scene.update({k: base[k] for k in ("viewmatrix", "projmatrix", "tanfovx", "tanfovy", "W", "H", "fovx", "fovy")})
    return scene
