"""PyTorch-CPU differentiable Gaussian rasterizer -- the oracle (TEST INFRASTRUCTURE, never shipped).

PARITY UNPINNED: the reference's rasterizer source (``diff_gauss_pose``; reference ``.gitmodules:1-4``) is an
un-vendored submodule; no source, test or golden vector for it exists under ``/root/reference``.  This file
restates the published 3DGS tile rasterizer at the boundary the reference calls
(``/root/reference/src/trainer/renderer.py:50-101``) and is the normative spec of this build.  What it shares
with importable reference code is pinned by ``tests/golden`` (SH basis ``src/utils/sh_utils.py:24-101``,
projection ``src/utils/graphic_utils.py:43-63``, covariance ``src/model/rodygs_static.py:26-30`` +
``src/utils/general_utils.py:92-127``).

Numerical contract (mirrored 1:1 by ``rodygs_amd/csrc/rdg_preprocess.hip``):
  * everything is float32, evaluated with separate (un-fused) multiplies and adds, left to right exactly as
    written below -- torch CPU elementwise ops never contract to FMA, and the HIP preprocess kernel is built
    with ``-ffp-contract=off`` -- so view-space depth bits, radii and tile rectangles (hence tile keys and the
    sorted order) are BIT-EXACT between oracle and HIP;
  * the two square roots on that bit-exact path (the radius) go through ``_sqrt_rn``: the square root is taken in float64
    and rounded to float32, i.e. the correctly rounded float32 square root (what ``sqrtf`` is on the GPU and in CUDA's default
    build).  ``torch.sqrt`` on float32 CPU tensors is NOT that on every host: on the GPU boxes' hosts it returns
    0x1.2aaaaap+1 for sqrt(0x1.5c71c8p+2) where the correctly rounded value is 0x1.2aaaacp+1 (numpy, libm and the container
    this file was written in agree), and 3 sqrt(lambda) = 7.0000005 then became 7.0: radius 7 instead of 8, one Gaussian of
    one scene in 7 000 of scripts/parity_sweep.py (seed 300000, case 713);
  * the compositing stage differs from the HIP kernel only by ``exp`` rounding and summation order
    (tolerance 1e-4 relative, see tests).

Backward is torch autograd of this forward, with the two places where the public 3DGS backward is known to
deviate from the exact derivative reproduced explicitly (``_ste_min`` for the 0.99 alpha cap, and the
detached frustum clamp of ``t`` in the EWA Jacobian).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

def _sqrt_rn(x: torch.Tensor) -> torch.Tensor:
    """Correctly rounded square root in x's own precision (see the numerical contract above)."""
    return torch.sqrt(x.double()).to(x.dtype) if x.dtype == torch.float32 else torch.sqrt(x)


# ---- constants of the algorithm (SURVEY.md §7 "open questions" 6: 3DGS constants kept verbatim) -------------
NEAR_CULL = 0.2
FOV_CLAMP = 1.3
DILATION = 0.3
ALPHA_CAP = 0.99
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4
LAMBDA_FLOOR = 0.1
TILE = 16

# SH constants -- same values as /root/reference/src/utils/sh_utils.py:24-41
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


@dataclass
class OracleSettings:
    """Mirror of GaussianRasterizationSettings (reference call site src/trainer/renderer.py:50-63)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    projmatrix: torch.Tensor
    sh_degree: int
    prefiltered: bool = False
    debug: bool = False
    enable_cov_grad: bool = True
    enable_sh_grad: bool = True
    # Not a field of the reference's settings: which (tile, Gaussian) instances the per-Gaussian stage hands to the binning
    # stage -- RdgRasterSettings.cull of the C-ABI.  False = the reference algorithm's 3-sigma square (the key stream the
    # north_star's bit-exact bar names); True (the product's default) = that square intersected with the tiles holding a
    # pixel centre inside the box of the splat's alpha >= 1/255 ellipse (`_tight_rect`).  Images, final_T and gradients are
    # identical either way (tests/test_oracle_cull.py proves it on the CPU: the dropped instances blend nowhere).
    cull: bool = True


class _SteMin(torch.autograd.Function):
    """min(x, cap) forward, identity backward: public 3DGS backward ignores the 0.99 alpha cap."""

    @staticmethod
    def forward(ctx, x, cap):
        return torch.clamp(x, max=cap)

    @staticmethod
    def backward(ctx, g):
        return g, None


def _ste_min(x, cap):
    return _SteMin.apply(x, cap)


def eval_sh_rgb(deg: int, shs: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """SH -> RGB (before the +0.5 / clamp).  shs [P,K,3] (Gaussian, coeff, channel), d [P,3] unit dirs."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
                   + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * shs[:, 9]
                       + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13]
                       + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * shs[:, 15])
    return res


def rotation_from_raw_quat(q: torch.Tensor):
    """R(q) WITHOUT normalisation.  Same polynomial as
    /root/reference/src/utils/general_utils.py:92-113 minus the normalise.  Returns 9 tensors R00..R22."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R00 = 1.0 - 2.0 * (y * y + z * z)
    R01 = 2.0 * (x * y - r * z)
    R02 = 2.0 * (x * z + r * y)
    R10 = 2.0 * (x * y + r * z)
    R11 = 1.0 - 2.0 * (x * x + z * z)
    R12 = 2.0 * (y * z - r * x)
    R20 = 2.0 * (x * z - r * y)
    R21 = 2.0 * (y * z + r * x)
    R22 = 1.0 - 2.0 * (x * x + y * y)
    return R00, R01, R02, R10, R11, R12, R20, R21, R22


def covariance3d(scales: torch.Tensor, scale_modifier: float, rotations: torch.Tensor) -> torch.Tensor:
    """Sigma = (R S)(R S)^T as the 6-vector xx,xy,xz,yy,yz,zz (order of general_utils.py:77-89)."""
    s = scale_modifier * scales
    R = rotation_from_raw_quat(rotations)
    L = [[R[3 * i + j] * s[:, j] for j in range(3)] for i in range(3)]

    def dot(a, b):
        return (L[a][0] * L[b][0] + L[a][1] * L[b][1]) + L[a][2] * L[b][2]

    return torch.stack([dot(0, 0), dot(0, 1), dot(0, 2), dot(1, 1), dot(1, 2), dot(2, 2)], dim=1)


def f32(x: float) -> float:
    """Round a python double to float32 (the value the C-ABI receives as a `float` argument)."""
    return float(np.float32(x))


def _ln_f32(x: np.ndarray) -> np.ndarray:
    """ln of positive normal float32 numbers from +, -, *, / and bit operations only, in the order of
    rdg_ln_exact_ops (csrc/rdg_preprocess_fwd.hip): the same bits on both sides."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    e = (u >> np.uint32(23)).astype(np.int32) - np.int32(127)
    m = ((u & np.uint32(0x007fffff)) | np.uint32(0x3f800000)).view(np.float32)
    big = m > np.float32(1.41421354)
    m = np.where(big, m * np.float32(0.5), m)
    e = np.where(big, e + 1, e)
    s = (m - np.float32(1.0)) / (m + np.float32(1.0))
    s2 = s * s
    p = np.float32(0.111111111) * s2 + np.float32(0.142857143)
    p = p * s2 + np.float32(0.2)
    p = p * s2 + np.float32(0.333333333)
    p = p * s2 + np.float32(1.0)
    return (np.float32(2.0) * s) * p + e.astype(np.float32) * np.float32(0.693147181)


def _tight_rect(px, py, conic_a, conic_b, inv_cyy, opacity, rect, gx, gy):
    """RdgRasterSettings.cull = 1, restated (rdg_splat_rect, csrc/rdg_preprocess_fwd.hip; float32: operation for
    operation): the reference rectangle intersected with the tiles that hold a pixel centre inside the axis-aligned box of the
    splat's alpha >= 1/255 ellipse.  The quadratic form the compositing evaluates is Q = a (dx + beta dy)^2 + dy^2 / cov_yy;
    a pixel blends only where Q <= 2 ln(255 opacity); over dx Q >= dy^2 / cov_yy, over dy Q >= dx^2 / cov_xx, cov_xx =
    c_eff cov_yy / a with c_eff = b^2 / a + 1 / cov_yy.  The bound carries a margin (0.02 in the log, 1e-3 relative).
    numpy arrays in, (x0, y0, x1, y1) int64 arrays out; a splat that cannot reach 1/255 gets an empty rectangle."""
    x0, y0, x1, y1 = [np.array(r, dtype=np.int64) for r in rect]
    f32 = px.dtype == np.float32
    one, two = px.dtype.type(1.0), px.dtype.type(2.0)
    with np.errstate(all="ignore"):
        t255 = px.dtype.type(255.0) * opacity
        dead = ~(t255 >= px.dtype.type(0.99))
        tl = np.where(dead, one, t255)
        ln = _ln_f32(tl) if f32 else np.log(tl)
        r2 = (two * (ln + px.dtype.type(0.02))) * px.dtype.type(1.001)
        cyy = one / inv_cyy
        c_eff = (conic_b * conic_b) / conic_a + inv_cyy
        cxx = (c_eff * cyy) / conic_a
        hx, hy = np.sqrt(r2 * cxx), np.sqrt(r2 * cyy)
        # fmaxf / fminf return the other operand for a NaN: a NaN extent keeps the reference rectangle
        lox = np.fmax(np.ceil(px - hx), px.dtype.type(0.0)); hix = np.fmin(np.floor(px + hx), px.dtype.type(gx * TILE - 1))
        loy = np.fmax(np.ceil(py - hy), px.dtype.type(0.0)); hiy = np.fmin(np.floor(py + hy), px.dtype.type(gy * TILE - 1))
        empty = dead | (hix < lox) | (hiy < loy)
        i = lambda v: np.nan_to_num(v, nan=0.0, posinf=2.0e9, neginf=-2.0e9).astype(np.int64)
        tx0, tx1 = i(lox) >> 4, (i(hix) >> 4) + 1
        ty0, ty1 = i(loy) >> 4, (i(hiy) >> 4) + 1
    nx0, nx1 = np.maximum(x0, tx0), np.minimum(x1, tx1)
    ny0, ny1 = np.maximum(y0, ty0), np.minimum(y1, ty1)
    nx1, ny1 = np.maximum(nx1, nx0), np.maximum(ny1, ny0)
    nx1 = np.where(empty, nx0, nx1)
    ny1 = np.where(empty, ny0, ny1)
    return nx0, ny0, nx1, ny1


def preprocess(means3D, means2D, opacities, viewmatrix, settings: OracleSettings, shs=None, colors_precomp=None,
               scales=None, rotations=None, cov3Ds_precomp=None):
    """Per-Gaussian stage.  All tensors float32 (or float64 for derivative studies)."""
    dt = means3D.dtype
    H, W = int(settings.image_height), int(settings.image_width)
    P = means3D.shape[0]
    cast = f32 if dt == torch.float32 else float
    tanx, tany = cast(settings.tanfovx), cast(settings.tanfovy)
    # the C-ABI receives tanfov as f32 and derives the focal lengths from that value in double
    focal_x = cast(W / (2.0 * tanx))
    focal_y = cast(H / (2.0 * tany))
    V = viewmatrix.reshape(16)
    Pm = settings.projmatrix.to(dt).reshape(16)
    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]

    # view transform (glm/column-major storage: flat[c*4+r] = M[r][c])
    vx = ((V[0] * x + V[4] * y) + V[8] * z) + V[12]
    vy = ((V[1] * x + V[5] * y) + V[9] * z) + V[13]
    vz = ((V[2] * x + V[6] * y) + V[10] * z) + V[14]
    valid = vz > NEAR_CULL

    hx = ((Pm[0] * vx + Pm[4] * vy) + Pm[8] * vz) + Pm[12]
    hy = ((Pm[1] * vx + Pm[5] * vy) + Pm[9] * vz) + Pm[13]
    hw = ((Pm[3] * vx + Pm[7] * vy) + Pm[11] * vz) + Pm[15]
    pw = 1.0 / (hw + 1e-7)
    # means2D is the zero "grad sink" of renderer.py:38-44; its gradient is dL/d(ndc) = dL/dpix * (W/2, H/2)
    ndc_x = hx * pw + means2D[:, 0]
    ndc_y = hy * pw + means2D[:, 1]

    # 3-D covariance
    if cov3Ds_precomp is not None:
        cov3D = cov3Ds_precomp
    else:
        cov3D = covariance3d(scales, cast(settings.scale_modifier), rotations)
    S00, S01, S02, S11, S12, S22 = [cov3D[:, i] for i in range(6)]

    # EWA 2-D covariance.  enable_cov_grad gates the viewmatrix gradient through this block only.
    Vc = V if settings.enable_cov_grad else V.detach()
    if settings.enable_cov_grad:
        cvx, cvy, cvz = vx, vy, vz
    else:
        xd, yd, zd = x, y, z
        cvx = ((Vc[0] * xd + Vc[4] * yd) + Vc[8] * zd) + Vc[12]
        cvy = ((Vc[1] * xd + Vc[5] * yd) + Vc[9] * zd) + Vc[13]
        cvz = ((Vc[2] * xd + Vc[6] * yd) + Vc[10] * zd) + Vc[14]
    limx = cast(np.float32(FOV_CLAMP) * np.float32(tanx)) if dt == torch.float32 else FOV_CLAMP * tanx
    limy = cast(np.float32(FOV_CLAMP) * np.float32(tany)) if dt == torch.float32 else FOV_CLAMP * tany
    txtz = cvx / cvz
    tytz = cvy / cvz
    cl_x = (txtz < -limx) | (txtz > limx)
    cl_y = (tytz < -limy) | (tytz > limy)
    tx_c = torch.clamp(txtz, min=-limx, max=limx) * cvz
    ty_c = torch.clamp(tytz, min=-limy, max=limy) * cvz
    # public 3DGS backward treats a clamped t.x / t.y as a constant
    tx = torch.where(cl_x, tx_c.detach(), tx_c)
    ty = torch.where(cl_y, ty_c.detach(), ty_c)
    # NB: python_scalar / tensor is reciprocal()*scalar in torch (two roundings) -- divide by a 0-dim tensor
    fx_t, fy_t = torch.tensor(focal_x, dtype=dt), torch.tensor(focal_y, dtype=dt)
    J00 = fx_t / cvz
    J02 = -(fx_t * tx) / (cvz * cvz)
    J11 = fy_t / cvz
    J12 = -(fy_t * ty) / (cvz * cvz)
    # W = rotation rows of W2C: W[i][j] = flat[j*4+i]
    T0 = [J00 * Vc[4 * j + 0] + J02 * Vc[4 * j + 2] for j in range(3)]
    T1 = [J11 * Vc[4 * j + 1] + J12 * Vc[4 * j + 2] for j in range(3)]
    Sg = [[S00, S01, S02], [S01, S11, S12], [S02, S12, S22]]
    u0 = [(T0[0] * Sg[0][j] + T0[1] * Sg[1][j]) + T0[2] * Sg[2][j] for j in range(3)]
    u1 = [(T1[0] * Sg[0][j] + T1[1] * Sg[1][j]) + T1[2] * Sg[2][j] for j in range(3)]
    ca = ((u0[0] * T0[0] + u0[1] * T0[1]) + u0[2] * T0[2]) + DILATION
    cb = (u0[0] * T1[0] + u0[1] * T1[1]) + u0[2] * T1[2]
    cc = ((u1[0] * T1[0] + u1[1] * T1[1]) + u1[2] * T1[2]) + DILATION
    det = ca * cc - cb * cb
    valid = valid & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    det_inv = 1.0 / det_safe
    conic_a = cc * det_inv
    conic_b = -cb * det_inv
    conic_c = ca * det_inv
    with torch.no_grad():
        mid = 0.5 * (ca + cc)
        disc = _sqrt_rn(torch.clamp(mid * mid - det, min=LAMBDA_FLOOR))
        lam = torch.maximum(mid + disc, mid - disc)
        radius_f = torch.ceil(3.0 * _sqrt_rn(lam))
    px = ((ndc_x + 1.0) * W - 1.0) * 0.5
    py = ((ndc_y + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    with torch.no_grad():
        rf = radius_f.to(dt)
        pxd, pyd = px.detach(), py.detach()

        def tr(v):  # (int) truncation toward zero, done in float64-safe int64
            return torch.trunc(torch.nan_to_num(v, nan=0.0, posinf=2.0e9, neginf=-2.0e9)).to(torch.int64)

        rminx = torch.clamp(tr((pxd - rf) / TILE), 0, gx)
        rminy = torch.clamp(tr((pyd - rf) / TILE), 0, gy)
        rmaxx = torch.clamp(tr((((pxd + rf) + TILE) - 1.0) / TILE), 0, gx)
        rmaxy = torch.clamp(tr((((pyd + rf) + TILE) - 1.0) / TILE), 0, gy)
        tiles = (rmaxx - rminx) * (rmaxy - rminy)
        valid = valid & (tiles > 0)
        tiles = torch.where(valid, tiles, torch.zeros_like(tiles))
        # visibility (radii > 0, /root/reference/src/trainer/renderer.py:111) is the reference rule's, whatever is binned
        radii = torch.where(valid, radius_f.to(torch.int64), torch.zeros_like(tiles)).to(torch.int32)
        if getattr(settings, "cull", False):
            inv_cyy = (torch.ones_like(cc) / cc).detach()
            t = _tight_rect(pxd.numpy(), pyd.numpy(), conic_a.detach().numpy(), conic_b.detach().numpy(), inv_cyy.numpy(),
                            opacities.detach().reshape(P).to(dt).numpy(),
                            (rminx.numpy(), rminy.numpy(), rmaxx.numpy(), rmaxy.numpy()), gx, gy)
            rminx, rminy, rmaxx, rmaxy = [torch.from_numpy(np.ascontiguousarray(v)) for v in t]
            tiles = torch.where(valid, (rmaxx - rminx) * (rmaxy - rminy), torch.zeros_like(tiles))

    # colour
    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        Vs = V if settings.enable_sh_grad else V.detach()
        camx = -((Vs[0] * Vs[12] + Vs[1] * Vs[13]) + Vs[2] * Vs[14])
        camy = -((Vs[4] * Vs[12] + Vs[5] * Vs[13]) + Vs[6] * Vs[14])
        camz = -((Vs[8] * Vs[12] + Vs[9] * Vs[13]) + Vs[10] * Vs[14])
        dx, dy, dz = x - camx, y - camy, z - camz
        ln = torch.sqrt((dx * dx + dy * dy) + dz * dz)
        d = torch.stack([dx / ln, dy / ln, dz / ln], dim=1)
        raw = eval_sh_rgb(int(settings.sh_degree), shs, d) + 0.5
        clamped = raw < 0
        rgb = torch.clamp(raw, min=0.0)

    # normal (SURVEY.md §7 open question 2): shortest axis of R*diag(s) in view space, facing the camera.
    if cov3Ds_precomp is None:
        with torch.no_grad():
            k = torch.argmin(scales, dim=1)
        R = rotation_from_raw_quat(rotations)
        Rm = torch.stack(R, dim=1).reshape(P, 3, 3)
        n = torch.gather(Rm, 2, k.view(P, 1, 1).expand(P, 3, 1)).squeeze(2)
        Vn = V.detach()
        nv = [(Vn[0 + i] * n[:, 0] + Vn[4 + i] * n[:, 1]) + Vn[8 + i] * n[:, 2] for i in range(3)]
        dotv = (nv[0] * vx.detach() + nv[1] * vy.detach()) + nv[2] * vz.detach()
        sgn = torch.where(dotv > 0, -torch.ones_like(dotv), torch.ones_like(dotv))
        normal = torch.stack([nv[0] * sgn, nv[1] * sgn, nv[2] * sgn], dim=1).detach()
    else:
        normal = torch.zeros(P, 3, dtype=dt)

    return dict(valid=valid, depth=vz, px=px, py=py, conic=torch.stack([conic_a, conic_b, conic_c], dim=1),
                opacity=opacities.reshape(P), rgb=rgb, clamped=clamped, normal=normal, radii=radii,
                tiles_touched=tiles.to(torch.int32), rect=(rminx, rminy, rmaxx, rmaxy), cov3D=cov3D,
                cov2D=torch.stack([ca, cb, cc], dim=1), grid=(gx, gy), vx=vx, vy=vy)


def bin_and_sort(geom):
    """Scan + duplicateWithKeys + stable sort + tile ranges -- integer, numpy."""
    gx, gy = geom["grid"]
    tiles = geom["tiles_touched"].numpy().astype(np.int64)
    rminx, rminy, rmaxx, rmaxy = [r.numpy().astype(np.int64) for r in geom["rect"]]
    depth_bits = geom["depth"].detach().to(torch.float32).numpy().view(np.uint32).astype(np.uint64)
    offsets = np.cumsum(tiles)
    D = int(offsets[-1]) if len(offsets) else 0
    ids = np.repeat(np.arange(len(tiles), dtype=np.int64), tiles)
    starts = offsets - tiles
    j = np.arange(D, dtype=np.int64) - starts[ids]
    w = (rmaxx - rminx)[ids]
    ty = rminy[ids] + j // np.maximum(w, 1)
    tx = rminx[ids] + j % np.maximum(w, 1)
    tile_id = (ty * gx + tx).astype(np.uint64)
    keys = (tile_id << np.uint64(32)) | depth_bits[ids]
    vals = ids.astype(np.uint32)
    order = np.argsort(keys, kind="stable")
    keys_sorted = keys[order]
    vals_sorted = vals[order]
    tile_sorted = (keys_sorted >> np.uint64(32)).astype(np.int64)
    ntiles = gx * gy
    start = np.searchsorted(tile_sorted, np.arange(ntiles), side="left")
    end = np.searchsorted(tile_sorted, np.arange(ntiles), side="right")
    ranges = np.stack([start, end], axis=1).astype(np.uint32)
    # tiles with no splats are left at (0,0) by the kernel (never written)
    ranges[start == end] = 0
    return dict(offsets=offsets.astype(np.uint32), num_rendered=D, keys_unsorted=keys, vals_unsorted=vals,
                keys_sorted=keys_sorted, vals_sorted=vals_sorted, ranges=ranges)


def _composite_group(a, pixx, pixy, bgc):
    """Front-to-back compositing of G tiles at once.  a [G,L,13+E] = (px, py, conic a b c, opacity, 7 features, E extra
    attributes) per list slot (padding slots: all zero -> alpha 0 -> never blended), pixx / pixy [G,256] pixel coordinates.
    Returns (values [G,8+E,256] = colour(+bg) 3, depth 1, normal 3, extra E, alpha 1;  final_T [G,256];  n_contrib [G,256])."""
    dt = a.dtype
    dx = a[:, :, 0:1] - pixx.unsqueeze(1)
    dy = a[:, :, 1:2] - pixy.unsqueeze(1)
    power = -0.5 * (a[:, :, 2:3] * dx * dx + a[:, :, 4:5] * dy * dy) - a[:, :, 3:4] * dx * dy
    ok = power <= 0
    G = torch.exp(torch.where(ok, power, torch.zeros_like(power)))
    alpha = _ste_min(a[:, :, 5:6] * G, ALPHA_CAP)
    ok = ok & (alpha >= ALPHA_MIN)
    a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
    cum = torch.cumprod(1.0 - a_eff, dim=1)
    inc = ok & (cum >= T_STOP)
    T_before = torch.cat([torch.ones(cum.shape[0], 1, cum.shape[2], dtype=dt), cum[:, :-1]], dim=1)
    wgt = torch.where(inc, a_eff * T_before, torch.zeros_like(a_eff))
    acc = torch.einsum("gnp,gnc->gcp", wgt, a[:, :, 6:])
    Tf = torch.prod(torch.where(inc, 1.0 - a_eff, torch.ones_like(a_eff)), dim=1)
    vals = torch.cat([acc[:, 0:3] + Tf.unsqueeze(1) * bgc.view(1, 3, 1), acc[:, 3:], (1.0 - Tf).unsqueeze(1)], dim=1)
    with torch.no_grad():
        idx = torch.arange(1, inc.shape[1] + 1, dtype=torch.int32).view(1, -1, 1)
        ncon = (inc.to(torch.int32) * idx).max(dim=1).values
    return vals, Tf.detach(), ncon


def render_tiles(geom, binning, bg, H, W, tile_subset=None, rows_per_group=8192):
    """Per-tile front-to-back compositing, vectorised over [tiles, list slots, 256 pixels].

    Tiles are processed in groups of similar list length (padded with all-zero slots to the group's longest list,
    at most ``rows_per_group`` slots per group); the per-instance attributes are gathered once per group and every
    group runs under activation checkpointing, so the autograd graph of a 1080p frame holds the gathered rows only
    (a per-tile Python loop that kept every [L,256] intermediate alive needed ~20 GB at 100 k Gaussians / 1080p and
    spent its time in per-tile framework overhead).  Pixels of ragged border tiles outside the image are computed
    and dropped."""
    from torch.utils.checkpoint import checkpoint
    dt = geom["px"].dtype
    gx, gy = geom["grid"]
    P = geom["px"].shape[0]
    vals = binning["vals_sorted"].astype(np.int64)
    ranges = binning["ranges"].astype(np.int64)
    bgc = bg.to(dt).reshape(3)
    tiles = np.asarray(list(range(gx * gy) if tile_subset is None else tile_subset), dtype=np.int64)
    E = 0 if geom.get("extra") is None else int(geom["extra"].shape[1])
    out = torch.zeros(8 + E, H * W, dtype=dt)
    final_T = torch.ones(H * W, dtype=dt)
    n_contrib = torch.zeros(H * W, dtype=torch.int32)
    if len(tiles):
        lens = ranges[tiles, 1] - ranges[tiles, 0]
        ly, lx = np.divmod(np.arange(256, dtype=np.int64), 16)
        py_all = (tiles // gx)[:, None] * TILE + ly[None, :]
        px_all = (tiles % gx)[:, None] * TILE + lx[None, :]
        # empty tiles: background colour, alpha 0 (one vectorised write)
        emp = lens == 0
        if emp.any():
            inside = (px_all[emp] < W) & (py_all[emp] < H)
            flat = torch.from_numpy((py_all[emp] * W + px_all[emp])[inside])
            bgv = torch.zeros(8 + E, flat.numel(), dtype=dt)
            bgv[0:3] = bgc.view(3, 1)
            out = out.index_copy(1, flat, bgv)
        attrs = torch.cat([geom["px"].unsqueeze(1), geom["py"].unsqueeze(1), geom["conic"],
                           geom["opacity"].unsqueeze(1), geom["rgb"], geom["depth"].unsqueeze(1), geom["normal"],
                           ] + ([geom["extra"].to(dt)] if E else []), dim=1)                # [P, 6 + 7 + E]
        attrs = torch.cat([attrs, torch.zeros(1, attrs.shape[1], dtype=dt)], dim=0)        # row P = padding slot
        order = np.argsort(lens, kind="stable")
        order = order[lens[order] > 0]
        pix_idx, pix_val = [], []
        i = 0
        while i < len(order):
            # the group's longest list is its last (sorted by length): grow while (tiles x longest list) fits
            j = i + 1
            while j < len(order) and (j + 1 - i) * int(lens[order[j]]) <= rows_per_group:
                j += 1
            sel = order[i:j]
            L = int(lens[sel].max())
            idx = np.full((len(sel), L), P, dtype=np.int64)
            for r, k in enumerate(sel):
                s0 = int(ranges[tiles[k], 0])
                idx[r, :lens[k]] = vals[s0:s0 + lens[k]]
            a = attrs[torch.from_numpy(idx)]                                                # [G, L, 13]
            pxx = torch.from_numpy(px_all[sel]).to(dt)
            pyy = torch.from_numpy(py_all[sel]).to(dt)
            if a.requires_grad:
                v, Tf, nc = checkpoint(_composite_group, a, pxx, pyy, bgc, use_reentrant=False)
            else:
                v, Tf, nc = _composite_group(a, pxx, pyy, bgc)
            inside = torch.from_numpy((px_all[sel] < W) & (py_all[sel] < H))
            flat = torch.from_numpy(py_all[sel] * W + px_all[sel])[inside]
            pix_idx.append(flat)
            pix_val.append(v.permute(1, 0, 2)[:, inside])                                  # [8 + E, n_inside]
            with torch.no_grad():
                final_T[flat] = Tf[inside]
                n_contrib[flat] = nc[inside]
            i = j
        if pix_idx:
            out = out.index_copy(1, torch.cat(pix_idx), torch.cat(pix_val, dim=1))
    out = out.reshape(8 + E, H, W)
    return dict(color=out[0:3], depth=out[3:4], normal=out[4:7], extra=out[7:7 + E], alpha=out[7 + E:8 + E],
                final_T=final_T.reshape(H, W), n_contrib=n_contrib.reshape(H, W))


def rasterize(means3D, means2D, opacities, viewmatrix, settings: OracleSettings, shs=None, colors_precomp=None,
              scales=None, rotations=None, cov3Ds_precomp=None, tile_subset=None, extra_attrs=None):
    """Full oracle forward.  Returns (color, depth, normal, alpha, radii, aux).  extra_attrs [P,E] (the upstream kwarg no
    reference caller passes; assumed semantics: composited with the colour's weights, no background term) -> aux["extra"]
    [E,H,W]."""
    geom = preprocess(means3D, means2D, opacities, viewmatrix, settings, shs=shs, colors_precomp=colors_precomp,
                      scales=scales, rotations=rotations, cov3Ds_precomp=cov3Ds_precomp)
    if extra_attrs is not None:
        geom["extra"] = extra_attrs
    binning = bin_and_sort(geom)
    img = render_tiles(geom, binning, settings.bg, int(settings.image_height), int(settings.image_width),
                       tile_subset=tile_subset)
    aux = dict(geom=geom, binning=binning, final_T=img["final_T"], n_contrib=img["n_contrib"], extra=img["extra"])
    return img["color"], img["depth"], img["normal"], img["alpha"], geom["radii"], aux


# ---- synthetic scene generators live in the product package (bench.py's inputs must not come from
# test infrastructure); re-exported here so the tests keep one import ---------------------------------------------
from rodygs_amd.synthetic import projection_matrix, skewed_scene, synthetic_scene  # noqa: E402,F401
