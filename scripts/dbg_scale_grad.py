"""CPU emulation (numpy float32) of steps (3)-(4) of rdg_preprocess_bwd.hip -- dL/dconic -> dL/dscale through the projected
axes -- on a sweep case, fed the float64 oracle's dL/dconic rounded to float32, against the float64 oracle's dL/dscales.
Shows which Gaussians decide a column's error and which intermediate loses the digits.
usage: [RDG_SWEEP_PROFILE=aniso] python scripts/dbg_scale_grad.py <seed0> <case> [variant]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rasterizer_oracle as O  # noqa: E402
from sweep_cases import sweep_case, sweep_case_aniso  # noqa: E402

NAMES = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
seed0, c = int(sys.argv[1]), int(sys.argv[2])
variant = sys.argv[3] if len(sys.argv) > 3 else "axes"
sc, deg, bg, kw = (sweep_case_aniso if os.environ.get("RDG_SWEEP_PROFILE") == "aniso" else sweep_case)(seed0, c)
P, H, W = sc["means3D"].shape[0], sc["H"], sc["W"]
gen = torch.Generator().manual_seed(kw["seed"])
wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
wn = torch.randn(3, H, W, generator=gen) * kw["normal_loss"]


def run(dtype):
    d = {k: sc[k].clone().to(dtype).requires_grad_(True) for k in NAMES}
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor(bg).to(dtype), kw["scale_modifier"],
                          sc["projmatrix"].to(dtype), deg, enable_cov_grad=kw["cov_grad"], enable_sh_grad=kw["sh_grad"])
    o = O.rasterize(d["means3D"], torch.zeros(P, 3, dtype=dtype, requires_grad=True), d["opacities"], d["viewmatrix"], st,
                    shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
    geom = o[5]["geom"]
    geom["conic"].retain_grad()
    ls = (o[0] * wc.to(dtype)).sum() + (o[3] * wa.to(dtype)).sum()
    if kw["depth_loss"]:
        ls = ls + (o[1] * wd.to(dtype)).sum() * kw["depth_loss"]
    if kw["normal_loss"]:
        ls = ls + (o[2] * wn.to(dtype)).sum()
    ls.backward()
    return d, geom, o


d64, g64, o64 = run(torch.float64)
d32, g32, o32 = run(torch.float32)
ref = d64["scales"].grad.numpy()
col_scale = np.abs(ref).max(0)
print("P", P, W, H, kw, "column scales", col_scale)
print("float32 oracle per column:", np.abs(d32["scales"].grad.numpy() - ref).max(0) / col_scale)

f = np.float32
gk64 = g64["conic"].grad.numpy()
gk = gk64.astype(f)              # dL/d(conic a, b, c) float64-exact, rounded
# the row format of the compositing backward: moments about (w, dy), w = dx + beta dy, beta = conic_b / conic_a (float32 conic)
con32 = g32["conic"].detach().numpy().astype(f)
beta = np.where(con32[:, 0] > 0, con32[:, 1] / con32[:, 0], 0).astype(f)
b64 = beta.astype(np.float64)
Gww = gk64[:, 0] + b64 * gk64[:, 1] + b64 * b64 * gk64[:, 2]
Gwy = gk64[:, 1] + 2 * b64 * gk64[:, 2]
Gyy = gk64[:, 2]
noise = float(os.environ.get("NOISE", "0"))
rng = np.random.default_rng(5)
rows = np.stack([Gww, Gwy, Gyy], 1) * (1 + noise * rng.standard_normal((P, 3)))
rows = rows.astype(f)                                  # what the kernel reads (with the accumulation noise of the sums)
vis = (o64[4].numpy() > 0)
V = sc["viewmatrix"].numpy().astype(f).reshape(-1)
smod = f(kw["scale_modifier"])
fx, fy = f(W / (2 * sc["tanfovx"])), f(H / (2 * sc["tanfovy"]))
limx, limy = f(1.3 * sc["tanfovx"]), f(1.3 * sc["tanfovy"])
out = np.zeros((P, 3), f)
for i in range(P):
    if not vis[i]:
        continue
    x, y, z = sc["means3D"][i].numpy().astype(f)
    vx = V[0] * x + V[4] * y + V[8] * z + V[12]; vy = V[1] * x + V[5] * y + V[9] * z + V[13]; vz = V[2] * x + V[6] * y + V[10] * z + V[14]
    s0, s1, s2 = (smod * sc["scales"][i].numpy().astype(f))
    qr, qx, qy, qz = sc["rotations"][i].numpy().astype(f)
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qr * qz), 2 * (qx * qz + qr * qy)],
                  [2 * (qx * qy + qr * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qr * qx)],
                  [2 * (qx * qz - qr * qy), 2 * (qy * qz + qr * qx), 1 - 2 * (qx * qx + qy * qy)]], f)
    tx = min(limx, max(-limx, vx / vz)) * vz; ty = min(limy, max(-limy, vy / vz)) * vz
    iz = f(1) / vz; iz2 = iz * iz
    J00, J02, J11, J12 = fx * iz, -fx * tx * iz2, fy * iz, -fy * ty * iz2
    Wm = np.array([[V[0], V[4], V[8]], [V[1], V[5], V[9]], [V[2], V[6], V[10]]], f)
    T0 = J00 * Wm[0] + J02 * Wm[2]; T1 = J11 * Wm[1] + J12 * Wm[2]
    ax = (T0[:, None] * R).sum(0).astype(f); ay = (T1[:, None] * R).sum(0).astype(f)      # a_kx = T0 . r_k
    q = np.array([s0 * s0, s1 * s1, s2 * s2], f)
    n = ax * ax + ay * ay
    x01 = ax[0] * ay[1] - ay[0] * ax[1]; x02 = ax[0] * ay[2] - ay[0] * ax[2]; x12 = ax[1] * ay[2] - ay[1] * ax[2]
    DIL = f(0.3)
    detp = ((q[0] * q[1]) * (x01 * x01) + (q[0] * q[2]) * (x02 * x02) + (q[1] * q[2]) * (x12 * x12)) + DIL * (q * n).sum(dtype=f) + DIL * DIL
    di = f(1) / detp
    gww, gwy, gyy = rows[i]
    bs = beta[i]
    if variant.startswith("skew"):
        # everything straight from the skew moments: p_k = a_kx + beta a_ky
        p = ax + bs * ay
        hw = f(0.5) * gwy
        ex = p * gyy - ay * hw
        ey = bs * ex + (ay * gww - p * hw)
        B = p * ex + ay * (ay * gww - p * hw)
        gca = gww; gcc = gyy
        trG = gww - bs * gwy + (f(1) + bs * bs) * gyy
    else:
        gcc = gyy
        gcb = np.float32(np.float32(-2.0) * bs * gcc + gwy)
        gca = np.float32(bs * (bs * gcc - gwy) + gww)
        hb = f(0.5) * gcb
        ex = gcc * ax - hb * ay; ey = gca * ay - hb * ax
        B = ax * ex + ay * ey
        trG = gca + gcc
    A = np.array([(q[1] * (x01 * x01) + q[2] * (x02 * x02)) + DIL * n[0], (q[0] * (x01 * x01) + q[2] * (x12 * x12)) + DIL * n[1],
                  (q[0] * (x02 * x02) + q[1] * (x12 * x12)) + DIL * n[2]], f)
    if variant in ("quad", "quadres"):
        # dL/dcov2D = -K Gm K (K = conic): u^T G u = -(K u)^T Gm (K u), ONE quadratic form, in the skew basis:
        # v = K u, v.d = vx w + r dy, r = vy - beta vx = uy / cov2D_yy
        cx = np.array([(q[1] * x01) * ay[1] + (q[2] * x02) * ay[2] + DIL * ax[0],
                       -(q[0] * x01) * ay[0] + (q[2] * x12) * ay[2] + DIL * ax[1],
                       -(q[0] * x02) * ay[0] - (q[1] * x12) * ay[1] + DIL * ax[2]], f)
        ccov = (q * ay * ay).sum(dtype=f) + DIL
        ic = f(1) / ccov
        vx = di * cx
        r = ay * ic
        if variant == "quadres":
            rka, rkb = con32[i, 0], con32[i, 1]
            resa = np.float32(np.float64(rkb) - np.float64(rka) * np.float64(bs))     # fma(-rka, beta, rkb)
            r = r + (resa / rka) * vx
        mww, mwy, myy = gww, f(0.5) * gwy, gyy
        hw = vx * mww + r * mwy
        hy = vx * mwy + r * myy
        out[i] = smod * (2 * np.array([s0, s1, s2], f)) * (-(vx * hw + r * hy))
    elif variant in ("axes", "skew"):
        ddi = (q * B).sum(dtype=f) + DIL * trG
        ddet = -(ddi * di) * di
        out[i] = smod * (2 * np.array([s0, s1, s2], f)) * (di * B + ddet * A)
    elif variant in ("split", "skewsplit"):
        # u_k^T G u_k = di^2 (B_k D_k - A_k E_k): the axis' own q_k A_k B_k term of B_k det - A_k ddi dropped analytically
        D = detp - q * A
        Dk = np.array([(q[1] * q[2]) * (x12 * x12) + DIL * (q[1] * n[1] + q[2] * n[2]) + DIL * DIL,
                       (q[0] * q[2]) * (x02 * x02) + DIL * (q[0] * n[0] + q[2] * n[2]) + DIL * DIL,
                       (q[0] * q[1]) * (x01 * x01) + DIL * (q[0] * n[0] + q[1] * n[1]) + DIL * DIL], f)
        tr = DIL * trG
        Ek = np.array([q[1] * B[1] + q[2] * B[2] + tr, q[0] * B[0] + q[2] * B[2] + tr, q[0] * B[0] + q[1] * B[1] + tr], f)
        out[i] = smod * (2 * np.array([s0, s1, s2], f)) * ((di * di) * (B * Dk - A * Ek))
err = np.abs(out.astype(np.float64) - ref)
print(f"variant {variant}: per column", err.max(0) / col_scale)
for col in range(3):
    j = int(err[:, col].argmax())
    print(f" col {col}: worst Gaussian {j}: emu {out[j, col]:.6e} ref {ref[j, col]:.6e} f32-oracle {d32['scales'].grad[j, col]:.6e} "
          f"scales {sc['scales'][j].numpy()} radius {int(o64[4][j])}")
