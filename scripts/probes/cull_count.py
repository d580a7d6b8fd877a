"""How many (tile, Gaussian) instances of the bench frame can never blend?  CPU, numpy, oracle geometry.
A = tight rectangle (axis-aligned box of the alpha >= 1/255 ellipse, intersected with the reference rectangle);
B = exact test ellipse vs the tile's rectangle of pixel centres; Q = any of the four quadrants' rectangles."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from rodygs_amd.synthetic import synthetic_scene
from oracle import rasterizer_oracle as ro

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
variant = sys.argv[2] if len(sys.argv) > 2 else "uniform"
W, H = 1920, 1080
sc = synthetic_scene(P, W, H, variant=variant)
st = ro.OracleSettings(image_height=H, image_width=W, tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=torch.zeros(3),
                       scale_modifier=1.0, projmatrix=sc["projmatrix"], sh_degree=3)
with torch.no_grad():
    g = ro.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                      scales=sc["scales"], rotations=sc["rotations"])
gx, gy = g["grid"]
tiles = g["tiles_touched"].numpy().astype(np.int64)
rx0, ry0, rx1, ry1 = [r.numpy().astype(np.int64) for r in g["rect"]]
D = tiles.sum()
a, b, c = [g["conic"][:, i].numpy().astype(np.float64) for i in range(3)]
o = g["opacity"].numpy().astype(np.float64)
px, py = g["px"].numpy().astype(np.float64), g["py"].numpy().astype(np.float64)
vis = tiles > 0
r2 = 2 * np.log(np.maximum(255 * o, 1e-30))
det = a * c - b * b
hx = np.sqrt(np.maximum(r2, 0) * c / det); hy = np.sqrt(np.maximum(r2, 0) * a / det)
plx, phx, ply, phy = np.ceil(px - hx), np.floor(px + hx), np.ceil(py - hy), np.floor(py + hy)
empty = (plx > phx) | (ply > phy)
tx0 = np.clip(np.floor(plx / 16), 0, gx).astype(np.int64); ty0 = np.clip(np.floor(ply / 16), 0, gy).astype(np.int64)
tx1 = np.clip(np.floor(phx / 16) + 1, 0, gx).astype(np.int64); ty1 = np.clip(np.floor(phy / 16) + 1, 0, gy).astype(np.int64)
ax0, ay0, ax1, ay1 = np.maximum(rx0, tx0), np.maximum(ry0, ty0), np.minimum(rx1, tx1), np.minimum(ry1, ty1)
tA = np.where(vis & (r2 > 0) & ~empty, np.maximum(ax1 - ax0, 0) * np.maximum(ay1 - ay0, 0), 0)
print(f"P {P} variant {variant}: visible {vis.sum()}  D_ref {D}  D_A(tight rect) {tA.sum()} ({1 - tA.sum() / D:.3f} culled)")

def exact(idsel, x0r, y0r, x1r, cnt, n_sub):
    ids = np.repeat(np.arange(P)[idsel], cnt[idsel])
    starts = np.cumsum(cnt[idsel]) - cnt[idsel]
    j = np.arange(len(ids)) - np.repeat(starts, cnt[idsel])
    w = np.maximum((x1r - x0r)[ids], 1)
    ty = y0r[ids] + j // w; tx = x0r[ids] + j % w
    A, B, C, R2 = a[ids], b[ids], c[ids], r2[ids]
    res = np.zeros(len(ids), bool)
    step = 16 // n_sub
    for sy in range(n_sub):
        for sx in range(n_sub):
            u1 = px[ids] - (16 * tx + step * sx); u0 = u1 - (step - 1)
            v1 = py[ids] - (16 * ty + step * sy); v0 = v1 - (step - 1)
            inside = (u0 <= 0) & (u1 >= 0) & (v0 <= 0) & (v1 >= 0)
            def em(A_, B2, C_, ue, w0, w1):
                t = B2 * ue; vs = np.clip(-0.5 * t / C_, w0, w1)
                return (C_ * vs + t) * vs + A_ * ue * ue
            m = np.minimum(np.minimum(em(A, 2 * B, C, u0, v0, v1), em(A, 2 * B, C, u1, v0, v1)),
                           np.minimum(em(C, 2 * B, A, v0, u0, u1), em(C, 2 * B, A, v1, u0, u1)))
            res |= inside | (m <= R2)
    return res.sum()

selA = tA > 0
print(f"  B on A (exact tile test inside the tight rect): {exact(selA, ax0, ay0, ax1, tA, 1)}  quadrant-any on A: {exact(selA, ax0, ay0, ax1, tA, 2)}")
print(f"  B on ref rect: {exact(vis, rx0, ry0, rx1, tiles, 1)}  quadrant-any on ref: {exact(vis, rx0, ry0, rx1, tiles, 2)}")
