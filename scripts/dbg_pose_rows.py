"""Where does the pose-gradient error of a deep-list sweep case come from?  Takes the HIP backward apart at the 64-byte
gradient rows between its two halves (rdg_composite_backward -> rows -> rdg_preprocess_backward) against the oracle run
in float64:
  (1) the rows themselves, column by column, against the float64 oracle's dL/d(conic, opacity, rgb, pixel centre);
  (2) dL/dviewmatrix of the per-Gaussian backward fed the HIP rows;
  (3) dL/dviewmatrix of the SAME per-Gaussian backward fed rows rebuilt from the float64 oracle (rounded to float32),
      all of them or one column group at a time.
usage: python scripts/dbg_pose_rows.py <seed0> <case>"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_stages as HS  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import rasterizer_oracle as O  # noqa: E402
from rodygs_amd import _lib  # noqa: E402
from rodygs_amd.rasterizer import _c_settings  # noqa: E402
from sweep_cases import sweep_case, sweep_case_aniso  # noqa: E402

seed0, c = int(sys.argv[1]), int(sys.argv[2])
sc, deg, bg, kw = (sweep_case_aniso if os.environ.get("RDG_SWEEP_PROFILE") == "aniso" else sweep_case)(seed0, c)
P, H, W = sc["means3D"].shape[0], sc["H"], sc["W"]
print(P, W, H, deg, kw)
dev = "cuda"
gen = torch.Generator().manual_seed(kw["seed"])
wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
wn = torch.randn(3, H, W, generator=gen) * kw["normal_loss"]

# ---- float64 oracle, with the per-Gaussian intermediates kept --------------------------------------------------------
d = {k: sc[k].clone().double().requires_grad_(True) for k in T.NAMES}
st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor(bg).double(), kw["scale_modifier"],
                      sc["projmatrix"].double(), deg, enable_cov_grad=kw["cov_grad"], enable_sh_grad=kw["sh_grad"])
o = O.rasterize(d["means3D"], torch.zeros(P, 3, dtype=torch.float64, requires_grad=True), d["opacities"],
                d["viewmatrix"], st, shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
geom = o[5]["geom"]
for k in ("conic", "rgb", "px", "py"):
    geom[k].retain_grad()
ls = (o[0] * wc.double()).sum() + (o[3] * wa.double()).sum() + (o[1] * wd.double()).sum() * kw["depth_loss"]
if kw["normal_loss"]:
    ls = ls + (o[2] * wn.double()).sum()
ls.backward()
tv = d["viewmatrix"].grad
scale = float(tv.abs().max())

# ---- HIP, stage by stage through the C-ABI ---------------------------------------------------------------------------
L = _lib.lib()
rs = HS.make_settings(sc, deg, bg=torch.tensor(bg), cov_grad=kw["cov_grad"], sh_grad=kw["sh_grad"],
                      scale_modifier=kw["scale_modifier"])
t = {k: sc[k].to(dev).contiguous() for k in T.NAMES}
cs = _c_settings(rs, P, t["shs"].shape[1])
u8 = dict(dtype=torch.uint8, device=dev)
f32 = dict(dtype=torch.float32, device=dev)
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
geom_ws = torch.empty(L.rdg_geom_bytes(P), **u8)
image_ws = torch.empty(L.rdg_image_bytes(H, W), **u8)
radii = torch.empty(P, dtype=torch.int32, device=dev)
nren = torch.zeros(2, dtype=torch.int32, device=dev)
pm = sc["projmatrix"].to(dev).contiguous()
bgd = torch.tensor(bg, **f32)
outs = [torch.empty(n, H, W, **f32) for n in (3, 1, 3, 1)]
s_ = _lib.stream_ptr()
cap = 64 * P + 65536
while True:
    bin_ws = torch.empty(L.rdg_binning_bytes(cap, n_tiles), **u8)
    _lib.check(L.rdg_rasterize_forward(C.byref(cs), bgd.data_ptr(), t["means3D"].data_ptr(), t["shs"].data_ptr(), None,
                                       t["opacities"].data_ptr(), t["scales"].data_ptr(), t["rotations"].data_ptr(), None,
                                       t["viewmatrix"].data_ptr(), pm.data_ptr(), geom_ws.data_ptr(), bin_ws.data_ptr(), cap,
                                       image_ws.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                       outs[3].data_ptr(), radii.data_ptr(), nren.data_ptr(), s_), "fwd")
    D = int(nren[0])
    if D <= cap:
        break
    cap = D + 4096
print("D", D, "largest list", int(nren[1]))
gws = torch.empty(L.rdg_grad_bytes(P), **u8)
gc, gd, ga = wc.to(dev).contiguous(), (wd * kw["depth_loss"]).to(dev).contiguous(), wa.to(dev).contiguous()
gn = wn.to(dev).contiguous() if kw["normal_loss"] else None
_lib.check(L.rdg_composite_backward(C.byref(cs), bgd.data_ptr(), geom_ws.data_ptr(), bin_ws.data_ptr(), cap,
                                    image_ws.data_ptr(), gc.data_ptr(), gd.data_ptr(), ga.data_ptr(),
                                    gn.data_ptr() if gn is not None else None, gws.data_ptr(), s_),
           "composite_backward")
rows = gws[:P * 64].view(torch.float32).view(P, 16)
rows_hip = rows.clone()


def pre_bwd():
    dm3, dm2 = torch.empty(P, 3, **f32), torch.empty(P, 3, **f32)
    dsh, dop = torch.empty_like(t["shs"]), torch.empty(P, 1, **f32)
    dsc, dro, dvm = torch.empty(P, 3, **f32), torch.empty(P, 4, **f32), torch.empty(4, 4, **f32)
    _lib.check(L.rdg_preprocess_backward(C.byref(cs), t["means3D"].data_ptr(), t["shs"].data_ptr(), None,
                                         t["opacities"].data_ptr(), t["scales"].data_ptr(), t["rotations"].data_ptr(), None,
                                         t["viewmatrix"].data_ptr(), pm.data_ptr(), radii.data_ptr(), geom_ws.data_ptr(),
                                         gws.data_ptr(), dm3.data_ptr(), dm2.data_ptr(), dsh.data_ptr(), None, dop.data_ptr(),
                                         dsc.data_ptr(), dro.data_ptr(), None, dvm.data_ptr(), s_), "preprocess_backward")
    torch.cuda.synchronize()
    pre_bwd.last = (dm3.cpu().double(), dsc.cpu().double(), dro.cpu().double())
    return dvm.cpu().double()


def err(v):
    return float((v - tv).abs().max()) / scale


print(f"pose gradient scale {scale:.4e}")
print(f"(2) per-Gaussian backward on the HIP rows:           max rel {err(pre_bwd()):.3e}")

# rows of the float64 oracle (rdg_common.h RDG_GROW layout; first five divided by the opacity)
op = d["opacities"].detach().reshape(P)
vis = geom["radii"] > 0
z = torch.zeros(P, dtype=torch.float64)
gcon = geom["conic"].grad if geom["conic"].grad is not None else torch.zeros(P, 3, dtype=torch.float64)
gpx = geom["px"].grad if geom["px"].grad is not None else z
gpy = geom["py"].grad if geom["py"].grad is not None else z
cov2 = geom["cov2D"].detach()
m1x = -(cov2[:, 0] * gpx + cov2[:, 1] * gpy)
m1y = -(cov2[:, 1] * gpx + cov2[:, 2] * gpy)
# the rows hold moments about (w, dy), w = dx + beta dy, beta = conic_b / conic_a (rdg_bwd_walk)
kon = geom["conic"].detach()
beta = torch.where(kon[:, 0] != 0, kon[:, 1] / kon[:, 0], z)
orow = torch.zeros(P, 16, dtype=torch.float64)
inv_o = torch.where(op != 0, 1.0 / op, z)
orow[:, 0], orow[:, 1] = (m1x + beta * m1y) * inv_o, m1y * inv_o
# conic-gradient columns in the (w, dy) moment basis of the rows: -1/2 S_ww = gca + beta gcb + beta^2 gcc, -S_wy = gcb + 2 beta gcc
gw = torch.stack([gcon[:, 0] + beta * gcon[:, 1] + beta * beta * gcon[:, 2], gcon[:, 1] + 2 * beta * gcon[:, 2], gcon[:, 2]], 1)
orow[:, 2:5] = gw * inv_o.unsqueeze(1)
orow[:, 6:9] = geom["rgb"].grad if geom["rgb"].grad is not None else 0.0
orow = torch.where(vis.unsqueeze(1), orow, torch.zeros_like(orow))
# the float32 oracle's own rows: what float32 autograd (every T_i kept from the forward) reaches
d32 = {k: sc[k].clone().requires_grad_(True) for k in T.NAMES}
st32 = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor(bg), kw["scale_modifier"], sc["projmatrix"], deg,
                        enable_cov_grad=kw["cov_grad"], enable_sh_grad=kw["sh_grad"])
o32 = O.rasterize(d32["means3D"], torch.zeros(P, 3, requires_grad=True), d32["opacities"], d32["viewmatrix"], st32,
                  shs=d32["shs"], scales=d32["scales"], rotations=d32["rotations"])
g32 = o32[5]["geom"]
for k in ("conic", "rgb"):
    g32[k].retain_grad()
((o32[0] * wc).sum() + (o32[3] * wa).sum() + (o32[1] * wd).sum() * kw["depth_loss"]
 + ((o32[2] * wn).sum() if kw["normal_loss"] else 0.0)).backward()
r32 = torch.zeros(P, 16, dtype=torch.float64)
g32c = g32["conic"].grad.double()
r32[:, 2:5] = torch.stack([g32c[:, 0] + beta * g32c[:, 1] + beta * beta * g32c[:, 2], g32c[:, 1] + 2 * beta * g32c[:, 2],
                           g32c[:, 2]], 1) * inv_o.unsqueeze(1)
r32[:, 6:9] = g32["rgb"].grad.double()
print("(0) float32-oracle rows vs float64 oracle:")
for name, sl in (("conic", slice(2, 5)), ("rgb", slice(6, 9))):
    a, b = r32[vis][:, sl], orow[vis][:, sl]
    print(f"    {name:8s}", [f"{float(x):.2e}" for x in ((a - b).abs().amax(0) / b.abs().amax(0).clamp_min(1e-300))])
print(f"    float32-oracle pose gradient vs float64: max rel {err(d32['viewmatrix'].grad.double()):.3e}")
hr = rows_hip.cpu().double()
print("(1) HIP rows vs float64 oracle, max |diff| / max |oracle| per column (visible Gaussians):")
for name, sl in (("moments", slice(0, 2)), ("conic", slice(2, 5)), ("rgb", slice(6, 9))):
    a, b = hr[vis][:, sl], orow[vis][:, sl]
    print(f"    {name:8s}", [f"{float(x):.2e}" for x in ((a - b).abs().amax(0) / b.abs().amax(0).clamp_min(1e-300))])
groups = {"moments": [0, 1], "conic": [2, 3, 4], "rgb": [6, 7, 8]}
for label, cols in list(groups.items()) + [("moments+conic+rgb", [0, 1, 2, 3, 4, 6, 7, 8])]:
    rows.copy_(rows_hip)
    rows[:, cols] = orow[:, cols].to(torch.float32).to(dev)
    print(f"(3) oracle-f64 {label:18s} in the rows:        max rel {err(pre_bwd()):.3e}")
rows.copy_(rows_hip)

# ---- per Gaussian: where the per-Gaussian backward itself (fed the float64 rows) is least accurate -----------------------
rows[:, [0, 1, 2, 3, 4, 6, 7, 8]] = orow[:, [0, 1, 2, 3, 4, 6, 7, 8]].to(torch.float32).to(dev)
pre_bwd()
dm3, dsc, dro = pre_bwd.last
rows.copy_(rows_hip)
torch.set_printoptions(precision=4, linewidth=220, sci_mode=True)
for name, hipg, org in (("means3D", dm3, d["means3D"].grad), ("scales", dsc, d["scales"].grad), ("rotations", dro, d["rotations"].grad)):
    colmax = org.abs().amax(0)
    e = ((hipg - org).abs() / colmax).amax(1)
    worst = torch.argsort(e, descending=True)[:6]
    print(f"per-Gaussian backward on float64 rows, d_{name}: worst Gaussians (error / column max)")
    for i in worst.tolist():
        a_, b_, c_ = [float(x) for x in cov2[i]]
        det = a_ * c_ - b_ * b_
        print(f"    #{i}: err {float(e[i]):.2e}  |grad|/colmax {float((org[i].abs() / colmax).max()):.2e}  depth {float(geom['depth'][i]):.3f} "
              f"radius {int(geom['radii'][i])} opacity {float(op[i]):.3f} cov2D ({a_:.3e}, {b_:.3e}, {c_:.3e}) det/(ac) {det / (a_ * c_):.2e} "
              f"tiles {int(geom['tiles_touched'][i])}")

# the rows as files: scripts/dbg_scale_grad.py replays the per-Gaussian backward on them on the CPU
import numpy as np  # noqa: E402
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
pre_bwd()
np.savez(os.path.join(ROOT, "gpurun_out", f"rows_{seed0}_{c}.npz"), rows_hip=rows_hip.cpu().numpy(), rows_f64=orow.numpy(),
         dsc_hip=pre_bwd.last[1].numpy(), dm3_hip=pre_bwd.last[0].numpy(), opac=op.numpy())

# ---- the same split against the float64 oracle AT THE FLOAT32 GEOMETRY (tests/resolution.py class "geom"): rows rebuilt from its
# dL/d(px, py, conic, rgb) and fed to the per-Gaussian backward -- isolates the float32 per-Gaussian backward from the rows --------
import resolution  # noqa: E402
gg, ggeom, g32_ = resolution.float32_geometry_run(sc, deg, bg, kw, return_geom=True)
kon32 = g32_["conic"].double()
beta32 = torch.where(kon32[:, 0] != 0, (kon32[:, 1].float() / kon32[:, 0].float()).double(), z)       # the record's float32 beta
cov32 = g32_["cov2D"].double()
gpx_, gpy_ = ggeom["px"], ggeom["py"]
m1x_ = -(cov32[:, 0] * gpx_ + cov32[:, 1] * gpy_)
m1y_ = -(cov32[:, 1] * gpx_ + cov32[:, 2] * gpy_)
grow = torch.zeros(P, 16, dtype=torch.float64)
grow[:, 0], grow[:, 1] = (m1x_ + beta32 * m1y_) * inv_o, m1y_ * inv_o
gcn = ggeom["conic"]
grow[:, 2:5] = torch.stack([gcn[:, 0] + beta32 * gcn[:, 1] + beta32 * beta32 * gcn[:, 2], gcn[:, 1] + 2 * beta32 * gcn[:, 2],
                            gcn[:, 2]], 1) * inv_o.unsqueeze(1)
grow[:, 5] = ggeom["opacity"]
grow[:, 6:9] = ggeom["rgb"]
grow[:, 9] = ggeom["depth"]
grow = torch.where(vis.unsqueeze(1), grow, torch.zeros_like(grow))
print("(4) HIP rows vs the float32-geometry arbiter's rows, per column:")
a_, b_ = hr[vis][:, :10], grow[vis][:, :10]
print("   ", [f"{float(x):.2e}" for x in ((a_ - b_).abs().amax(0) / b_.abs().amax(0).clamp_min(1e-300))])
for label, use in (("HIP rows", None), ("arbiter rows", grow)):
    rows.copy_(rows_hip)
    if use is not None:
        rows[:, :10] = use[:, :10].to(torch.float32).to(dev)
    pre_bwd()
    dm3_, dsc_, dro_ = pre_bwd.last
    for name, hipg, ref_ in (("means3D", dm3_, gg["means3D"]), ("scales", dsc_, gg["scales"]), ("rotations", dro_, gg["rotations"])):
        colmax = d[name].grad.abs().amax(0)
        print(f"(5) per-Gaussian backward on {label:12s}: d_{name} vs the float32-geometry arbiter, per column",
              [f"{float(x):.2e}" for x in ((hipg - ref_).abs().amax(0) / colmax)])
rows.copy_(rows_hip)
