"""One sweep case in detail: the pose gradient of the HIP path and of the f32 oracle against the oracle run in float64."""
import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from oracle import rasterizer_oracle as O
from sweep_cases import sweep_case, sweep_case_aniso
seed0, c = int(sys.argv[1]), int(sys.argv[2])
sc, deg, bg, kw = (sweep_case_aniso if os.environ.get("RDG_SWEEP_PROFILE") == "aniso" else sweep_case)(seed0, c)
P, W, H, deg_max = sc["means3D"].shape[0], sc["W"], sc["H"], int(round(sc["shs"].shape[1] ** 0.5)) - 1
for a in sys.argv[3:]:                                    # e.g. cov_grad=0 sh_grad=0 normal_loss=0 depth_loss=0
    k_, v_ = a.split("="); kw[k_] = type(kw[k_])(float(v_))
print(P, W, H, deg, deg_max, kw)
res = T.run_pair(sc, deg, bg, **kw)
hv, ov = res[0]["viewmatrix"].grad.cpu().double(), res[3]["viewmatrix"].grad.double()
# float64 oracle with the same loss weights
gen = torch.Generator().manual_seed(kw["seed"])
wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
wn = torch.randn(3, H, W, generator=gen) * kw["normal_loss"]
d = {k: sc[k].clone().double().requires_grad_(True) for k in T.NAMES}
st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor(bg).double(), kw["scale_modifier"], sc["projmatrix"].double(), deg,
                      enable_cov_grad=kw["cov_grad"], enable_sh_grad=kw["sh_grad"])
o = O.rasterize(d["means3D"], torch.zeros(P, 3, dtype=torch.float64, requires_grad=True), d["opacities"], d["viewmatrix"], st,
                shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
ls = (o[0] * wc.double()).sum() + (o[3] * wa.double()).sum()
if kw["depth_loss"]: ls = ls + (o[1] * wd.double()).sum() * kw["depth_loss"]
if kw["normal_loss"]: ls = ls + (o[2] * wn.double()).sum()
ls.backward()
tv = d["viewmatrix"].grad
sc_ = tv.abs().max()
print("scale", float(sc_))
print("HIP    vs f64 oracle: max rel", float((hv - tv).abs().max() / sc_))
print("f32 or vs f64 oracle: max rel", float((ov - tv).abs().max() / sc_))
print("HIP    vs f32 oracle: max rel", float((hv - ov).abs().max() / sc_))
torch.set_printoptions(precision=5, linewidth=200, sci_mode=False)
print("f64 oracle:\n", tv)
print("HIP - f64:\n", hv - tv)
print("f32 oracle - f64:\n", ov - tv)
for k in ("means3D", "scales", "rotations", "opacities"):
    a, b = res[0][k].grad.cpu().double(), d[k].grad
    print(k, "HIP vs f64 per-column max rel:", [float(x) for x in ((a - b).abs().amax(0) / b.abs().amax(0).clamp_min(1e-30))])
    o32 = res[3][k].grad.double()
    print(k, "f32 oracle vs f64 per-column :", [float(x) for x in ((o32 - b).abs().amax(0) / b.abs().amax(0).clamp_min(1e-30))])
