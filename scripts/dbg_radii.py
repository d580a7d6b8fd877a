"""One sweep case whose radii differ between the HIP per-Gaussian forward and the oracle: which Gaussians, and what their
conics / depths / pixel centres look like on both sides (bit patterns)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_stages as HS
from oracle import rasterizer_oracle as O
from sweep_cases import sweep_case
seed0, c = int(sys.argv[1]), int(sys.argv[2])
sc, deg, bg, kw = sweep_case(seed0, c)
P = sc["means3D"].shape[0]
rs = HS.make_settings(sc, deg, scale_modifier=kw["scale_modifier"])
import ctypes as C
from rodygs_amd import _lib
from rodygs_amd.rasterizer import _c_settings
L = _lib.lib()
dev = "cuda"
m3 = sc["means3D"].to(dev).contiguous(); shs = sc["shs"].to(dev).contiguous()
cs = _c_settings(rs, P, shs.shape[1])
geom = torch.empty(L.rdg_geom_bytes(P), dtype=torch.uint8, device=dev)
radii = torch.empty(P, dtype=torch.int32, device=dev); nren = torch.zeros(1, dtype=torch.int32, device=dev)
op = sc["opacities"].to(dev).contiguous(); scl = sc["scales"].to(dev).contiguous(); rot = sc["rotations"].to(dev).contiguous()
vm = sc["viewmatrix"].to(dev).contiguous(); pm = sc["projmatrix"].to(dev).contiguous()
st = _lib.stream_ptr()
_lib.check(L.rdg_preprocess_forward(C.byref(cs), m3.data_ptr(), shs.data_ptr(), None, op.data_ptr(), scl.data_ptr(),
                                    rot.data_ptr(), None, vm.data_ptr(), pm.data_ptr(), geom.data_ptr(), radii.data_ptr(),
                                    nren.data_ptr(), st), "preprocess")
f = dict(dtype=torch.float32, device=dev)
depth = torch.empty(P, **f); xy = torch.empty(P, 2, **f); co = torch.empty(P, 4, **f); rgb = torch.empty(P, 3, **f)
nrm = torch.empty(P, 3, **f); tt = torch.empty(P, dtype=torch.int32, device=dev)
_lib.check(L.rdg_geom_export(P, geom.data_ptr(), depth.data_ptr(), xy.data_ptr(), co.data_ptr(), rgb.data_ptr(), nrm.data_ptr(),
                             tt.data_ptr(), st), "export")
ost = O.OracleSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], torch.zeros(3), kw["scale_modifier"], sc["projmatrix"], deg)
with torch.no_grad():
    g = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], ost, shs=sc["shs"], scales=sc["scales"],
                     rotations=sc["rotations"])
print("oracle keys", sorted(g.keys()))
rh, ro = radii.cpu(), g["radii"]
bad = torch.nonzero(rh != ro).flatten().tolist()
print("radii differ on", len(bad), "of", P, bad[:10])
coh = co.cpu()
for i in bad[:5]:
    print(i, "radius HIP", int(rh[i]), "oracle", int(ro[i]), "scales", sc["scales"][i].tolist())
    print("   conic HIP   ", [float(v).hex() for v in coh[i, :3]])
    if "conic" in g: print("   conic oracle", [float(v).hex() for v in g["conic"][i]])
    print("   depth HIP", float(depth[i]).hex(), "oracle", float(g["depths"][i]).hex() if "depths" in g else None)
    for k in ("cov2d", "cov2D", "lambda1", "mid", "det"):
        if k in g: print("   oracle", k, [float(v).hex() for v in torch.as_tensor(g[k][i]).flatten()])
