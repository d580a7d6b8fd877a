"""Per-tile list lengths of the bench frame (how uneven is the compositing work across workgroups?)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rasterizer_oracle as O
from rodygs_amd.synthetic import synthetic_scene
from rodygs_amd import _lib
from rodygs_amd.rasterizer import GaussianRasterizationSettings, _c_settings
L = _lib.lib()
P, W, H, K = 1000000, 1920, 1080, 16
sc = synthetic_scene(P, W, H, 3, seed=777)
dev = torch.device("cuda")
t = {k: sc[k].to(dev).contiguous() for k in ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix")}
rs = GaussianRasterizationSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3, device=dev), 1.0, t["projmatrix"], 3, False, False, True, True)
cs = _c_settings(rs, P, K)
u8 = dict(dtype=torch.uint8, device=dev)
n_tiles, cap = ((W + 15) // 16) * ((H + 15) // 16), 6 * P
geom = torch.empty(L.rdg_geom_bytes(P), **u8); binning = torch.empty(L.rdg_binning_bytes(cap, n_tiles), **u8)
image = torch.empty(L.rdg_image_bytes(H, W), **u8)
radii = torch.empty(P, dtype=torch.int32, device=dev); nren = torch.zeros(1, dtype=torch.int32, device=dev)
st = _lib.stream_ptr()
_lib.check(L.rdg_preprocess_forward(C.byref(cs), t["means3D"].data_ptr(), t["shs"].data_ptr(), None, t["opacities"].data_ptr(),
                                    t["scales"].data_ptr(), t["rotations"].data_ptr(), None, t["viewmatrix"].data_ptr(),
                                    t["projmatrix"].data_ptr(), geom.data_ptr(), radii.data_ptr(), nren.data_ptr(), st), "pre")
ranges = torch.zeros(n_tiles, 2, dtype=torch.int32, device=dev)
_lib.check(L.rdg_bin_forward(C.byref(cs), geom.data_ptr(), radii.data_ptr(), binning.data_ptr(), cap, image.data_ptr(),
                             nren.data_ptr(), None, None, None, None, ranges.data_ptr(), st), "bin")
n = (ranges[:, 1] - ranges[:, 0]).float().cpu()
q = torch.quantile(n, torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0]))
print(json.dumps({"D": int(nren.item()), "tiles": n_tiles, "mean": float(n.mean()), "quantiles_0_10_50_90_99_100": q.tolist()}))
