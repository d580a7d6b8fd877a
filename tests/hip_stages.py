"""Helpers that drive librodygs_hip.so stage by stage through the C-ABI (used by the -m gpu parity tests)."""
import ctypes as C

import numpy as np
import torch

from rodygs_amd import _lib
from rodygs_amd.rasterizer import GaussianRasterizationSettings, _c_settings


def make_settings(sc, sh_degree, bg=None, dev="cuda", cov_grad=True, sh_grad=True, scale_modifier=1.0):
    bg = torch.zeros(3) if bg is None else bg
    return GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=bg.to(dev),
        scale_modifier=scale_modifier, projmatrix=sc["projmatrix"].to(dev), sh_degree=sh_degree, prefiltered=False,
        debug=False, enable_cov_grad=cov_grad, enable_sh_grad=sh_grad)


def run_stages(sc, sh_degree, dev="cuda", capacity=None, bin_mode=None, cull=None):
    """preprocess -> export -> bin (with copies) on the GPU; returns numpy views of every intermediate.
    bin_mode: 0 = bucket binning, 1 = LSD radix sort (RdgRasterSettings.bin_mode); None = the default of the settings.
    cull: RdgRasterSettings.cull -- False = the reference's tile rectangles (the north_star's key stream), True = the tight
    ones; None = the product's default (rasterizer.CULL)."""
    L = _lib.lib()
    rs = make_settings(sc, sh_degree, dev=dev)
    m3 = sc["means3D"].to(dev).contiguous()
    P = m3.shape[0]
    shs = sc["shs"].to(dev).contiguous()
    cs = _c_settings(rs, P, shs.shape[1])
    if bin_mode is not None:
        cs.bin_mode = int(bin_mode)
    if cull is not None:
        cs.cull = int(bool(cull))
    H, W = sc["H"], sc["W"]
    n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    u8 = dict(dtype=torch.uint8, device=dev)
    geom = torch.empty(L.rdg_geom_bytes(P), **u8)
    image = torch.empty(L.rdg_image_bytes(H, W), **u8)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    nren = torch.zeros(1, dtype=torch.int32, device=dev)
    op = sc["opacities"].to(dev).contiguous()
    scl = sc["scales"].to(dev).contiguous()
    rot = sc["rotations"].to(dev).contiguous()
    vm = sc["viewmatrix"].to(dev).contiguous()
    pm = sc["projmatrix"].to(dev).contiguous()
    st = _lib.stream_ptr()
    _lib.check(L.rdg_preprocess_forward(C.byref(cs), m3.data_ptr(), shs.data_ptr(), None, op.data_ptr(),
                                        scl.data_ptr(), rot.data_ptr(), None, vm.data_ptr(), pm.data_ptr(),
                                        geom.data_ptr(), radii.data_ptr(), nren.data_ptr(), st), "preprocess")
    D = int(nren.item())
    f = dict(dtype=torch.float32, device=dev)
    depth = torch.empty(P, **f); xy = torch.empty(P, 2, **f); co = torch.empty(P, 4, **f)
    rgb = torch.empty(P, 3, **f); nrm = torch.empty(P, 3, **f)
    tt = torch.empty(P, dtype=torch.int32, device=dev)
    _lib.check(L.rdg_geom_export(P, geom.data_ptr(), depth.data_ptr(), xy.data_ptr(), co.data_ptr(), rgb.data_ptr(),
                                 nrm.data_ptr(), tt.data_ptr(), st), "export")
    cap = capacity if capacity is not None else D + 17
    binning = torch.empty(L.rdg_binning_bytes(cap, n_tiles), **u8)
    ku = torch.zeros(cap, dtype=torch.int64, device=dev); vu = torch.zeros(cap, dtype=torch.int32, device=dev)
    ks = torch.zeros(cap, dtype=torch.int64, device=dev); vs = torch.zeros(cap, dtype=torch.int32, device=dev)
    rng = torch.zeros(n_tiles, 2, dtype=torch.int32, device=dev)
    _lib.check(L.rdg_bin_forward(C.byref(cs), geom.data_ptr(), radii.data_ptr(), binning.data_ptr(), cap,
                                 image.data_ptr(), nren.data_ptr(), ku.data_ptr(), vu.data_ptr(), ks.data_ptr(),
                                 vs.data_ptr(), rng.data_ptr(), st), "bin")
    torch.cuda.synchronize()
    n = min(D, cap)
    return dict(D=D, depth=depth.cpu().numpy(), xy=xy.cpu().numpy(), conic_opacity=co.cpu().numpy(),
                rgb=rgb.cpu().numpy(), normal=nrm.cpu().numpy(), tiles_touched=tt.cpu().numpy().astype(np.uint32),
                radii=radii.cpu().numpy(), keys_unsorted=ku.cpu().numpy().view(np.uint64)[:n],
                vals_unsorted=vu.cpu().numpy().view(np.uint32)[:n], keys_sorted=ks.cpu().numpy().view(np.uint64)[:n],
                vals_sorted=vs.cpu().numpy().view(np.uint32)[:n], ranges=rng.cpu().numpy().view(np.uint32))
