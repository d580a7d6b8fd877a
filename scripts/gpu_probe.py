"""Diagnostic run on the GPU box: prints parity statistics of every stage against the oracle + rough timings.
Not a test (tests/ holds the asserting versions); used while developing kernels."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import rasterizer_oracle as O  # noqa: E402
from rodygs_amd.synthetic import synthetic_scene
from oracle import deform_oracle as DO  # noqa: E402
from oracle import knn_oracle as KO  # noqa: E402
import hip_stages as HS  # noqa: E402
from rodygs_amd import GaussianRasterizer, _lib, distCUDA2, gaussian_deformation  # noqa: E402

dev = "cuda"


def stats(name, a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    d = (a - b).abs()
    sc = b.abs().max().item() + 1e-30
    print(f"  {name:14s} max_abs={d.max().item():.3e} rel_to_max={d.max().item() / sc:.3e} "
          f"frac>1e-4*max={(d > 1e-4 * sc).double().mean().item():.2e} nan={int(torch.isnan(a).sum())}")


def probe_stages(P, W, H, deg, seed=1):
    print(f"== stages P={P} {W}x{H} deg={deg}")
    sc = synthetic_scene(P, W, H, 3, seed=seed)
    hs = HS.run_stages(sc, deg)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], deg)
    with torch.no_grad():
        g = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                         scales=sc["scales"], rotations=sc["rotations"])
    b = O.bin_and_sort(g)
    vis = g["valid"].numpy()
    print("  D hip/oracle:", hs["D"], b["num_rendered"], " visible:", int(vis.sum()))
    print("  radii equal:", np.array_equal(hs["radii"], g["radii"].numpy()),
          " tiles_touched equal:", np.array_equal(hs["tiles_touched"], g["tiles_touched"].numpy().astype(np.uint32)))
    dbits_h = hs["depth"].view(np.uint32)[vis]
    dbits_o = g["depth"].numpy().view(np.uint32)[vis]
    print("  depth bits equal:", np.array_equal(dbits_h, dbits_o),
          " xy bits equal:", np.array_equal(hs["xy"].view(np.uint32)[vis],
                                            torch.stack([g["px"], g["py"]], 1).numpy().view(np.uint32)[vis]))
    co = torch.cat([g["conic"], g["opacity"].unsqueeze(1)], 1).numpy()
    print("  conic bits equal:", np.array_equal(hs["conic_opacity"].view(np.uint32)[vis], co.view(np.uint32)[vis]))
    ch, cor = hs["conic_opacity"].view(np.int32)[vis].astype(np.int64), co.view(np.int32)[vis].astype(np.int64)
    for c in range(4):
        dd = np.abs(ch[:, c] - cor[:, c])
        print(f"   conic col {c}: mismatches {int((dd > 0).sum())} max ulp {int(dd.max())}")
    stats("rgb", hs["rgb"][vis], g["rgb"].numpy()[vis])
    stats("normal", hs["normal"][vis], g["normal"].numpy()[vis])
    if hs["D"] == b["num_rendered"]:
        print("  keys_unsorted equal:", np.array_equal(hs["keys_unsorted"], b["keys_unsorted"]),
              " vals_unsorted equal:", np.array_equal(hs["vals_unsorted"], b["vals_unsorted"]))
        print("  keys_sorted equal:", np.array_equal(hs["keys_sorted"], b["keys_sorted"]),
              " vals_sorted equal:", np.array_equal(hs["vals_sorted"], b["vals_sorted"]))
        print("  ranges equal:", np.array_equal(hs["ranges"], b["ranges"]))
        if not np.array_equal(hs["keys_sorted"], b["keys_sorted"]):
            bad = np.nonzero(hs["keys_sorted"] != b["keys_sorted"])[0]
            print("   first bad sorted idx", bad[:5], "of", len(bad))
            ks = hs["keys_sorted"]
            print("   hip sorted monotone:", bool(np.all(ks[1:] >= ks[:-1])))


def probe_full(P, W, H, deg, seed=2, bg=(0.1, 0.2, 0.3)):
    print(f"== full fwd/bwd P={P} {W}x{H} deg={deg}")
    sc = synthetic_scene(P, W, H, 3, seed=seed)
    bgt = torch.tensor(bg)
    names = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
    ins = {k: sc[k].clone().to(dev).requires_grad_(True) for k in names}
    rs = HS.make_settings(sc, deg, bg=bgt)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                 scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
    color, depth, normal, alpha, radii, _ = out
    gen = torch.Generator().manual_seed(3)
    wc = torch.rand(3, H, W, generator=gen)
    wd = torch.rand(1, H, W, generator=gen)
    wa = torch.rand(1, H, W, generator=gen)
    loss = (color * wc.to(dev)).sum() + (depth * wd.to(dev)).sum() * 0.1 + (alpha * wa.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    oi = {k: sc[k].clone().requires_grad_(True) for k in names}
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], bgt, 1.0, sc["projmatrix"], deg)
    om2 = torch.zeros(P, 3, requires_grad=True)
    t0 = time.time()
    oc, od, on, oa, orad, aux = O.rasterize(oi["means3D"], om2, oi["opacities"], oi["viewmatrix"], st, shs=oi["shs"],
                                            scales=oi["scales"], rotations=oi["rotations"])
    ol = (oc * wc).sum() + (od * wd).sum() * 0.1 + (oa * wa).sum()
    ol.backward()
    print(f"  oracle fwd+bwd {time.time() - t0:.1f}s  loss hip/oracle {loss.item():.6f} {ol.item():.6f}")
    stats("color", color, oc); stats("depth", depth, od); stats("alpha", alpha, oa); stats("normal", normal, on)
    print("  radii equal:", torch.equal(radii.cpu(), orad))
    for k in names:
        stats("d_" + k, ins[k].grad, oi[k].grad)
    stats("d_means2D", m2.grad, om2.grad)
    print("  d_view hip:\n", ins["viewmatrix"].grad.cpu().numpy(), "\n  d_view oracle:\n", oi["viewmatrix"].grad.numpy())


def probe_misc():
    print("== sort_pairs")
    import ctypes as C
    L = _lib.lib()
    for n, bits in ((1000, 40), (100000, 45), (3000000, 47), (77, 64)):
        g = torch.Generator().manual_seed(n)
        keys = torch.randint(0, 2 ** 62, (n,), generator=g, dtype=torch.int64) & ((1 << bits) - 1)
        keys[::7] = keys[0]  # duplicates -> stability matters
        vals = torch.arange(n, dtype=torch.int32)
        kd, vd = keys.to(dev), vals.to(dev)
        nd = torch.tensor([n], dtype=torch.int32, device=dev)
        tmp = torch.empty(L.rdg_sort_tmp_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(L.rdg_sort_pairs(kd.data_ptr(), vd.data_ptr(), n, nd.data_ptr(), bits, tmp.data_ptr(),
                                    _lib.stream_ptr()), "sort")
        torch.cuda.synchronize()
        order = np.argsort(keys.numpy().view(np.uint64), kind="stable")
        print(f"  n={n} bits={bits} keys ok:", np.array_equal(kd.cpu().numpy(), keys.numpy()[order]),
              " vals ok:", np.array_equal(vd.cpu().numpy(), vals.numpy()[order]))
    print("== deform")
    g = torch.Generator().manual_seed(0)
    P, B, Tu = 5000, 16, 37
    coeff = (0.1 * torch.randn(P, 1, B, generator=g)).requires_grad_(True)
    ind = torch.randint(0, Tu, (P,), generator=g)
    bt = torch.randn(B, 7, generator=g).requires_grad_(True)
    tb = torch.randn(Tu, B, 7, generator=g).requires_grad_(True)
    wx, wr = torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)
    ox, orr = DO.gaussian_deformation(coeff, ind, bt, tb, 1.7)
    ((ox * wx).sum() + (orr * wr).sum()).backward()
    c2 = coeff.detach().clone().to(dev).requires_grad_(True)
    bt2 = bt.detach().clone().to(dev).requires_grad_(True)
    tb2 = tb.detach().clone().to(dev).requires_grad_(True)
    hx, hr = gaussian_deformation(c2, ind.to(dev), bt2, tb2, 1.7)
    ((hx * wx.to(dev)).sum() + (hr * wr.to(dev)).sum()).backward()
    stats("dxyz", hx, ox); stats("drot", hr, orr); stats("d_coeff", c2.grad, coeff.grad)
    stats("d_basis_t", bt2.grad, bt.grad); stats("d_table", tb2.grad, tb.grad)
    print("== knn")
    pts = torch.rand(20000, 3, generator=g) * torch.tensor([3.0, 1.0, 0.2])
    stats("dist2", distCUDA2(pts.to(dev)), KO.dist2_knn3(pts))
    print("== adam")
    n = 100003
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-2, eps=1e-15)
    pd = p.clone().to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        pt.grad = gr * step
        opt.step()
        gd = (gr * step).to(dev)
        _lib.check(L.rdg_adam_step(n, pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-2, 0.9, 0.999, 1e-15,
                                   step, _lib.stream_ptr()), "adam")
    stats("adam", pd, pt.detach())


def probe_speed(P=1000000, W=1920, H=1080, deg=3, iters=5):
    print(f"== speed P={P} {W}x{H}")
    sc = synthetic_scene(P, W, H, 3, seed=777)
    names = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
    ins = {k: sc[k].clone().to(dev).requires_grad_(True) for k in names}
    rs = HS.make_settings(sc, deg)
    rast = GaussianRasterizer(rs)
    _lib.timing_enable(True)
    for it in range(iters + 2):
        if it == 2:
            torch.cuda.synchronize(); _lib.timing_reset(); t0 = time.time()
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        color, depth, normal, alpha, radii, _ = rast(means3D=ins["means3D"], means2D=m2, shs=ins["shs"],
                                                     opacities=ins["opacities"], scales=ins["scales"],
                                                     rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
        (color.mean() + depth.mean()).backward()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    print(f"  fwd+bwd wall {dt * 1e3:.2f} ms/iter; visible={(radii > 0).sum().item()} ")
    for k, (ms, n) in _lib.stage_times().items():
        if n:
            print(f"   stage {k:15s} {ms / n:8.3f} ms x{n}")
    _lib.timing_enable(False)


if __name__ == "__main__":
    what = sys.argv[1:] or ["stages", "full", "misc", "speed"]
    print(torch.cuda.get_device_name(0))
    if "stages" in what:
        probe_stages(1000, 256, 256, 0)
        probe_stages(20000, 640, 360, 3)
        probe_stages(100000, 1920, 1080, 3)
    if "full" in what:
        probe_full(1000, 256, 256, 0)
        probe_full(5000, 320, 200, 3)
    if "misc" in what:
        probe_misc()
    if "speed" in what:
        probe_speed()
