"""-m gpu tests added in round 6: the model variant of render(), the culled key stream at full size, the torch-free C caller
of the C-ABI, the hard-regime full-frame parity cases."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import rasterizer_oracle as O
from oracle.parity import flipped_pixels

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tp():
    import test_gpu_parity as TP
    return TP


@pytest.mark.parametrize("isotropic", [False, True])
def test_model_render_variant_translation_rotation_keys_and_values(isotropic):
    """``render_model`` = StaticRoDyGS.render (/root/reference/src/model/rodygs_static.py:184-296): the two extra kwargs shift
    the means / the quaternions (the latter not for an isotropic model, :247-249), the result carries the ten keys of the
    reference, values and gradients (through ``translation`` and ``rotation`` too) equal the oracle's on the shifted cloud."""
    TP = _tp()
    from rodygs_amd import render_model
    from rodygs_amd.rasterizer import RasterState, last_compositing_state
    P, W, H = 2000, 160, 120
    sc = O.synthetic_scene(P, W, H, 3, seed=21)
    sc["viewmatrix"] = TP.orbit_view(3.0, -2.0, (0.1, 0.05, 0.2))
    bg = torch.tensor([0.2, 0.4, 0.1])

    class Cam:
        FoVx, FoVy = sc["fovx"], sc["fovy"]
        image_height, image_width = H, W
        projection_matrix = sc["projmatrix"].t().contiguous().to(DEV)
        world_view_transform = sc["viewmatrix"].t().contiguous().to(DEV)

    class Model:
        active_sh_degree = 2
        get_xyz = sc["means3D"].to(DEV).requires_grad_(True)
        get_opacity = sc["opacities"].to(DEV)
        get_scaling = sc["scales"].to(DEV)
        get_rotation = sc["rotations"].to(DEV)
        get_features = sc["shs"].to(DEV)
    Model.isotropic = isotropic
    t_cpu, r_cpu = torch.tensor([0.05, -0.02, 0.1]), torch.tensor([0.03, -0.04, 0.02, 0.05])
    t = t_cpu.to(DEV).requires_grad_(True)
    r = r_cpu.to(DEV).requires_grad_(True)
    st = RasterState()
    pkg = render_model(Model, Cam, bg.to(DEV), translation=t, rotation=r, raster_state=st, some_unknown_kwarg=1)
    fT, nc = last_compositing_state(st)
    assert set(pkg) == {"rendered_image", "rendered_depth", "rendered_normal", "rendered_alpha", "viewspace_points",
                        "visibility_filter", "radii", "extra", "translation", "rotation"}
    assert pkg["translation"] is t and pkg["rotation"] is r
    g = torch.Generator().manual_seed(5)
    wc, wd = torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g)
    ((pkg["rendered_image"] * wc.to(DEV)).sum() + 0.1 * (pkg["rendered_depth"] * wd.to(DEV)).sum()).backward()
    # the oracle on the cloud the reference would hand its rasterizer
    ot, orr = t_cpu.clone().requires_grad_(True), r_cpu.clone().requires_grad_(True)
    oxyz = sc["means3D"].clone().requires_grad_(True)
    om2 = torch.zeros(P, 3, requires_grad=True)
    ost = O.OracleSettings(H, W, math.tan(sc["fovx"] * 0.5), math.tan(sc["fovy"] * 0.5), bg, 1.0, sc["projmatrix"], 2,
                           enable_cov_grad=False, enable_sh_grad=False)
    rot = sc["rotations"] if isotropic else sc["rotations"] + orr
    oc, od, on, oa, orad, oaux = O.rasterize(oxyz + ot, om2, sc["opacities"], sc["viewmatrix"], ost, shs=sc["shs"],
                                             scales=sc["scales"], rotations=rot)
    ((oc * wc).sum() + 0.1 * (od * wd).sum()).backward()
    flips = flipped_pixels(fT, nc, oaux["final_T"], oaux["n_contrib"])
    for key, ref in (("rendered_image", oc), ("rendered_depth", od), ("rendered_normal", on), ("rendered_alpha", oa)):
        TP.rel_ok(pkg[key], ref, outliers=TP.OUTLIER_FRAC, what=key, flips=flips)
    assert torch.equal(pkg["radii"].cpu(), orad) and torch.equal(pkg["visibility_filter"].cpu(), orad > 0)
    TP.rel_ok(pkg["viewspace_points"].grad, om2.grad, outliers=TP.OUTLIER_FRAC, what="viewspace_points.grad", flips=flips)
    TP.rel_ok(Model.get_xyz.grad, oxyz.grad, outliers=TP.OUTLIER_FRAC, what="d_xyz", flips=flips)
    TP.rel_ok(t.grad, ot.grad, what="d_translation", tol=2e-4)
    if isotropic:
        assert r.grad is None
    else:
        TP.rel_ok(r.grad, orr.grad, what="d_rotation", tol=2e-4)
    # the defaults are the NUMBER 0.0, handed back as given
    with torch.no_grad():
        pkg0 = render_model(Model, Cam, bg.to(DEV), raster_state=st)
    assert pkg0["translation"] == 0.0 and pkg0["rotation"] == 0.0


def test_torch_free_c_caller_of_the_abi_reproduces_the_committed_fixture(tmp_path):
    """"no torch headers, the .so never allocates".  tests/abi_caller.c -- plain C compiled with gcc,
    hipMalloc'd buffers, the library through dlopen, no Python and no torch in the process -- runs BASELINE configs[0]'s
    fixture scene through rdg_preprocess_forward + rdg_bin_forward, rdg_rasterize_forward and rdg_rasterize_backward as a
    CHILD process; this test only writes its raw input files and compares its raw output files with
    tests/golden/rasterizer_golden_c1.npz: radii, sorted keys / Gaussian indices / ranges bit-exact (under the reference's
    rectangles AND the tight ones), images and every gradient <= 1e-4 per column."""
    import importlib.util
    TP = _tp()
    from rodygs_amd import _lib
    exe = os.path.join(ROOT, "tests", "abi_caller")
    assert os.path.exists(exe), "tests/abi_caller is not built (make -C rodygs_amd/csrc abi_caller)"
    spec = importlib.util.spec_from_file_location("mrg", os.path.join(ROOT, "tests", "golden", "make_rasterizer_golden.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    g = np.load(os.path.join(ROOT, "tests", "golden", "rasterizer_golden_c1.npz"))
    P, Msh = g["in_shs"].shape[0], g["in_shs"].shape[1]
    H, W = int(g["in_H"]), int(g["in_W"])
    wc, wd, wa = M.fixture_weights(H, W)          # dL/dcolour, dL/ddepth (x 0.1), dL/dalpha of fixture_loss
    for cull in (0, 1):
        d = tmp_path / f"cull{cull}"
        d.mkdir()
        for k in ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "bg"):
            np.ascontiguousarray(g["in_" + k], dtype=np.float32).tofile(d / f"in_{k}.bin")
        wc.numpy().astype(np.float32).tofile(d / "in_g_color.bin")
        (0.1 * wd).numpy().astype(np.float32).tofile(d / "in_g_depth.bin")
        wa.numpy().astype(np.float32).tofile(d / "in_g_alpha.bin")
        (d / "meta.txt").write_text(f"{P} {Msh} {int(g['in_sh_degree'])} {H} {W} {float(g['in_tanfovx']):.9g} "
                                    f"{float(g['in_tanfovy']):.9g} {cull}\n")
        env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
        r = subprocess.run([exe, _lib.LIB_PATH, str(d)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        pre = "cull_" if cull else ""
        D = int(np.fromfile(d / "out_nren.bin", dtype=np.int32)[0])
        assert D == int(g[pre + "num_rendered"])
        assert np.array_equal(np.fromfile(d / "out_radii_stage.bin", dtype=np.int32), g["radii"])
        assert np.array_equal(np.fromfile(d / "out_radii.bin", dtype=np.int32), g["radii"])
        assert np.array_equal(np.fromfile(d / "out_keys_sorted.bin", dtype=np.uint64), g[pre + "keys_sorted"])
        assert np.array_equal(np.fromfile(d / "out_vals_sorted.bin", dtype=np.uint32), g[pre + "vals_sorted"])
        assert np.array_equal(np.fromfile(d / "out_ranges.bin", dtype=np.uint32).reshape(-1, 2), g[pre + "ranges"])
        fT = torch.from_numpy(np.fromfile(d / "out_final_T.bin", dtype=np.float32).reshape(H, W))
        nc = torch.from_numpy(np.fromfile(d / "out_n_contrib.bin", dtype=np.uint32).astype(np.int64).reshape(H, W))
        flips = flipped_pixels(fT, nc, torch.from_numpy(g["final_T"]), torch.from_numpy(g[pre + "n_contrib"].astype(np.int64)))
        for name, shape in (("color", (3, H, W)), ("depth", (1, H, W)), ("normal", (3, H, W)), ("alpha", (1, H, W))):
            got = np.fromfile(d / f"out_{name}.bin", dtype=np.float32).reshape(shape)
            TP.rel_ok(got, g[name], outliers=TP.OUTLIER_FRAC, what=f"C caller {name} (cull {cull})", flips=flips)
        for name in ("means3D", "means2D", "shs", "opacities", "scales", "rotations", "viewmatrix"):
            want = g["grad_" + name]
            got = np.fromfile(d / f"out_d_{name}.bin", dtype=np.float32).reshape(want.shape)
            TP.rel_ok(got, want, outliers=TP.OUTLIER_FRAC, what=f"C caller d_{name} (cull {cull})", flips=flips)
