"""-m gpu parity tests: the HIP path (through the C-ABI / the drop-in Python surface) against the CPU oracle on
the same seeded inputs.  Bars (BASELINE.json north_star): tile keys, sort order, radii, tile ranges BIT-EXACT;
rendered RGB / depth / alpha and all gradients within 1e-4 relative.

"relative" = |hip - oracle| <= TOL * max|oracle| PER COLUMN (rel_ok / _columns: every component of a per-Gaussian
tensor, every SH band x channel, every image channel has its own scale).  A handful of pixels may sit exactly on a
discontinuity of the algorithm itself (alpha vs 1/255, T vs 1e-4: an exp() ulp flips whether a splat is blended), so
at most OUTLIER_FRAC of a column's entries may be above the bar, and none of them by more than OUTLIER_CAP of the
column's scale.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from oracle import deform_oracle as DO
from oracle import knn_oracle as KO
from oracle import rasterizer_oracle as O
from oracle.parity import column_stats, columns as _columns, flip_mask, flipped_pixels  # noqa: F401

pytestmark = pytest.mark.gpu

TOL = 1e-4
OUTLIER_FRAC = 2e-5
OUTLIER_CAP = 2e-2          # small scenes (check_pair): ONE flipped pixel is a visible share of a 777-Gaussian column
FULL_FRAME_CAP = 5e-3       # whole frames (full_frame_report): what a witnessed flip may move, column scales of 10^5 Gaussians
DEV = "cuda"
NAMES = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")


# entries of a column (of >= 256) that may sit above the bar wherever outliers are allowed at all: one flipped pixel moves the
# gradient rows of the flipped splat and of the few behind it in that pixel.  1 in the test-suite; scripts/parity_sweep.py
# raises it to 4 for its random scenes (3 of 400 needed more than one entry, none more than three)
FLIP_ENTRIES = 1


def rel_ok(a, b, tol=TOL, outliers=0.0, what="", cap=OUTLIER_CAP, flips=None):
    """|a - b| <= tol * max|b| COLUMN BY COLUMN (_columns): a small column (a degree-3 SH band next to the DC band, one
    quaternion component) is held to its own scale, not to the tensor's.  At most `outliers` of a column's entries (and
    always one entry of a column of 256 or more, when outliers are allowed at all) may sit above the bar (a decision flipped on an alpha = 1/255 / T = 1e-4 discontinuity) and none of them above
    cap * scale: a flipped decision changes an entry by one pixel-splat pair's share, never by the entry itself.  A
    column the oracle has exactly zero must be exactly zero.
    flips: the number of pixels with a WITNESSED decision flip (flipped_pixels: contributor count or final transmittance of
    the pixel differs between the two implementations).  When given, the one-entry-per-column allowance is granted only if
    it is > 0 -- the allowance exists for that cause and no other; None = the comparison has no per-pixel state to look at
    (two HIP paths against each other, float-atomic noise)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert not torch.isnan(a).any(), what
    A, B = _columns(a), _columns(b)
    d = (A - B).abs()
    scale = B.abs().amax(dim=1, keepdim=True)
    rel = d / scale.clamp_min(1e-300)
    rel = torch.where((scale == 0) & (d == 0), torch.zeros_like(rel), rel)
    frac = (rel > tol).double().mean(dim=1)
    worst = rel.amax(dim=1)
    # where outliers are allowed at all, ONE entry of a column of >= 256 is: a single pixel on the alpha = 1/255 boundary
    # is 2.6e-4 of a 16x240 image and moves one Gaussian's gradient row of a 777-Gaussian cloud by 2e-4 of the column's
    # scale (scripts/parity_sweep.py: 2 such pixels in 200 random scenes), still bounded by `cap`
    n_col = A.shape[1]
    grant = outliers > 0 and n_col >= 256 and (flips is None or flips > 0)
    allowed = max(outliers, (FLIP_ENTRIES + 0.5) / n_col) if grant else outliers
    if bool((frac > allowed).any()) or bool((worst > (cap if outliers > 0 else tol)).any()):
        order = torch.argsort(worst, descending=True)[:5]
        rows = "; ".join(f"col {int(c)}: max rel {worst[c].item():.3e}, frac over bar {frac[c].item():.2e}, "
                         f"scale {scale[c].item():.3e}" for c in order)
        raise AssertionError(f"{what} [{A.shape[0]} columns, tol {tol:g}, outliers {outliers:g}, cap {cap:g}, "
                             f"witnessed flips {flips}]: {rows}")


def orbit_view(deg_y=8.0, deg_x=-5.0, t=(0.4, -0.3, 0.8)):
    """A non-identity W2C so every entry of the pose gradient is exercised; returns glm-flat [4,4]."""
    ay, ax = math.radians(deg_y), math.radians(deg_x)
    Ry = torch.tensor([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    w2c = torch.eye(4)
    w2c[:3, :3] = Rx @ Ry
    w2c[:3, 3] = torch.tensor(t)
    return w2c.t().contiguous()


def run_pair(sc, deg, bg, cov_grad=True, sh_grad=True, use_colors=False, use_cov3d=False, scale_modifier=1.0, seed=3,
             normal_loss=0.0, depth_loss=0.1):
    """Forward+backward through rodygs_amd (GPU) and through the oracle (CPU) with the same random loss weights.
    normal_loss: weight of a random linear loss on rendered_normal (drawn last: the other weights keep their values)."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    P, H, W = sc["means3D"].shape[0], sc["H"], sc["W"]
    bgt = torch.tensor(bg)
    gen = torch.Generator().manual_seed(seed)
    wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
    extra = {}
    if use_colors:
        extra["colors_precomp"] = torch.rand(P, 3, generator=gen)
    if use_cov3d:
        with torch.no_grad():
            extra["cov3Ds_precomp"] = O.covariance3d(sc["scales"], scale_modifier, sc["rotations"])

    def inputs(dev):
        d = {k: sc[k].clone().to(dev).requires_grad_(True) for k in NAMES}
        for k, v in extra.items():
            d[k] = v.clone().to(dev).requires_grad_(True)
        return d

    hi = inputs(DEV)
    rs = HS.make_settings(sc, deg, bg=bgt, cov_grad=cov_grad, sh_grad=sh_grad, scale_modifier=scale_modifier)
    hm2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    kw = dict(means3D=hi["means3D"], means2D=hm2, opacities=hi["opacities"], viewmatrix=hi["viewmatrix"])
    kw.update(dict(colors_precomp=hi["colors_precomp"]) if use_colors else dict(shs=hi["shs"]))
    kw.update(dict(cov3Ds_precomp=hi["cov3Ds_precomp"]) if use_cov3d else dict(scales=hi["scales"],
                                                                               rotations=hi["rotations"]))
    hout = GaussianRasterizer(rs)(**kw)
    from rodygs_amd.rasterizer import last_compositing_state
    hout = tuple(hout) + (last_compositing_state(),)
    wn = torch.randn(3, H, W, generator=gen) * normal_loss

    def loss_of(o, dev):
        ls = (o[0] * wc.to(dev)).sum().add((o[3] * wa.to(dev)).sum())
        if depth_loss:
            ls = ls.add((o[1] * wd.to(dev)).sum() * depth_loss)
        return ls.add((o[2] * wn.to(dev)).sum()) if normal_loss else ls

    loss_of(hout, DEV).backward()
    torch.cuda.synchronize()

    oi = inputs("cpu")
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], bgt, scale_modifier, sc["projmatrix"], deg,
                          enable_cov_grad=cov_grad, enable_sh_grad=sh_grad)
    om2 = torch.zeros(P, 3, requires_grad=True)
    okw = dict(shs=None if use_colors else oi["shs"], colors_precomp=oi.get("colors_precomp"),
               scales=None if use_cov3d else oi["scales"], rotations=None if use_cov3d else oi["rotations"],
               cov3Ds_precomp=oi.get("cov3Ds_precomp"))
    oout = O.rasterize(oi["means3D"], om2, oi["opacities"], oi["viewmatrix"], st, **okw)
    ol = loss_of(oout, "cpu")
    if ol.requires_grad:          # nothing visible -> the oracle image has no graph at all
        ol.backward()
    for t in list(oi.values()) + [om2]:
        if t.grad is None:
            t.grad = torch.zeros_like(t)
    return hi, hm2, hout, oi, om2, oout


def check_pair(res, grads):
    hi, hm2, hout, oi, om2, oout = res
    assert torch.equal(hout[4].cpu(), oout[4]), "radii"
    assert hout[4].dtype == torch.int32 and hout[5].numel() == 0
    # the per-pixel state the backward replays from: where it differs, a blend / stop decision was
    # flipped on an alpha = 1/255 or T = 1e-4 discontinuity -- the only cause the one-entry allowance of rel_ok is for
    fT, nc = hout[6]
    flips = flipped_pixels(fT, nc, oout[5]["final_T"], oout[5]["n_contrib"])
    for i, name in ((0, "color"), (1, "depth"), (2, "normal"), (3, "alpha")):
        rel_ok(hout[i], oout[i], outliers=OUTLIER_FRAC, what=name, flips=flips)
    for k in grads:
        rel_ok(hi[k].grad, oi[k].grad, outliers=OUTLIER_FRAC, what="d_" + k, flips=flips)
    rel_ok(hm2.grad, om2.grad, outliers=OUTLIER_FRAC, what="d_means2D", flips=flips)
    assert float(hm2.grad[:, 2].abs().max()) == 0.0
    # n_contrib is an index -> exact, except on the flipped pixels
    rel_ok(fT, oout[5]["final_T"], outliers=OUTLIER_FRAC, what="final_T", flips=flips)
    mism = (nc.cpu() != oout[5]["n_contrib"]).double().mean().item()
    assert mism <= OUTLIER_FRAC, f"n_contrib differs on {mism:.2e} of the pixels"


# ---- stage-level bit-exactness -------------------------------------------------------------------------------

@pytest.mark.parametrize("P,W,H,deg", [(1000, 256, 256, 0),        # BASELINE config 1
                                       (20000, 640, 360, 3),
                                       (7000, 333, 211, 2),          # ragged: W,H not multiples of 16
                                       (100000, 1920, 1080, 3)])     # BASELINE config 2
@pytest.mark.parametrize("cull", [False, True], ids=["reference_rects", "tight_rects"])
def test_preprocess_and_binning_bit_exact(P, W, H, deg, cull):
    """cull = False: the reference algorithm's key stream (the north_star's bit-exact bar); cull = True: the product default's
    (tight rectangles, restated by the oracle operation for operation; tests/test_oracle_cull.py ties it to the reference's)."""
    import hip_stages as HS
    sc = O.synthetic_scene(P, W, H, 3, seed=P % 97)
    if P == 7000:
        sc["viewmatrix"] = orbit_view()
    hs = HS.run_stages(sc, deg, cull=cull)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], deg, cull=cull)
    with torch.no_grad():
        g = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                         scales=sc["scales"], rotations=sc["rotations"])
    b = O.bin_and_sort(g)
    vis = g["valid"].numpy()
    assert hs["D"] == b["num_rendered"]
    assert np.array_equal(hs["radii"], g["radii"].numpy())
    assert np.array_equal(hs["tiles_touched"], g["tiles_touched"].numpy().astype(np.uint32))
    assert np.array_equal(hs["depth"].view(np.uint32)[vis], g["depth"].numpy().view(np.uint32)[vis])
    xy = torch.stack([g["px"], g["py"]], 1).numpy()
    assert np.array_equal(hs["xy"].view(np.uint32)[vis], xy.view(np.uint32)[vis])
    co = torch.cat([g["conic"], g["opacity"].unsqueeze(1)], 1).numpy()
    assert np.array_equal(hs["conic_opacity"].view(np.uint32)[vis], co.view(np.uint32)[vis])
    rel_ok(hs["rgb"][vis], g["rgb"].numpy()[vis], tol=1e-5, what="rgb")
    rel_ok(hs["normal"][vis], g["normal"].numpy()[vis], tol=1e-6, what="normal")
    assert np.array_equal(hs["keys_unsorted"], b["keys_unsorted"])
    assert np.array_equal(hs["vals_unsorted"], b["vals_unsorted"])
    assert np.array_equal(hs["keys_sorted"], b["keys_sorted"])
    assert np.array_equal(hs["vals_sorted"], b["vals_sorted"])
    assert np.array_equal(hs["ranges"], b["ranges"])


@pytest.mark.parametrize("n,bits", [(0, 40), (1, 40), (63, 33), (64, 45), (1000, 40), (100000, 45), (3000000, 47),
                                    (77, 64), (5000, 8),
                                    (5000000, 40), (5000000, 16)])     # >= 2^22 pairs: the 4096-pair-tile kernels
def test_sort_pairs_stable_and_exact(n, bits):
    from rodygs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(n + bits)
    cap = max(n, 1)
    keys = torch.randint(0, 2 ** 62, (cap,), generator=g, dtype=torch.int64) & ((1 << min(bits, 62)) - 1)
    if n > 8:
        keys[::7] = keys[0]            # many duplicates: stability decides the value order
    vals = torch.arange(cap, dtype=torch.int32)
    kd, vd = keys.to(DEV), vals.to(DEV)
    nd = torch.tensor([n], dtype=torch.int32, device=DEV)
    tmp = torch.empty(L.rdg_sort_tmp_bytes(cap), dtype=torch.uint8, device=DEV)
    _lib.check(L.rdg_sort_pairs(kd.data_ptr(), vd.data_ptr(), cap, nd.data_ptr(), bits, tmp.data_ptr(),
                                _lib.stream_ptr()), "sort")
    torch.cuda.synchronize()
    kn = keys.numpy()[:n].view(np.uint64)
    order = np.argsort(kn, kind="stable")
    assert np.array_equal(kd.cpu().numpy()[:n].view(np.uint64), kn[order])
    assert np.array_equal(vd.cpu().numpy()[:n], vals.numpy()[:n][order])


# ---- full forward + backward ----------------------------------------------------------------------------------

def test_config1_forward_backward():
    """BASELINE config 1 shape (1k Gaussians, 256x256, SH degree 0), here with gradients as well."""
    sc = O.synthetic_scene(1000, 256, 256, 3, seed=2)
    check_pair(run_pair(sc, 0, (0.0, 0.0, 0.0)), NAMES)


def test_sh3_nonidentity_pose_background_ragged_image():
    sc = O.synthetic_scene(6000, 333, 211, 3, seed=4)
    sc["viewmatrix"] = orbit_view()
    check_pair(run_pair(sc, 3, (0.1, 0.2, 0.3)), NAMES)


@pytest.mark.parametrize("deg_max,deg,P", [(0, 0, 700), (1, 1, 2111), (2, 2, 3000), (2, 1, 513), (3, 3, 64)])
def test_sh_storage_widths(deg_max, deg, P):
    """shs stored as [P,1,3] / [P,4,3] / [P,9,3] (the wave-level LDS staging of the SH rows takes its scalar path for
    rows of 3 and 27 floats, its float4 path for 12 and 48), active degree <= stored degree, P not a multiple of 64."""
    sc = O.synthetic_scene(P, 200, 136, deg_max, seed=40 + deg_max)
    assert sc["shs"].shape[1] == (deg_max + 1) ** 2
    sc["viewmatrix"] = orbit_view(5.0, 3.0, (0.2, 0.1, 0.4))
    check_pair(run_pair(sc, deg, (0.05, 0.0, 0.1)), NAMES)


@pytest.mark.parametrize("cov_grad,sh_grad", [(False, False), (True, False), (False, True)])
def test_pose_gradient_gates(cov_grad, sh_grad):
    sc = O.synthetic_scene(3000, 200, 152, 3, seed=6)
    sc["viewmatrix"] = orbit_view(5.0, 3.0, (0.2, 0.1, 0.5))
    check_pair(run_pair(sc, 2, (0.3, 0.3, 0.3), cov_grad=cov_grad, sh_grad=sh_grad), NAMES)


def test_precomputed_colors_and_covariance_paths():
    sc = O.synthetic_scene(3000, 208, 160, 3, seed=8)
    sc["viewmatrix"] = orbit_view(-6.0, 4.0, (0.1, 0.2, 0.3))
    res = run_pair(sc, 0, (0.0, 0.1, 0.0), use_colors=True, use_cov3d=True, scale_modifier=1.0)
    check_pair(res, ("means3D", "opacities", "viewmatrix", "colors_precomp", "cov3Ds_precomp"))
    res = run_pair(sc, 1, (0.0, 0.0, 0.0), scale_modifier=0.7)
    check_pair(res, NAMES)


def test_non_unit_quaternions_are_not_renormalised():
    """RoDyGS feeds normalize(q) + dq: the kernel must use the raw quaternion."""
    sc = O.synthetic_scene(2000, 160, 128, 3, seed=9)
    sc["rotations"] = sc["rotations"] + 0.15 * torch.randn(2000, 4, generator=torch.Generator().manual_seed(1))
    check_pair(run_pair(sc, 1, (0.0, 0.0, 0.0)), NAMES)


def test_culled_and_degenerate_inputs():
    from rodygs_amd import GaussianRasterizer
    import hip_stages as HS
    sc = O.synthetic_scene(1500, 128, 96, 3, seed=10)
    sc["means3D"][:100, 2] = -2.0      # behind the camera
    sc["means3D"][100:150, 2] = 0.15   # inside the near cull
    sc["opacities"][150:200] = 0.0     # never reach 1/255
    check_pair(run_pair(sc, 3, (0.2, 0.2, 0.2)), NAMES)
    # everything culled -> background image, zero gradients, radii all zero
    sc2 = O.synthetic_scene(300, 64, 48, 3, seed=1)
    sc2["means3D"][:, 2] = -1.0
    res = run_pair(sc2, 3, (0.5, 0.25, 0.125))
    check_pair(res, ())
    assert float(res[0]["means3D"].grad.abs().max()) == 0.0 and int(res[2][4].abs().sum()) == 0
    # P = 0
    rs = HS.make_settings(sc2, 3, bg=torch.tensor([0.5, 0.25, 0.125]))
    z = torch.zeros(0, 3, device=DEV)
    out = GaussianRasterizer(rs)(means3D=z, means2D=z, shs=torch.zeros(0, 16, 3, device=DEV),
                                 opacities=torch.zeros(0, 1, device=DEV), scales=z, rotations=torch.zeros(0, 4, device=DEV),
                                 viewmatrix=sc2["viewmatrix"].to(DEV))
    assert torch.allclose(out[0][:, 0, 0].cpu(), torch.tensor([0.5, 0.25, 0.125])) and out[4].numel() == 0


def test_non_finite_gaussians_do_not_poison_the_frame():
    """A Gaussian whose scale / rotation / position is NaN or inf (a diverged optimiser row) must not turn the image or
    the other Gaussians' gradients into NaN: its conic is not positive definite, so no pixel blends it (the compositing
    kernels drop such a splat when they stage it; the per-Gaussian stage culls most of them before that)."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    sc = O.synthetic_scene(3000, 160, 128, 3, seed=21)
    bad = torch.arange(0, 3000, 250)
    good = torch.ones(3000, dtype=torch.bool); good[bad] = False

    def render(scene):
        ins = {k: scene[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
        rs = HS.make_settings(scene, 3, bg=torch.tensor([0.1, 0.2, 0.3]))
        m2 = torch.zeros(3000, 3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                     scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
        (out[0].sum() + 0.1 * out[1].sum()).backward()
        torch.cuda.synchronize()
        return out, ins

    ref = {k: v.clone() if torch.is_tensor(v) else v for k, v in sc.items()}
    ref["opacities"][bad] = 0.0                       # the same frame without those Gaussians
    out_ref, ins_ref = render(ref)
    poisoned = {k: v.clone() if torch.is_tensor(v) else v for k, v in sc.items()}
    poisoned["scales"][bad[0::4]] = float("nan")
    poisoned["scales"][bad[1::4], 1] = float("inf")
    poisoned["rotations"][bad[2::4]] = float("nan")
    poisoned["means3D"][bad[3::4], 0] = float("nan")
    poisoned["opacities"][bad[0]] = float("nan")
    poisoned["opacities"][bad[5]] = -0.7
    out, ins = render(poisoned)
    for i in range(4):
        assert bool(torch.isfinite(out[i]).all()), i
        rel_ok(out[i], out_ref[i].detach(), tol=1e-5, what=f"output {i}")
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        g = ins[k].grad[good.to(DEV)]
        assert bool(torch.isfinite(g).all()), k
        rel_ok(g, ins_ref[k].grad[good.to(DEV)], tol=1e-4, what="d_" + k)


def test_render_without_the_normal_channels_is_the_same_frame():
    """`rasterizer.RENDER_NORMAL = False` (the compositing forward without its normal accumulators: another kernel
    instantiation) gives bit-identical colour / depth / alpha / radii and gradients, and a zero normal image."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer, rasterizer
    sc = O.synthetic_scene(4000, 200, 152, 3, seed=33)

    def render():
        ins = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
        rs = HS.make_settings(sc, 3, bg=torch.tensor([0.3, 0.1, 0.2]))
        m2 = torch.zeros(4000, 3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                     scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
        w = torch.linspace(0.5, 1.5, 200, device=DEV)
        ((out[0] * w).sum() + 0.1 * (out[1] * w).sum() + (out[3] * w).sum()).backward()
        torch.cuda.synchronize()
        return out, ins

    with_n, ins_n = render()
    try:
        rasterizer.RENDER_NORMAL = False
        without, ins_w = render()
    finally:
        rasterizer.RENDER_NORMAL = True
    for i in (0, 1, 3, 4):
        assert torch.equal(with_n[i], without[i]), i
    assert float(without[2].detach().abs().max()) == 0.0 and float(with_n[2].detach().abs().max()) > 0.0
    for k in ("means3D", "opacities", "scales", "rotations", "viewmatrix"):
        rel_ok(ins_w[k].grad, ins_n[k].grad, tol=1e-5, outliers=1e-4, cap=1e-4, what="d_" + k)     # float atomics: not bitwise


def test_capacity_overflow_retries_and_retain_graph():
    """A too-small binning capacity must be detected on the device and retried, and backward must be repeatable
    (loss.backward(retain_graph=True) at /root/reference/src/trainer/rodygs.py:310)."""
    from rodygs_amd import GaussianRasterizer, rasterizer
    import hip_stages as HS
    sc = O.synthetic_scene(3000, 320, 240, 3, seed=12)
    sc["scales"] = sc["scales"] * 6.0   # many tiles per Gaussian: D >> 4 P
    rs = HS.make_settings(sc, 1)
    ins = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    rasterizer._CAPACITY_HINT.clear()
    m2 = torch.zeros(3000, 3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                 scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
    D = rasterizer._CAPACITY_HINT[(3000, 240, 320)]
    assert D > 4 * 3000 + 4096, "scene did not overflow the initial capacity; enlarge scales"
    loss = out[0].sum()
    loss.backward(retain_graph=True)
    g1 = ins["means3D"].grad.clone()
    ins["means3D"].grad = None
    loss.backward()
    rel_ok(ins["means3D"].grad, g1, tol=1e-5, outliers=1e-4, cap=1e-4, what="repeat backward")     # float atomics: summation order
    st = O.OracleSettings(240, 320, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 1)
    with torch.no_grad():
        oc = O.rasterize(sc["means3D"], torch.zeros(3000, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                         scales=sc["scales"], rotations=sc["rotations"])
    rel_ok(out[0], oc[0], outliers=OUTLIER_FRAC, what="color after retry")


def test_instance_count_beyond_32_bits_raises_instead_of_indexing_with_a_wrapped_count():
    """Garbage scales (a diverged run, an uninitialised buffer): every Gaussian covers every tile and the (tile, Gaussian)
    instance count passes 2^31.  The device total saturates (rdg_scan_block_sums_kernel), every later kernel of the
    frame leaves, and the host raises -- a wrapped 32-bit count would have passed the capacity check and indexed out of
    bounds.  The next, sane frame renders normally."""
    from rodygs_amd import GaussianRasterizer, rasterizer
    import hip_stages as HS
    P, W, H = 300000, 1920, 1080                               # 300 k x 8160 tiles = 2.4e9 instances
    sc = O.synthetic_scene(P, W, H, 0, seed=5)
    rs = HS.make_settings(sc, 0)
    ins = {k: sc[k].clone().to(DEV) for k in NAMES}
    big = dict(ins, scales=torch.full_like(ins["scales"], 1.0e4))
    rasterizer._CAPACITY_HINT.pop((P, H, W), None)

    def fwd(d):
        return GaussianRasterizer(rs)(means3D=d["means3D"], means2D=torch.zeros(P, 3, device=DEV), shs=d["shs"],
                                      opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"],
                                      viewmatrix=d["viewmatrix"])
    with pytest.raises(RuntimeError, match="2\\^31 - 1 or more"):
        fwd(big)
    torch.cuda.synchronize()
    out = fwd(ins)
    assert 0 < rasterizer._CAPACITY_HINT[(P, H, W)] < 2 ** 31 - 1
    assert bool(torch.isfinite(out[0]).all()) and float(out[0].abs().max()) > 0


@pytest.mark.parametrize("mirror", [True, False])
def test_deferred_instance_count_check(mirror):
    """rasterizer.DEFERRED_OVERFLOW_CHECK (what bench.py's timed region runs with): the forward does not wait for the
    instance count -- the binning stage mirrors (D, largest list) into pinned host memory (RdgRasterSettings
    .num_rendered_host; `mirror` False: the copy-engine form) and poll_overflow() reads it later.  A frame whose count
    fits: the hint follows the device value.  A frame that outgrows a stale hint: rendered EMPTY (background, zero
    gradients), RasterizerCapacityOverflow at the poll, and the re-run with the raised hint matches the oracle."""
    from rodygs_amd import GaussianRasterizer, rasterizer
    import hip_stages as HS
    P = 3000
    sc = O.synthetic_scene(P, 320, 240, 3, seed=12)
    big = dict(sc, scales=sc["scales"] * 6.0)                      # many tiles per Gaussian: D >> 4 P + 4096
    bg = torch.tensor([0.1, 0.2, 0.3])
    rs = HS.make_settings(sc, 1, bg=bg)
    key = (P, 240, 320)

    def fwd(scene, grad=False):
        ins = {k: scene[k].clone().to(DEV).requires_grad_(grad) for k in NAMES}
        out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=torch.zeros(P, 3, device=DEV, requires_grad=grad),
                                     shs=ins["shs"], opacities=ins["opacities"], scales=ins["scales"],
                                     rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
        return ins, out

    keep = (rasterizer.DEFERRED_OVERFLOW_CHECK, rasterizer.NREN_HOST_MIRROR)
    try:
        rasterizer._CAPACITY_HINT.pop(key, None)
        rasterizer.NREN_HOST_MIRROR = mirror
        fwd(sc)                                                    # immediate mode: sets the hint
        d_small = rasterizer._CAPACITY_HINT[key]
        rasterizer.DEFERRED_OVERFLOW_CHECK = True
        _, out = fwd(sc)
        assert len(rasterizer._PENDING) == 1
        rasterizer.poll_overflow(block=True)
        assert not rasterizer._PENDING and rasterizer._CAPACITY_HINT[key] == d_small
        want = int(rasterizer.last_num_rendered()[0][0].item())
        assert want == d_small
        # a frame that needs far more instances than the stale hint allows
        ins, out = fwd(big, grad=True)
        out[0].sum().backward()
        torch.cuda.synchronize()
        assert torch.allclose(out[0].detach().cpu(), bg.view(3, 1, 1).expand(3, 240, 320))     # rendered empty
        assert float(ins["means3D"].grad.abs().max()) == 0.0
        with pytest.raises(rasterizer.RasterizerCapacityOverflow):
            rasterizer.poll_overflow(block=True)
        d_big = rasterizer._CAPACITY_HINT[key]
        assert d_big > 4 * P + 4096
        _, out = fwd(big)                                          # deferred again, now with room
        rasterizer.poll_overflow(block=True)
        st = O.OracleSettings(240, 320, sc["tanfovx"], sc["tanfovy"], bg, 1.0, sc["projmatrix"], 1)
        with torch.no_grad():
            oc = O.rasterize(big["means3D"], torch.zeros(P, 3), big["opacities"], big["viewmatrix"], st, shs=big["shs"],
                             scales=big["scales"], rotations=big["rotations"])
        rel_ok(out[0], oc[0], outliers=OUTLIER_FRAC, what="color after the deferred overflow")
    finally:
        rasterizer.DEFERRED_OVERFLOW_CHECK, rasterizer.NREN_HOST_MIRROR = keep
        rasterizer._PENDING.clear()
        rasterizer._CAPACITY_HINT.pop(key, None)


def test_render_wrapper_returns_reference_dict():
    """render() mirror of /root/reference/src/trainer/renderer.py:17-114: keys, shapes, and every VALUE of the dict
    (plus the gradients that flow back through it) against the oracle rendering the same camera."""
    from rodygs_amd import render
    sc = O.synthetic_scene(2000, 160, 120, 3, seed=13)
    sc["viewmatrix"] = orbit_view(3.0, -2.0, (0.1, 0.05, 0.2))
    bg = torch.tensor([0.2, 0.4, 0.1])

    class Cam:
        FoVx, FoVy = sc["fovx"], sc["fovy"]
        image_height, image_width = 120, 160
        projection_matrix = sc["projmatrix"].t().contiguous().to(DEV)      # un-transposed P, as FixedCameraTorch
        world_view_transform = sc["viewmatrix"].t().contiguous().to(DEV).requires_grad_(True)

    xyz = sc["means3D"].to(DEV).requires_grad_(True)
    shs = sc["shs"].to(DEV).requires_grad_(True)
    pkg = render(xyz, 3, sc["opacities"].to(DEV), sc["scales"].to(DEV), sc["rotations"].to(DEV), shs,
                 Cam, bg.to(DEV), enable_sh_grad=True, enable_cov_grad=True)
    from rodygs_amd.rasterizer import last_compositing_state
    fT, nc = last_compositing_state()
    assert set(pkg) == {"rendered_image", "rendered_depth", "rendered_normal", "rendered_alpha", "viewspace_points",
                        "visibility_filter", "radii", "extra"}
    assert pkg["rendered_image"].shape == (3, 120, 160) and pkg["rendered_depth"].shape == (1, 120, 160)
    assert pkg["visibility_filter"].dtype == torch.bool and pkg["visibility_filter"].shape == (2000,)
    g = torch.Generator().manual_seed(5)
    wc, wd = torch.rand(3, 120, 160, generator=g), torch.rand(1, 120, 160, generator=g)
    ((pkg["rendered_image"] * wc.to(DEV)).sum() + 0.1 * (pkg["rendered_depth"] * wd.to(DEV)).sum()).backward()
    # oracle with the arguments the reference's render() builds (renderer.py:50-101)
    oi = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
    om2 = torch.zeros(2000, 3, requires_grad=True)
    st = O.OracleSettings(120, 160, math.tan(sc["fovx"] * 0.5), math.tan(sc["fovy"] * 0.5), bg, 1.0, sc["projmatrix"], 3)
    oc, od, on, oa, orad, oaux = O.rasterize(oi["means3D"], om2, oi["opacities"], oi["viewmatrix"], st, shs=oi["shs"],
                                             scales=oi["scales"], rotations=oi["rotations"])
    ((oc * wc).sum() + 0.1 * (od * wd).sum()).backward()
    flips = flipped_pixels(fT, nc, oaux["final_T"], oaux["n_contrib"])
    for key, ref in (("rendered_image", oc), ("rendered_depth", od), ("rendered_normal", on), ("rendered_alpha", oa)):
        rel_ok(pkg[key], ref, outliers=OUTLIER_FRAC, what=key, flips=flips)
    assert torch.equal(pkg["radii"].cpu(), orad) and torch.equal(pkg["visibility_filter"].cpu(), orad > 0)
    assert pkg["extra"].numel() == 0
    rel_ok(pkg["viewspace_points"].grad, om2.grad, outliers=OUTLIER_FRAC, what="viewspace_points.grad", flips=flips)
    rel_ok(xyz.grad, oi["means3D"].grad, outliers=OUTLIER_FRAC, what="d_xyz through render()", flips=flips)
    rel_ok(shs.grad, oi["shs"].grad, outliers=OUTLIER_FRAC, what="d_shs through render()", flips=flips)
    # the camera's world_view_transform is W2C; the rasterizer saw its transpose
    rel_ok(Cam.world_view_transform.grad, oi["viewmatrix"].grad.t(), outliers=OUTLIER_FRAC, what="d_world_view_transform", flips=flips)


# ---- the committed rasterizer fixture and the long-list paths ---------------------------------

def _fixture_scene(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"rasterizer_golden_{name}.npz"))
    sc = {k: torch.from_numpy(g["in_" + k]) for k in NAMES}
    sc.update(projmatrix=torch.from_numpy(g["in_projmatrix"]), tanfovx=float(g["in_tanfovx"]),
              tanfovy=float(g["in_tanfovy"]), W=int(g["in_W"]), H=int(g["in_H"]))
    return g, sc, int(g["in_sh_degree"]), torch.from_numpy(g["in_bg"])


@pytest.mark.parametrize("scene", ["c1", "skewed"])
def test_hip_matches_committed_rasterizer_fixture(scene):
    """HIP path against tests/golden/rasterizer_golden_*.npz (oracle output frozen by make_rasterizer_golden.py):
    binning integers bit for bit, images / per-pixel state / every gradient to 1e-4."""
    import importlib.util
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    from rodygs_amd.rasterizer import last_compositing_state
    spec = importlib.util.spec_from_file_location(
        "make_rasterizer_golden", os.path.join(os.path.dirname(__file__), "golden", "make_rasterizer_golden.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    g, sc, deg, bg = _fixture_scene(scene)
    P = sc["means3D"].shape[0]
    for cull, pre in ((False, ""), (True, "cull_")):       # the reference's rectangles, then the tight ones
        hs = HS.run_stages(sc, deg, cull=cull)
        assert hs["D"] == int(g[pre + "num_rendered"])
        assert np.array_equal(hs["radii"], g["radii"])
        assert np.array_equal(hs["tiles_touched"], g[pre + "tiles_touched"].astype(np.uint32))
        for k in ("keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted", "ranges"):
            assert np.array_equal(hs[k], g[pre + k]), (cull, k)
    hi = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(HS.make_settings(sc, deg, bg=bg))(
        means3D=hi["means3D"], means2D=m2, shs=hi["shs"], opacities=hi["opacities"], scales=hi["scales"],
        rotations=hi["rotations"], viewmatrix=hi["viewmatrix"])
    fT, nc = last_compositing_state()
    M.fixture_loss(out[0], out[1], out[3]).backward()
    assert np.array_equal(out[4].cpu().numpy(), g["radii"])
    # the product path runs with the tight rectangles unless told otherwise: n_contrib is a position in THOSE lists
    import rodygs_amd.rasterizer as R
    g_nc = g["cull_n_contrib"] if R.DEFAULT_STATE.mode("cull") else g["n_contrib"]
    flips = flipped_pixels(fT, nc, torch.from_numpy(g["final_T"]), torch.from_numpy(g_nc.astype(np.int64)))
    for i_, k in ((0, "color"), (1, "depth"), (2, "normal"), (3, "alpha")):
        rel_ok(out[i_], g[k], outliers=OUTLIER_FRAC, what="fixture " + k, flips=flips)
    rel_ok(fT, g["final_T"], outliers=OUTLIER_FRAC, what="fixture final_T", flips=flips)
    assert (nc.cpu().numpy() != g_nc).mean() <= OUTLIER_FRAC
    for k in NAMES:
        rel_ok(hi[k].grad, g["grad_" + k], outliers=OUTLIER_FRAC, what="fixture d_" + k, flips=flips)
    rel_ok(m2.grad, g["grad_means2D"], outliers=OUTLIER_FRAC, what="fixture d_means2D", flips=flips)


def _hip_grads(sc, deg, bg, deterministic, with_depth=True, seed=11, normal_loss=0.0):
    """One forward + backward through the drop-in surface; returns every input gradient (incl. means2D, viewmatrix)."""
    import hip_stages as HS
    import rodygs_amd.rasterizer as R
    from rodygs_amd import GaussianRasterizer
    P, H, W = sc["means3D"].shape[0], sc["H"], sc["W"]
    gen = torch.Generator().manual_seed(seed)
    wc, wd, wa = (torch.rand(3, H, W, generator=gen).to(DEV), torch.rand(1, H, W, generator=gen).to(DEV),
                  torch.rand(1, H, W, generator=gen).to(DEV))
    wn = (torch.randn(3, H, W, generator=gen) * normal_loss).to(DEV)          # same draw order as run_pair
    hi = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    old = R.DETERMINISTIC
    R.DETERMINISTIC = deterministic
    try:
        out = GaussianRasterizer(HS.make_settings(sc, deg, bg=torch.tensor(bg)))(
            means3D=hi["means3D"], means2D=m2, shs=hi["shs"], opacities=hi["opacities"], scales=hi["scales"],
            rotations=hi["rotations"], viewmatrix=hi["viewmatrix"])
        loss = (out[0] * wc).sum() + (out[3] * wa).sum()
        if with_depth:
            loss = loss + 0.1 * (out[1] * wd).sum()
        if normal_loss:
            loss = loss + (out[2] * wn).sum()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        R.DETERMINISTIC = old
    g = {k: hi[k].grad.clone() for k in NAMES}
    g["means2D"] = m2.grad.clone()
    return g


@pytest.mark.parametrize("scene", ["uniform", "skewed", "no_depth"])
def test_deterministic_backward_is_bit_reproducible(scene):
    """RDG_DETERMINISTIC (rdg_composite_backward_det) accumulates without float atomics -- two runs
    give the SAME BITS for every gradient, the atomic path agrees to 1e-5 per column (summation order only), and the
    oracle to the usual 1e-4.  `skewed`: a 23 k-instance tile + depth ties (multi-workgroup sort, equal composites'
    neighbours in the binary search); `no_depth`: the kernel variant without a depth gradient."""
    if scene == "skewed":
        W, H = 320, 240
        sc = O.skewed_scene(W, H, [(5, 6, 23000), (14, 3, 7900), (9, 11, 4200), (2, 2, 2500)], background=3000,
                            sh_degree_max=3, seed=78, equal_depth_every=5)
        sc["viewmatrix"] = orbit_view(1.0, -0.7, (0.04, -0.02, 0.08))
    else:
        sc = O.synthetic_scene(20000, 640, 360, 3, seed=41)
        sc["viewmatrix"] = orbit_view()
    bg = (0.1, 0.2, 0.3)
    depth = scene != "no_depth"
    a = _hip_grads(sc, 3, bg, True, depth)
    b = _hip_grads(sc, 3, bg, True, depth)
    for k in a:
        assert torch.equal(a[k], b[k]), f"deterministic mode: d_{k} differs between two runs"
        assert bool(torch.isfinite(a[k]).all())
    at = _hip_grads(sc, 3, bg, False, depth)
    for k in a:
        rel_ok(a[k], at[k], tol=1e-5, outliers=1e-4, cap=1e-4, what=f"deterministic vs atomic d_{k}")
    if scene == "uniform":
        res = run_pair(sc, 3, bg, seed=11)     # same loss weights (seed) as _hip_grads
        oi, om2 = res[3], res[4]
        flips = flipped_pixels(res[2][6][0], res[2][6][1], res[5][5]["final_T"], res[5][5]["n_contrib"])   # same forward
        for k in NAMES:
            rel_ok(a[k], oi[k].grad, outliers=OUTLIER_FRAC, what=f"deterministic vs oracle d_{k}", flips=flips)
        rel_ok(a["means2D"], om2.grad, outliers=OUTLIER_FRAC, what="deterministic vs oracle d_means2D", flips=flips)


def test_deterministic_backward_full_size_1m_1080p():
    """Bit reproducibility at BASELINE configs[2] size (1 M Gaussians, 1080p, D ~ 3.5 M), and agreement with the atomic
    path per column."""
    from rodygs_amd import synthetic
    sc = synthetic.synthetic_scene(1_000_000, 1920, 1080, 3, seed=777)
    a = _hip_grads(sc, 3, (0.0, 0.0, 0.0), True)
    b = _hip_grads(sc, 3, (0.0, 0.0, 0.0), True)
    for k in a:
        assert torch.equal(a[k], b[k]), f"deterministic mode at 1 M: d_{k} differs between two runs"
    at = _hip_grads(sc, 3, (0.0, 0.0, 0.0), False)
    for k in a:
        # summation order only: <= 2e-5 of the column's scale, up to a handful of the 1 M rows just above it (seen: one
        # quaternion component at 2.2e-5 on one run of the float-atomic path), none above 1e-4
        rel_ok(a[k], at[k], tol=2e-5, outliers=1e-5, cap=1e-4, what=f"deterministic vs atomic at 1 M: d_{k}")


def _tile_counts(ranges):
    r = ranges.astype(np.int64)
    return np.sort(r[:, 1] - r[:, 0])


@pytest.mark.parametrize("heavy", [23000, 70000])
def test_skewed_scene_long_tile_lists(heavy):
    """Densified scenes are not uniform (configs/train/train_kubric_mrig.yaml:168-173): one tile holding `heavy`
    instances (the multi-workgroup merge path of the per-tile sort), tiles at 2-8 k (its LDS path), a few at 1-2 k,
    depth ties, and compositing thousands of splats deep.  Keys / order / ranges bit-exact, image + every gradient
    <= 1e-4 against the oracle."""
    import hip_stages as HS
    W, H = 320, 240
    sc = O.skewed_scene(W, H, [(5, 6, heavy), (14, 3, 7900), (9, 11, 4200), (2, 2, 2500), (17, 12, 1300)],
                        background=3000, sh_degree_max=3, seed=77, equal_depth_every=5)
    sc["viewmatrix"] = orbit_view(1.0, -0.7, (0.04, -0.02, 0.08))
    P = sc["means3D"].shape[0]
    hs = HS.run_stages(sc, 3)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 3)
    with torch.no_grad():
        g = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                         scales=sc["scales"], rotations=sc["rotations"])
    b = O.bin_and_sort(g)
    n = _tile_counts(b["ranges"])
    # (the tight rectangles drop a sixth of the cluster's instances)
    assert n[-1] > max(8192, heavy * 0.75) and ((n > 2048) & (n <= 8192)).sum() >= 2 and ((n > 1024) & (n <= 2048)).sum() >= 1
    assert (b["keys_sorted"][1:] == b["keys_sorted"][:-1]).sum() > 1000
    assert hs["D"] == b["num_rendered"]
    for k in ("keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted", "ranges"):
        assert np.array_equal(hs[k], b[k]), k
    res = run_pair(sc, 3, (0.1, 0.2, 0.3))
    check_pair(res, NAMES)
    assert int(res[5][5]["n_contrib"].max()) > 8192      # compositing really walked the long list


@pytest.mark.parametrize("deterministic", [False, True])
def test_split_compositing_of_long_tile_lists(deterministic):
    """Lists above 4096 instances composited by several workgroups (per-segment partial composites + ordered combine,
    RdgRasterSettings.split_lists) against the oracle AND against the one-workgroup walk of the same frame: a 70 k-
    instance tile (35 segments), tiles at 4.2-7.9 k (3-4 segments, ragged last segment), shorter ones untouched, depth
    ties, coloured background.  Images, final_T and every gradient to 1e-4 / oracle and 2e-5 / unsplit; n_contrib exact
    up to the discontinuity allowance; pixels that stop early inside a segment (the opaque variant) included."""
    import hip_stages as HS
    import rodygs_amd.rasterizer as R
    from rodygs_amd import GaussianRasterizer
    W, H = 320, 240
    bg = (0.1, 0.2, 0.3)
    for opaque in (False, True):
        sc = O.skewed_scene(W, H, [(5, 6, 70000), (14, 3, 7900), (9, 11, 4200), (2, 2, 2500), (17, 12, 1300)],
                            background=3000, sh_degree_max=3, seed=79, equal_depth_every=5,
                            opacity=(0.02, 0.5) if opaque else (0.006, 0.03))
        sc["viewmatrix"] = orbit_view(1.0, -0.7, (0.04, -0.02, 0.08))
        P = sc["means3D"].shape[0]
        key = (P, H, W)
        old_det = R.DETERMINISTIC
        R.DETERMINISTIC = deterministic
        try:
            R._SPLIT_HINT.pop(key, None)
            R.SPLIT_ABOVE, keep = 10 ** 9, R.SPLIT_ABOVE          # the hint can never set: one workgroup per tile
            try:
                unsplit = _hip_grads(sc, 3, bg, deterministic)
            finally:
                R.SPLIT_ABOVE = keep
            R._SPLIT_HINT[key] = 1
            res = run_pair(sc, 3, bg, seed=11)                    # HIP (split path) + oracle, same loss weights
            assert R._SPLIT_HINT.get(key) == 1                    # the frame's largest list keeps the hint set
            split = {k: res[0][k].grad for k in NAMES}
            split["means2D"] = res[1].grad
        finally:
            R.DETERMINISTIC = old_det
            R._SPLIT_HINT.pop(key, None)
        check_pair(res, NAMES)
        if not opaque:  # thousands of splats deep, and pixels that saturate only in a LATER segment of the long list
            nc, fT = res[5][5]["n_contrib"], res[5][5]["final_T"]
            assert int(nc.max()) > 8192 and int(((fT < 2e-4) & (nc > 2048)).sum()) > 0
        for k in split:
            rel_ok(split[k], unsplit[k], tol=2e-5, outliers=OUTLIER_FRAC, what=f"split vs one-workgroup d_{k} (opaque={opaque})")


@pytest.mark.parametrize("case", ["uniform", "no_depth", "split", "split_det"])
def test_rendered_normal_backward(case):
    """A loss on rendered_normal: the per-Gaussian normals are constants of the graph (oracle/rasterizer_oracle.py:268
    detaches them, as the image the reference's rasterizer returns has them precomputed), so the gradient reaches
    opacity / conic / position / pose through the compositing weights only.  Same 1e-4 per column as every other
    gradient; with and without a depth gradient (two kernel variants); through the segment-parallel backward of lists
    above 4096 instances (the behind-sums of the normal channels cross segments); deterministic mode gives the same bits
    twice.  A normal gradient with the normal channels switched off raises."""
    import rodygs_amd.rasterizer as R
    bg = (0.1, 0.2, 0.3)
    if case.startswith("split"):
        W, H = 320, 240
        sc = O.skewed_scene(W, H, [(5, 6, 23000), (14, 3, 7900), (9, 11, 4200), (2, 2, 2500)], background=3000,
                            sh_degree_max=3, seed=78, equal_depth_every=5)
        sc["viewmatrix"] = orbit_view(1.0, -0.7, (0.04, -0.02, 0.08))
    else:
        sc = O.synthetic_scene(6000, 320, 200, 3, seed=43)
        sc["viewmatrix"] = orbit_view()
    key = (sc["means3D"].shape[0], sc["H"], sc["W"])
    det = case == "split_det"
    old = R.DETERMINISTIC
    R.DETERMINISTIC = det
    try:
        if case == "split":
            R._SPLIT_HINT[key] = 1
        res = run_pair(sc, 3, bg, seed=11, normal_loss=0.7, depth_loss=0.0 if case == "no_depth" else 0.1)
    finally:
        R.DETERMINISTIC = old
        R._SPLIT_HINT.pop(key, None)
    check_pair(res, NAMES)
    # the normal term is a real part of the gradient, not noise under the colour term
    base = _hip_grads(sc, 3, bg, det, with_depth=case != "no_depth")
    d = (res[0]["opacities"].grad - base["opacities"]).abs().max() / base["opacities"].abs().max()
    assert float(d) > 1e-2, float(d)
    if det:
        a = _hip_grads(sc, 3, bg, True, normal_loss=0.7)
        b = _hip_grads(sc, 3, bg, True, normal_loss=0.7)
        for k in a:
            assert torch.equal(a[k], b[k]), k
            ref = res[0][k].grad if k != "means2D" else res[1].grad
            assert torch.equal(a[k], ref), k
    if case == "uniform":
        import hip_stages as HS
        from rodygs_amd import GaussianRasterizer
        hi = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
        keep, R.RENDER_NORMAL = R.RENDER_NORMAL, False
        try:
            out = GaussianRasterizer(HS.make_settings(sc, 3))(
                means3D=hi["means3D"], means2D=torch.zeros(key[0], 3, device=DEV, requires_grad=True), shs=hi["shs"],
                opacities=hi["opacities"], scales=hi["scales"], rotations=hi["rotations"], viewmatrix=hi["viewmatrix"])
            with pytest.raises(RuntimeError, match="RENDER_NORMAL = False"):
                (out[0].sum() + out[2].sum()).backward()
        finally:
            R.RENDER_NORMAL = keep


# ---- full-size properties (BASELINE config 3 shape: 1 M Gaussians, 1080p) --------------------------------------

def test_full_size_properties_1m_1080p():
    """Too big for the oracle: size-independent properties instead -- sortedness of the key stream, a
    partition-of-unity check of the tile ranges, determinism of the forward, linearity of the backward in the
    upstream gradient, alpha/colour consistency."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    P, W, H = 1000000, 1920, 1080
    sc = O.synthetic_scene(P, W, H, 3, seed=777)
    hs = HS.run_stages(sc, 3)
    ks = hs["keys_sorted"]
    assert hs["D"] == int(hs["tiles_touched"].sum()) == len(ks)
    assert np.all(ks[1:] >= ks[:-1])
    eq = ks[1:] == ks[:-1]
    assert np.all(hs["vals_sorted"][1:][eq] > hs["vals_sorted"][:-1][eq])                 # stable
    assert np.array_equal(np.sort(hs["keys_unsorted"], kind="stable"), ks)                 # a permutation
    assert int((hs["ranges"][:, 1].astype(np.int64) - hs["ranges"][:, 0]).sum()) == hs["D"]
    tiles = (ks >> np.uint64(32)).astype(np.int64)
    t_check = np.unique(tiles)[::97]
    for t in t_check:
        s, e = hs["ranges"][t]
        assert tiles[s] == t and tiles[e - 1] == t and (e == len(ks) or tiles[e] != t)
    rs = HS.make_settings(sc, 3, bg=torch.tensor([1.0, 1.0, 1.0]))
    ins = {k: sc[k].to(DEV).requires_grad_(True) for k in NAMES}

    def fwd():
        m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
        return GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                      scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])

    a, b = fwd(), fwd()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])   # forward is deterministic
    assert float(a[3].min()) >= 0.0 and float(a[3].max()) <= 1.0
    # white background + colours in [0, inf): colour >= (1 - alpha) everywhere
    assert bool((a[0] >= (1.0 - a[3]) - 1e-5).all())
    gw = torch.rand(3, H, W, device=DEV)
    g1 = torch.autograd.grad(a[0], ins["shs"], gw, retain_graph=True)[0]
    g2 = torch.autograd.grad(a[0], ins["shs"], 2.5 * gw)[0]
    rel_ok(g2, 2.5 * g1, tol=2e-5, outliers=1e-4, cap=2e-4, what="backward linearity")      # float atomics: order-dependent rounding only


@pytest.mark.parametrize("P,W,H,n_sample", [(100000, 1920, 1080, 20),       # BASELINE configs[1]
                                            (1000000, 1920, 1080, 20),      # configs[2..3]
                                            (4000000, 3840, 2160, 10)])     # configs[4]: the largest size
def test_full_size_sampled_tiles_against_oracle(P, W, H, n_sample):
    """BASELINE's full sizes (the frames bench.py renders): the oracle is affordable on a SAMPLE of tiles (its
    per-Gaussian stage and binning run in full).  Image, depth and alpha of those tiles, and every gradient of a loss
    restricted to those tiles, against the HIP path rendering the whole frame."""
    from rodygs_amd import GaussianRasterizer
    import hip_stages as HS
    sc = O.synthetic_scene(P, W, H, 3, seed=777)
    sc["viewmatrix"] = orbit_view(4.0, -2.0, (0.3, -0.2, 0.5))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    subset = list(range(37, gx * gy, (gx * gy) // n_sample))[:n_sample]
    mask = torch.zeros(1, H, W)
    for t in subset:
        ty, tx = divmod(t, gx)
        mask[:, ty * 16:min(ty * 16 + 16, H), tx * 16:min(tx * 16 + 16, W)] = 1.0
    g = torch.Generator().manual_seed(1)
    wc, wd = torch.rand(3, H, W, generator=g) * mask, torch.rand(1, H, W, generator=g) * mask
    bg = torch.tensor([0.2, 0.1, 0.3])
    # oracle: full per-Gaussian stage + binning, compositing of the sampled tiles only
    oi = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], bg, 1.0, sc["projmatrix"], 3)
    om2 = torch.zeros(P, 3, requires_grad=True)
    geom = O.preprocess(oi["means3D"], om2, oi["opacities"], oi["viewmatrix"], st, shs=oi["shs"], scales=oi["scales"],
                        rotations=oi["rotations"])
    binning = O.bin_and_sort(geom)
    # north_star "bit-exact on tile keys / sort indices" AT THE FULL SIZES: the whole (tile | depth) key stream, the sorted
    # Gaussian indices, the tile ranges and D of the HIP binning against the oracle's, through both binning algorithms --
    # under the REFERENCE's tile rectangles (cull = False: the reference algorithm's stream) and under the tight ones the
    # product runs with (the oracle's default; every list a subsequence of the reference's: tests/test_oracle_cull.py)
    import dataclasses
    with torch.no_grad():
        geom_ref = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"],
                                dataclasses.replace(st, cull=False), shs=sc["shs"], scales=sc["scales"],
                                rotations=sc["rotations"])
    bins = {False: O.bin_and_sort(geom_ref), True: binning}
    assert st.cull and bins[True]["num_rendered"] < 0.85 * bins[False]["num_rendered"]
    del geom_ref
    for cull in (False, True):
        for bin_mode in (0, 1):
            hs = HS.run_stages(sc, 3, bin_mode=bin_mode, cull=cull)
            what = f"bin_mode {bin_mode}, cull {cull}"
            assert hs["D"] == bins[cull]["num_rendered"], what
            assert np.array_equal(hs["keys_sorted"], bins[cull]["keys_sorted"]), "sorted keys, " + what
            assert np.array_equal(hs["vals_sorted"], bins[cull]["vals_sorted"]), "sorted Gaussian indices, " + what
            assert np.array_equal(hs["ranges"], bins[cull]["ranges"]), "tile ranges, " + what
            del hs
    del bins
    img = O.render_tiles(geom, binning, st.bg, H, W, tile_subset=subset)
    ((img["color"] * wc).sum() + 0.1 * (img["depth"] * wd).sum()).backward()
    # HIP: the whole frame, loss weights zero outside the sample
    hi = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    hm2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(HS.make_settings(sc, 3, bg=bg))(means3D=hi["means3D"], means2D=hm2, shs=hi["shs"],
                                                             opacities=hi["opacities"], scales=hi["scales"],
                                                             rotations=hi["rotations"], viewmatrix=hi["viewmatrix"])
    from rodygs_amd.rasterizer import last_compositing_state
    fT, nc = last_compositing_state()
    ((out[0] * wc.to(DEV)).sum() + 0.1 * (out[1] * wd.to(DEV)).sum()).backward()
    m = mask.to(DEV)
    mb = mask[0].bool()
    flips = flipped_pixels(fT.cpu()[mb], nc.cpu()[mb], img["final_T"][mb], img["n_contrib"][mb])
    for i_, name in ((0, "color"), (1, "depth"), (3, "alpha")):
        rel_ok(out[i_] * m, img[name] * mask, outliers=OUTLIER_FRAC, what="sampled tiles " + name, flips=flips)
    for k in NAMES:
        rel_ok(hi[k].grad, oi[k].grad, outliers=OUTLIER_FRAC, what="sampled tiles d_" + k, flips=flips)
    rel_ok(hm2.grad, om2.grad, outliers=OUTLIER_FRAC, what="sampled tiles d_means2D", flips=flips)


def _full_frame_pair(P, W, H, seed=5, variant="uniform", sc=None):
    """run_pair on a BASELINE-size frame with an UNRESTRICTED loss (random weights on every pixel of colour, depth and
    alpha), packed for oracle.parity.full_frame_report.  sc: the rasterizer inputs to use instead of the section-8d generator's."""
    from rodygs_amd import rasterizer
    if sc is None:
        sc = O.synthetic_scene(P, W, H, 3, seed=777, variant=variant)
        sc["viewmatrix"] = orbit_view(4.0, -2.0, (0.3, -0.2, 0.5))
    P = sc["means3D"].shape[0]
    # the oracle's tensors are a few hundred KB each: beyond ~16 threads torch's intra-op fork / join costs more than it buys
    # (bench.py's CPU leg: 1 M / 1080p in 15 s on 16 threads of the 256-core box, 100 s on all of them)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))
    try:
        hi, hm2, hout, oi, om2, oout = run_pair(sc, 3, [0.2, 0.1, 0.3], seed=seed)
    finally:
        torch.set_num_threads(threads)
    aux = oout[5]
    fT, nc = hout[6]
    hip = {"images": {"color": hout[0], "depth": hout[1], "normal": hout[2], "alpha": hout[3]}, "final_T": fT, "n_contrib": nc,
           "radii": hout[4], "D": rasterizer.DEFAULT_STATE.capacity_hint[(P, H, W)],
           "grads": {**{k: hi[k].grad for k in NAMES}, "means2D": hm2.grad}}
    orc = {"images": {"color": oout[0], "depth": oout[1], "normal": oout[2], "alpha": oout[3]}, "final_T": aux["final_T"],
           "n_contrib": aux["n_contrib"], "radii": oout[4], "D": aux["binning"]["num_rendered"],
           "grads": {**{k: oi[k].grad for k in NAMES}, "means2D": om2.grad}}
    return hip, orc, aux["binning"], (W + 15) // 16


@pytest.mark.parametrize("P,W,H", [(100000, 1920, 1080),        # BASELINE configs[1]: "bit/tol parity" at 100 k / 1080p / SH 3
                                   (1000000, 1920, 1080),       # configs[2..3]: the frame the headline is quoted on
                                   (4000000, 3840, 2160)])      # configs[4]: the largest size, all 32 400 tiles
def test_full_frame_parity(P, W, H):
    """(the section-8d generator's untrained `uniform` cloud; the hard regimes: test_full_frame_parity_hard_regimes)
    EVERY tile of the frame: the oracle composites all 8 160 tiles forward and backward (seconds on the host), the loss
    is unrestricted, and every pixel of colour / depth / normal / alpha / final_T, n_contrib, radii, D and every entry of every
    gradient (incl. viewmatrix and means2D) is held to 1e-4 of its column -- no outlier fraction.  The only entries that
    may miss the bar (and then must stay below 2e-2) are the ones a WITNESSED decision flip explains: the flipped pixels
    themselves, and the gradient rows of the Gaussians standing in such a pixel's list up to its last contributor
    (oracle.parity.full_frame_report).  The flips themselves are bounded: at most 2e-5 of the pixels."""
    from oracle.parity import full_frame_report
    hip, orc, binning, gx = _full_frame_pair(P, W, H)
    rep = full_frame_report(hip, orc, binning["vals_sorted"], binning["ranges"], gx, tol=TOL, cap=FULL_FRAME_CAP)
    assert rep["ok"], (rep["violations"], rep["witnessed_flips"], rep["flip_candidate_gaussians"])
    assert rep["witnessed_flips"] <= OUTLIER_FRAC * H * W, rep
    assert rep["n_contrib_mismatch_off_flips"] == 0
    assert float(hip["grads"]["means2D"][:, 2].abs().max()) == 0.0


# ---- deformation, knn, adam ------------------------------------------------------------------------------------

@pytest.mark.parametrize("P,Tu,inverse", [(5000, 37, True), (1, 1, True), (4097, 100, True), (3000, 0, False),
                                          (2000, 200, True)])
def test_deformation_forward_backward(P, Tu, inverse):
    from rodygs_amd import gaussian_deformation
    g = torch.Generator().manual_seed(P + Tu)
    B = 16
    coeff = (0.1 * torch.randn(P, 1, B, generator=g)).requires_grad_(True)
    ind = torch.randint(0, max(Tu, 1), (P,), generator=g)
    bt = torch.randn(B, 7, generator=g).requires_grad_(True)
    tb = torch.randn(Tu, B, 7, generator=g).requires_grad_(True) if inverse else None
    wx, wr = torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)
    if inverse:
        ox, orr = DO.gaussian_deformation(coeff, ind, bt, tb, 1.7)
    else:
        d = coeff.reshape(P, B) @ bt
        ox, orr = d[:, :3] * 1.7, d[:, 3:]
    ((ox * wx).sum() + (orr * wr).sum()).backward()
    c2 = coeff.detach().clone().to(DEV).requires_grad_(True)
    bt2 = bt.detach().clone().to(DEV).requires_grad_(True)
    tb2 = tb.detach().clone().to(DEV).requires_grad_(True) if inverse else None
    hx, hr = gaussian_deformation(c2, ind.to(DEV), bt2, tb2, 1.7)
    ((hx * wx.to(DEV)).sum() + (hr * wr.to(DEV)).sum()).backward()
    rel_ok(hx, ox, tol=1e-5, what="dxyz"); rel_ok(hr, orr, tol=1e-5, what="drot")
    rel_ok(c2.grad, coeff.grad, tol=1e-5, what="d_coeff")
    rel_ok(bt2.grad, bt.grad, tol=2e-5, what="d_basis_t")
    if inverse:
        rel_ok(tb2.grad, tb.grad, tol=2e-5, what="d_table")


def test_deformation_packed_bases_and_coeff_sink():
    """gaussian_deformation_packed(bases[Tu+1]) + grad sink == gaussian_deformation(basis_t, table) through autograd."""
    from rodygs_amd import gaussian_deformation
    from rodygs_amd.deform import gaussian_deformation_packed
    g = torch.Generator().manual_seed(9)
    P, Tu, B = 3001, 23, 16
    coeff = (0.1 * torch.randn(P, 1, B, generator=g)).to(DEV)
    ind = torch.randint(0, Tu, (P,), generator=g).to(DEV)
    bases = torch.randn(Tu + 1, B, 7, generator=g).to(DEV)
    wx, wr = torch.randn(P, 3, generator=g).to(DEV), torch.randn(P, 4, generator=g).to(DEV)
    c1 = coeff.clone().requires_grad_(True)
    b1 = bases.clone().requires_grad_(True)
    x1, r1 = gaussian_deformation(c1, ind, b1[-1], b1[:-1], 2.5)
    ((x1 * wx).sum() + (r1 * wr).sum()).backward()
    c2 = coeff.clone().requires_grad_(True)
    b2 = bases.clone().requires_grad_(True)
    sink = torch.full((P, 1, B), float("nan"), device=DEV)
    x2, r2 = gaussian_deformation_packed(c2, ind, b2, 2.5, grad_sinks={"coeff": sink})
    ((x2 * wx).sum() + (r2 * wr).sum()).backward()
    assert torch.equal(x1, x2) and torch.equal(r1, r2)
    assert c2.grad is None
    rel_ok(sink, c1.grad, tol=1e-6, what="coeff sink")
    rel_ok(b2.grad, b1.grad, tol=1e-5, what="packed bases grad")


def test_mlp_and_pose_grad_sinks_match_autograd():
    from rodygs_amd.deform import MLPBasisNetwork
    from rodygs_amd.model_ops import pose_view_matrix
    torch.manual_seed(3)
    net = MLPBasisNetwork(128, 16, 26, False).to(DEV)
    x = torch.randn(37, 53, device=DEV)
    w = torch.randn(37, 16, 7, device=DEV)
    (net.motion_basis(x) * w).sum().backward()
    tn = net.timenet
    params = [tn[0].weight, tn[0].bias, tn[2].weight, tn[2].bias, tn[4].weight, tn[4].bias, net.head_w1, net.head_b1,
              net.head_w2, net.head_b2]
    want = [p.grad.clone() for p in params]
    for p_ in params:
        p_.grad = None
    net.grad_sinks = [torch.full_like(p_, float("nan")) for p_ in params]
    (net.motion_basis(x) * w).sum().backward()
    assert all(p_.grad is None for p_ in params)
    for a, b in zip(net.grad_sinks, want):
        assert torch.equal(a, b)
    q = torch.randn(5, 4, device=DEV, requires_grad=True)
    t = torch.randn(5, 3, device=DEV, requires_grad=True)
    wv = torch.randn(4, 4, device=DEV)
    (pose_view_matrix(q, t, 3) * wv).sum().backward()
    sq, st = torch.full_like(q, float("nan")), torch.full_like(t, float("nan"))
    q2, t2 = q.detach().clone().requires_grad_(True), t.detach().clone().requires_grad_(True)
    (pose_view_matrix(q2, t2, 3, grad_sinks={"q": sq, "t": st}) * wv).sum().backward()
    assert q2.grad is None and torch.equal(sq, q.grad) and torch.equal(st, t.grad)


def test_fused_adam_over_two_flat_buckets_matches_torch_groups():
    """One rdg_adam_step_multi launch over the Gaussian bucket + the small (MLP, pose) bucket, equal-lr neighbours
    merged, against torch.optim.Adam with one group per tensor."""
    from rodygs_amd.dp import FlatParams
    from rodygs_amd.trainstep import fused_adam_
    g = torch.Generator().manual_seed(21)
    spec_a = {"xyz": ((1001, 3), 1e-3), "features": ((1001, 4, 3), 2e-3), "scaling": ((1001, 3), 5e-3),
              "rotation": ((1001, 4), 5e-3)}
    spec_b = {"w0": ((33, 7), 1e-2), "b0": ((33,), 1e-2), "cam_q": ((9, 4), 1e-4), "cam_t": ((9, 3), 1e-5)}
    fa, fb = FlatParams(spec_a, DEV), FlatParams(spec_b, DEV)
    ref, groups = {}, []
    for f in (fa, fb):
        for k in f.names:
            v = torch.randn(f.shapes[k], generator=g)
            with torch.no_grad():
                f[k].copy_(v)
            ref[k] = v.clone().requires_grad_(True)
            if k == "features":
                continue
            groups.append({"params": [ref[k]], "lr": f.lr[k]})
    # features: DC row (first 3 floats of every 12) at lr 2e-3, the rest at 1e-4 -> two torch tensors
    fdc = ref["features"].detach()[:, :1].clone().requires_grad_(True)
    frest = ref["features"].detach()[:, 1:].clone().requires_grad_(True)
    groups += [{"params": [fdc], "lr": 2e-3}, {"params": [frest], "lr": 1e-4}]
    opt = torch.optim.Adam(groups, eps=1e-15)
    for step in range(3):
        for f in (fa, fb):
            for k in f.names:
                gk = torch.randn(f.shapes[k], generator=g) * (step + 1)
                f[k].grad.copy_(gk)
                if k == "features":
                    fdc.grad, frest.grad = gk[:, :1].clone(), gk[:, 1:].clone()
                else:
                    ref[k].grad = gk
        opt.step()
        fused_adam_(fa, row_lr={"features": (12, 3, 1e-4)}, extra=(fb,))
    for f in (fa, fb):
        for k in f.names:
            want = torch.cat([fdc, frest], dim=1) if k == "features" else ref[k]
            rel_ok(f[k], want.detach(), tol=2e-6, what="adam " + k)


@pytest.mark.parametrize("use_sinks", [False, True])
def test_fused_dynamic_getter_equals_deformation_plus_activations(use_sinks):
    """dynamic_gaussians (one kernel each way) vs gaussian_deformation_packed + activate_gaussians (each pinned to the
    reference / oracle by its own tests): outputs and all gradients."""
    from rodygs_amd.deform import dynamic_gaussians, gaussian_deformation_packed
    from rodygs_amd.model_ops import activate_gaussians
    g = torch.Generator().manual_seed(17)
    P, Tu, B = 4099, 31, 16
    raw = dict(xyz=torch.randn(P, 3, generator=g), scaling=torch.randn(P, 3, generator=g) - 2,
               rotation=torch.randn(P, 4, generator=g), opacity=torch.randn(P, 1, generator=g),
               coeff=0.2 * torch.randn(P, 1, B, generator=g), bases=torch.randn(Tu + 1, B, 7, generator=g))
    ind = torch.randint(0, Tu, (P,), generator=g).to(DEV)
    ws = [torch.randn(P, 3, generator=g).to(DEV), torch.randn(P, 3, generator=g).to(DEV),
          torch.randn(P, 4, generator=g).to(DEV), torch.randn(P, 1, generator=g).to(DEV)]
    a = {k: v.clone().to(DEV).requires_grad_(True) for k, v in raw.items()}
    dx, dr = gaussian_deformation_packed(a["coeff"], ind, a["bases"], 3.5)
    m, s_, r, o, _ = activate_gaussians(a["xyz"], dx, a["scaling"], a["rotation"], dr, a["opacity"], None, None)
    ((m * ws[0]).sum() + (s_ * ws[1]).sum() + (r * ws[2]).sum() + (o * ws[3]).sum()).backward()
    b = {k: v.clone().to(DEV).requires_grad_(True) for k, v in raw.items()}
    sinks = None
    if use_sinks:
        sinks = {k: torch.full_like(b[k], float("nan")) for k in ("xyz", "scaling", "rotation", "opacity")}
        sinks["coeff"] = torch.full((P, 1, B), float("nan"), device=DEV)
    m2, s2, r2, o2 = dynamic_gaussians(b["xyz"], b["scaling"], b["rotation"], b["opacity"], b["coeff"], ind, b["bases"],
                                       3.5, grad_sinks=sinks)
    ((m2 * ws[0]).sum() + (s2 * ws[1]).sum() + (r2 * ws[2]).sum() + (o2 * ws[3]).sum()).backward()
    for x, y, nm in ((m2, m, "means3D"), (s2, s_, "scales"), (r2, r, "rots"), (o2, o, "opac")):
        rel_ok(x, y, tol=2e-6, what="fused getter " + nm)
    for k in ("xyz", "scaling", "rotation", "opacity", "coeff"):
        got = sinks[k] if use_sinks else b[k].grad
        if use_sinks:
            assert b[k].grad is None
        rel_ok(got.reshape(a[k].grad.shape), a[k].grad, tol=2e-6, what="fused getter d_" + k)
    rel_ok(b["bases"].grad, a["bases"].grad, tol=2e-5, what="fused getter d_bases")


def test_unsorted_birth_order_poisons_the_basis_gradient():
    """The dB reduction stores one partial total per (wave, birth index) and adds them in a fixed order -- right only for
    a sequence that really is sorted by birth index.  A wrong permutation (a stale host cache) must not produce a
    plausible-looking gradient: the kernel notices and the basis gradient comes back NaN."""
    from rodygs_amd import deform as DF
    g = torch.Generator().manual_seed(3)
    P, Tu = 6000, 9
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(DEV)   # noqa: E731
    leaves = [t.requires_grad_(True) for t in (rnd(P, 3), 0.3 * rnd(P, 3), rnd(P, 4), rnd(P, 1), 0.2 * rnd(P, 16))]
    ti = torch.randint(0, Tu, (P,), generator=g).to(DEV)
    bases = (0.1 * rnd(Tu + 1, 16, 7)).requires_grad_(True)
    out = DF.dynamic_gaussians(*leaves, ti, bases, 2.0)
    order, inv, seg = DF._birth_order(ti, Tu)
    key = next(k for k, v in DF._ORDER_CACHE.items() if v[0] is order)
    bad = torch.roll(order, 1234)                     # still a permutation, no longer sorted by birth index
    bad_inv = torch.empty_like(bad)
    bad_inv[bad.long()] = torch.arange(P, device=DEV, dtype=torch.int32)
    try:
        DF._ORDER_CACHE[key] = (bad, bad_inv, seg, ti)
        torch.autograd.backward([out[0], out[2]], [rnd(P, 3), rnd(P, 4)])
        torch.cuda.synchronize()
    finally:
        DF.invalidate_birth_order_cache()
    assert bool(torch.isnan(bases.grad).all())
    assert bool(torch.isfinite(leaves[4].grad).all())            # dL/dcoeff does not go through the reduction


def test_deformation_field_matches_reference_golden_on_gpu():
    """End to end against the imported-reference golden (MLP in torch on the GPU + HIP per-Gaussian op)."""
    from rodygs_amd.deform import MLPBasisNetwork, gaussian_deformation
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "deform_golden.npz"))
    net = MLPBasisNetwork(128, 16, 26, False).to(DEV)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    coeff = torch.from_numpy(g["coeff"]).to(DEV).requires_grad_(True)
    emb = net.t_embedder(torch.tensor(float(g["t_now"]), device=DEV))
    rel_ok(emb, g["emb_now"], tol=1e-5, what="time embedding on device (full-range sin/cos)")
    basis_t = net.motion_basis(torch.from_numpy(g["emb_now"]).to(DEV).reshape(1, -1)).squeeze(0)
    table = net.batch_inference(torch.from_numpy(g["embs"]).to(DEV))
    tr, ro = gaussian_deformation(coeff, torch.from_numpy(g["time_ind"]).to(DEV), basis_t, table, float(g["spatial"]))
    rel_ok(tr, g["trans"], tol=1e-4, what="translation"); rel_ok(ro, g["rot"], tol=1e-4, what="rotation")
    ((tr * torch.from_numpy(g["wx"]).to(DEV)).sum() + (ro * torch.from_numpy(g["wr"]).to(DEV)).sum()).backward()
    rel_ok(coeff.grad, g["d_coeff"], tol=1e-4, what="d_coeff")
    gmax = max(np.abs(g[k]).max() for k in g.files if k.startswith("dsd."))
    stacked = {"head_w1": ("0", "weight"), "head_b1": ("0", "bias"), "head_w2": ("2", "weight"), "head_b2": ("2", "bias")}
    for n, p in net.named_parameters():
        if n in stacked:   # our heads are stacked parameters; the reference golden is keyed per head
            layer, kind = stacked[n]
            ref = np.stack([g[f"dsd.basis_xyz.{b}.basis.{layer}.{kind}"] for b in range(16)])
        else:
            ref = g["dsd." + n]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 2e-6 * gmax, (n, err)
    # reference forward(): coeff @ basis(t)
    t2, r2 = net(torch.from_numpy(g["coeff"]).to(DEV), torch.tensor(float(g["t_now"]), device=DEV))
    rel_ok(t2, g["trans_fwd"], tol=1e-4, what="forward translation"); rel_ok(r2, g["rot_fwd"], tol=1e-4, what="fwd rot")


@pytest.mark.parametrize("P", [4, 5, 1000, 20000, 120000])
def test_dist_cuda2(P):
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(P)
    pts = torch.rand(P, 3, generator=g) * torch.tensor([3.0, 1.0, 0.2])
    if P >= 1000:
        pts[:50] = pts[50:100]          # exact duplicates -> zero distances
    out = distCUDA2(pts.to(DEV))
    rel_ok(out, KO.dist2_knn3(pts), tol=1e-5, what="dist2")


@pytest.mark.parametrize("Pq,Pt,K,same", [(5000, 5000, 8, True), (777, 777, 3, True), (300, 300, 16, True),
                                          (1000, 4000, 8, False), (2500, 257, 5, False), (9, 9, 9, True),
                                          (40000, 40000, 8, True)])
def test_knn_points_matches_brute_force(Pq, Pt, K, same):
    """pytorch3d.ops.knn_points replacement vs the brute-force restatement: indices exact, squared distances and
    the gradient of the distances w.r.t. both point sets."""
    from rodygs_amd.knn import knn_points
    g = torch.Generator().manual_seed(Pq + K)
    p2 = torch.randn(Pt, 3, generator=g) * torch.tensor([3.0, 1.0, 0.3])
    p1 = p2 if same else torch.randn(Pq, 3, generator=g) * 2.0
    w = torch.rand(Pq, K, generator=g)
    o1 = p1.clone().double().requires_grad_(True)
    o2 = o1 if same else p2.clone().double().requires_grad_(True)
    d2 = ((o1[:, None, :] - o2[None, :, :]) ** 2).sum(-1) if Pq * Pt <= 25_000_000 else None
    if d2 is not None:
        od, oi = torch.topk(d2, K, dim=1, largest=False, sorted=True)
    else:                       # large case: k-d tree for the indices, distances recomputed differentiably
        from scipy.spatial import cKDTree
        _, oi_np = cKDTree(p2.double().numpy()).query(p1.double().numpy(), k=K)
        oi = torch.from_numpy(oi_np.reshape(Pq, K))
        od = ((o1[:, None, :] - o2[oi]) ** 2).sum(-1)
    (od * w.double()).sum().backward()
    h1 = p1.clone().to(DEV).requires_grad_(True)
    h2 = h1 if same else p2.clone().to(DEV).requires_grad_(True)
    res = knn_points(h1[None], h2[None], K=K)
    assert res.dists.shape == (1, Pq, K) and res.idx.shape == (1, Pq, K) and res.idx.dtype == torch.int64
    (res.dists[0] * w.to(DEV)).sum().backward()
    assert torch.equal(res.idx[0].cpu(), oi), "neighbour indices"
    rel_ok(res.dists[0], od.float(), tol=1e-5, what="dists")
    rel_ok(h1.grad, o1.grad.float(), tol=1e-5, what="d_p1")
    if not same:
        rel_ok(h2.grad, o2.grad.float(), tol=1e-5, what="d_p2")


@pytest.mark.parametrize("K", [3, 8, 16])
def test_knn_points_on_a_cloud_with_a_dense_core_outliers_and_duplicates(K):
    """The neighbour search on the kind of cloud its 48-bit curve codes are for: most points in a core a thousandth of the
    bounding box wide, a few far outliers that set the box, and blocks of exact duplicates (equal distances: any of the tied
    indices is a correct answer, so the DISTANCES are compared with a float32 brute force, to rounding: 1e-6 relative -- and
    every returned index must be at its returned distance and appear once per query)."""
    from rodygs_amd.knn import knn_points
    g = torch.Generator().manual_seed(77 + K)
    n = 20000
    p = torch.randn(n, 3, generator=g) * (torch.rand(n, 1, generator=g) ** 3) * 4.0
    p[:40] = torch.randn(40, 3, generator=g) * 500.0            # outliers: the bounding box is 1000x the core
    p[1000:1300] = p[2000:2300]                                   # exact duplicates
    p[5000:5064] = p[5000]                                        # 64 copies of one point: more ties than K
    h = p.to(DEV)
    res = knn_points(h[None], h[None], K=K)
    d_h, i_h = res.dists[0], res.idx[0]
    ref = torch.empty(n, K, device=DEV)
    for a in range(0, n, 2000):
        diff = h[a:a + 2000, None, :] - h[None, :, :]
        ref[a:a + 2000] = torch.topk((diff * diff).sum(-1), K, dim=1, largest=False, sorted=True)[0]
    assert bool(((d_h - ref).abs() <= 1e-6 * ref).all()), f"distances differ on {int(((d_h - ref).abs() > 1e-6 * ref).any(1).sum())} queries"
    assert bool((d_h[:, 1:] >= d_h[:, :-1]).all()), "distances not ascending"
    dd = h[:, None, :] - h[i_h]
    at = (dd * dd).sum(-1)
    assert bool(((at - d_h).abs() <= 1e-6 * d_h).all()), "an index is not at its returned distance"
    srt = torch.sort(i_h, dim=1)[0]
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "a neighbour is listed twice"
    assert float(d_h[:, 0].max()) == 0.0                          # every query finds itself (or a copy of itself) first


def test_knn_gather_forward_backward():
    from rodygs_amd.knn import knn_gather
    g = torch.Generator().manual_seed(5)
    M, L, K, U = 1234, 777, 8, 19
    x = torch.randn(1, M, U, generator=g)
    idx = torch.randint(0, M, (1, L, K), generator=g)
    w = torch.randn(1, L, K, U, generator=g)
    xo = x.clone().requires_grad_(True)
    oo = KO.knn_gather(xo, idx)
    (oo * w).sum().backward()
    xh = x.clone().to(DEV).requires_grad_(True)
    oh = knn_gather(xh, idx.to(DEV))
    (oh * w.to(DEV)).sum().backward()
    assert torch.equal(oh.cpu(), oo.detach())
    rel_ok(xh.grad, xo.grad, tol=1e-6, what="d_x")


@pytest.mark.parametrize("name", ["coeff", "coeff_l1_nocolor", "all"])
def test_rigidity_loss_on_hip_knn_matches_reference_golden(name):
    """RigidityLoss on the GPU through the HIP knn_points / knn_gather against (a) what the imported reference
    returned on the CPU (tests/golden/rigidity_golden.npz; same seeds, same random subsets) and (b) the same module
    on the same device with the brute-force restatement of the two ops.

    (b) isolates the ops: <= 1e-5.  (a) also contains torch-CPU vs torch-GPU arithmetic of the loss itself; its
    distance_preserving term is Charbonnier with eps = 1e-6 evaluated at the self neighbour, where the two compared
    quantities are equal up to rounding -- d/dx = (x-y)/sqrt((x-y)^2 + 1e-12) turns a 1e-7 rounding difference into
    an O(0.1) slope, so gradients of mode "all" agree with the CPU golden to 5e-3 only (measured 2e-3, identical for
    the HIP ops and for torch's own ops on the GPU); value and the "coeff" modes hold the 1e-4 bar."""
    from test_oracle_golden import run_rigidity_case
    loss, grads, gold = run_rigidity_case(name, DEV)
    loss_o, grads_o, _ = run_rigidity_case(name, DEV, KO.knn_points_batched, KO.knn_gather)
    want = float(gold[name + ".loss"])
    assert abs(float(loss) - want) <= 2e-5 * abs(want)
    assert abs(float(loss) - float(loss_o)) <= 2e-6 * abs(want)
    for k, gr in grads.items():
        w_ = torch.from_numpy(gold[f"{name}.d_{k}"])
        if gr is None:
            assert float(w_.abs().sum()) == 0.0
            continue
        rel_ok(gr, grads_o[k], tol=1e-5, what=f"rigidity {name} d_{k} vs same-device brute force")
        rel_ok(gr, w_, tol=5e-3 if name == "all" else 1e-4, what=f"rigidity {name} d_{k} vs reference golden")


_DP, _SURF, _COEFF = "distance_preserving", "surface", "coeff"


@pytest.mark.parametrize("Tu,K,mode", [(16, 8, [_DP]), (100, 8, [_DP]), (140, 5, [_DP]), (300, 8, [_DP]), (100, 8, [_SURF]),
                                       (100, 5, [_DP, _SURF]), (100, 8, [_COEFF, _SURF, _DP])])
def test_fused_rigidity_terms_equal_the_unfused_module_on_the_same_device(Tu, K, mode):
    """The fused HIP path of RigidityLoss (neighbour-search backward, surface term and distance-preserving term through the
    step's stored-order graph: no float atomics, no [t,n,K,3] tensors) vs the module's own tensor-by-tensor expression (the
    reference's formulation, on the public knn_points / knn_gather ops) on the same device and the same random draws: value
    and all gradients.  Drawn times 4 / 25 / 35 / 75: every lane-group width of the kernels, fewer times than neighbours
    (d2 gradient through atomics), more times than a wave has lanes; K = 5: the run-time neighbour count."""
    import random
    from rodygs_amd.rigidity import RigidityLoss
    from test_oracle_golden import _FakeDynModel
    g = torch.Generator().manual_seed(12)
    P, B = 3000, 16
    base = dict(xyz=torch.rand(P, 3, generator=g) * 4 - 2, transl=0.05 * torch.randn(P, 3, generator=g),
                coeff=0.3 * torch.randn(P, 1, B, generator=g), fdc=torch.rand(P, 1, 3, generator=g),
                table=0.2 * torch.randn(Tu, B, 7, generator=g))
    names = ["xyz", "transl"] + (["coeff"] if _DP in mode or _COEFF in mode else []) + (["table"] if _DP in mode else []) \
        + (["fdc"] if _COEFF in mode else [])
    outs = []
    for fused in (True, False):
        t = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
        random.seed(5)
        torch.manual_seed(6)
        mod = RigidityLoss(mode=mode, K=K)
        mod.fused_dp = fused
        loss = mod(_FakeDynModel(t["xyz"], t["coeff"], t["fdc"], t["table"]), t["transl"])
        grads = torch.autograd.grad(loss, [t[k] for k in names])
        outs.append((loss, grads))
    (lf, gf), (lu, gu) = outs
    assert abs(float(lf) - float(lu)) <= 2e-6 * abs(float(lu))
    for a, b, nm in zip(gf, gu, names):
        rel_ok(a, b, tol=2e-5, what=f"fused rigidity {mode} d_" + nm)


def test_graph_backward_of_the_neighbour_search_equals_the_atomic_one():
    """rdg_graph_points_backward (every gradient row written once through the stored-order reverse adjacency) vs
    rdg_knn_points_backward (float atomics) on the same upstream gradient, K = 8 and K = 3."""
    from rodygs_amd import knn as KN
    from rodygs_amd.rigidity import _GraphSlot, _KnnPointsGraph, _NeighbourGraph
    g = torch.Generator().manual_seed(3)
    for n, K in ((5000, 8), (777, 3)):
        pts = (torch.rand(n, 3, generator=g) * 3).to(DEV)
        up = torch.randn(n, K, generator=g).to(DEV)
        a = pts.clone().requires_grad_(True)
        res = KN.knn_points(a[None], a[None], K=K)
        (ga,) = torch.autograd.grad((res.dists[0] * up).sum(), a)
        b = pts.clone().requires_grad_(True)
        slot = _GraphSlot()
        d2, idx = _KnnPointsGraph.apply(b, K, slot)
        assert torch.equal(idx, res.idx[0]) and torch.equal(d2, res.dists[0])
        slot.graph = _NeighbourGraph(b, idx)
        (gb,) = torch.autograd.grad((d2 * up).sum(), b)
        rel_ok(gb, ga, tol=1e-5, what=f"graph knn backward n={n} K={K}")


def test_pytorch3d_shim_resolves_to_hip_ops():
    import pytorch3d.ops as torch3d
    from rodygs_amd import knn
    assert torch3d.knn_points is knn.knn_points and torch3d.knn_gather is knn.knn_gather


@pytest.mark.parametrize("mode", [None, "static", "dynamic"])
def test_pearson_depth_losses_match_reference_golden(mode):
    """Fused HIP Global / Local Pearson depth loss vs what the imported reference returned (golden G8): value and
    the gradient of the predicted depth, with the reference's own box corners and motion mask."""
    from rodygs_amd.depth_losses import GlobalPearsonDepthLoss, LocalPearsonDepthLoss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "depth_loss_golden.npz"))
    tag = str(mode)
    gt, motion = torch.from_numpy(g["gt"]).to(DEV), torch.from_numpy(g["motion"]).to(DEV)
    mm = None if mode is None else motion
    pred = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    lg = GlobalPearsonDepthLoss(mode)(pred, gt, mm)
    (3.0 * lg).backward()
    assert abs(float(lg) - float(g[f"global.{tag}.loss"])) <= 2e-6
    rel_ok(pred.grad / 3.0, g[f"global.{tag}.d_pred"], tol=1e-4, what="global d_pred")
    pred2 = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    rows, cols = torch.from_numpy(g[f"local.{tag}.rows"]).to(DEV), torch.from_numpy(g[f"local.{tag}.cols"]).to(DEV)
    ll = LocalPearsonDepthLoss(int(g["box_p"]), float(g["p_corr"]), mode)(pred2, gt, mm, boxes=(rows, cols))
    ll.backward()
    assert abs(float(ll) - float(g[f"local.{tag}.loss"])) <= 2e-6
    rel_ok(pred2.grad, g[f"local.{tag}.d_pred"], tol=1e-4, what="local d_pred")


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840)])     # BASELINE configs[2..3] and configs[4] image sizes
def test_pearson_depth_loss_full_hd_against_oracle(H, W):
    """1080p and 4K, box_p 128, p_corr 0.5 (configs/train/train_kubric_mrig.yaml depth losses): 60 / 240 boxes drawn on
    the GPU exactly as the reference draws them; value and gradient against the CPU oracle on the same corners."""
    import math as _m
    from oracle import depth_loss_oracle as DL
    from rodygs_amd.depth_losses import GlobalPearsonDepthLoss, LocalPearsonDepthLoss
    g = torch.Generator().manual_seed(8)
    n_box = int(0.5 * _m.floor(H / 128) * _m.floor(W / 128))
    gt = torch.rand(1, H, W, generator=g) * 15 + 2
    p0 = gt * 1.3 - 1.0 + torch.randn(1, H, W, generator=g)
    torch.manual_seed(5)
    rows = torch.randint(0, H - 128, size=(n_box,), device=DEV)
    cols = torch.randint(0, W - 128, size=(n_box,), device=DEV)
    torch.manual_seed(5)
    pred = p0.clone().to(DEV).requires_grad_(True)
    loss = LocalPearsonDepthLoss(128, 0.5)(pred, gt.to(DEV)) + GlobalPearsonDepthLoss()(pred, gt.to(DEV))
    loss.backward()
    po = p0.clone().requires_grad_(True)
    lo = DL.local_pearson_depth_loss(po, gt, rows.cpu(), cols.cpu(), 128, n_box) + DL.pearson_depth_loss(po, gt)
    lo.backward()
    assert abs(float(loss) - float(lo)) <= 1e-5 * abs(float(lo))
    rel_ok(pred.grad, po.grad, tol=1e-4, what=f"d_pred {W}x{H}")


def test_config5_shape_full_loss_step_at_4m_4k():
    """BASELINE configs[4] shape on one GPU: 4 M dynamic Gaussians, 3840x2160, the whole loss set of the reference's
    dynamic sub-step (photometric + global / local Pearson depth + motion L1 / sparsity / basis regularisers + rigidity
    on the HIP K-NN every 5th step).  Too big for the oracle as a whole (its pieces are compared at this size above and
    in the sampled-tile test): here the step must run on the rigidity path and on the plain path, keep every parameter
    finite, move the loss down, and the per-Gaussian motion regularisers must equal their torch expression at 4 M rows."""
    from rodygs_amd.motion_losses import fused_motion_l1_sparsity
    from rodygs_amd.trainstep import DynamicScene
    P, W, H = 4000000, 3840, 2160
    sc = O.synthetic_scene(P, W, H, 3, seed=777)
    tgt = O.synthetic_scene(P // 4, W, H, 3, seed=1234)
    ds = DynamicScene(sc, num_frames=100, device=DEV, full_losses=True, spatial_order=True)
    frames = [0, 25, 50, 75]
    ds.make_ground_truth(tgt, frames)
    del tgt
    losses = [float(ds.train_step(s_, perm=frames)) for s_ in range(7)]       # steps 0 and 5: rigidity path
    assert all(np.isfinite(losses)), losses
    assert losses[6] < losses[1] and losses[5] < losses[0], losses
    for k in ds.fp.names:
        assert bool(torch.isfinite(ds.fp[k]).all()), k
    assert float(ds.fp["motion_coeff"].grad.abs().sum()) > 0 and float(ds.sp["cam_t"].grad.abs().sum()) > 0
    # motion L1 + sparsity at 4 M rows against the reference's torch expression (losses.py:363-420)
    c = ds.fp["motion_coeff"].detach().clone().requires_grad_(True)
    sink = torch.zeros_like(c)
    lf = fused_motion_l1_sparsity(c, 0.01, 0.002, grad_sink=sink)
    lf.backward()
    from rodygs_amd.motion_losses import MotionL1Loss, MotionSparsityLoss      # host mirrors pinned by golden G9
    cr = c.detach().clone().requires_grad_(True)

    class _M:
        _motion_coeff = cr

    lref = 0.01 * MotionL1Loss()(_M) + 0.002 * MotionSparsityLoss()(_M)
    lref.backward()
    assert abs(float(lf) - float(lref)) <= 2e-5 * abs(float(lref)), (float(lf), float(lref))
    rel_ok(sink, cr.grad, tol=1e-4, what="motion regulariser gradient at 4 M")


@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_densify_and_prune_matches_oracle(max_screen_size):
    """One-gather-per-buffer densify/prune over the flat buckets vs the tensor-by-tensor restatement of the reference
    trainer: same rows in the same order, Adam moments carried / zeroed alike, split children placed alike."""
    from oracle import densify_oracle as DZ
    from rodygs_amd.densify import DensifyStats, densify_and_prune
    from rodygs_amd.dp import FlatParams
    g = torch.Generator().manual_seed(77)
    P, K, B, N = 5000, 16, 16, 2
    extent, percent_dense, max_grad, min_opacity = 5.0, 0.01, 0.0002, 0.05
    spec = {"xyz": ((P, 3), 1e-3), "features": ((P, K, 3), 1e-3), "scaling": ((P, 3), 1e-3), "rotation": ((P, 4), 1e-3),
            "opacity": ((P, 1), 1e-3), "motion_coeff": ((P, 1, B), 1e-3)}
    vals = {"xyz": torch.randn(P, 3, generator=g), "features": torch.randn(P, K, 3, generator=g),
            # half of the Gaussians below the percent_dense * extent = 0.05 size threshold, a few huge ones
            "scaling": torch.log(torch.rand(P, 3, generator=g) * 0.08 + 0.005 + (torch.rand(P, 1, generator=g) > 0.97) * 1.0),
            "rotation": torch.randn(P, 4, generator=g), "opacity": torch.randn(P, 1, generator=g) * 2.5,
            "motion_coeff": torch.randn(P, 1, B, generator=g)}
    fp = FlatParams(spec, DEV)
    m1 = {k: torch.randn(v.shape, generator=g) for k, v in vals.items()}
    m2 = {k: torch.rand(v.shape, generator=g) for k, v in vals.items()}
    with torch.no_grad():
        for k in fp.names:
            o, n = fp.offsets[k]
            fp[k].copy_(vals[k])
            fp.exp_avg[o:o + n].copy_(m1[k].reshape(-1))
            fp.exp_avg_sq[o:o + n].copy_(m2[k].reshape(-1))
    fp.step_count = 41
    denom = torch.randint(0, 4, (P, 1), generator=g).float()                       # zeros -> NaN average -> 0
    accum = torch.rand(P, 1, generator=g) * 0.0006 * denom
    radii = torch.rand(P, generator=g) * 40
    t_ind = torch.randint(0, 30, (P,), generator=g)
    t_val = t_ind.float() / 30
    z = torch.randn(2 * P, 3, generator=g)
    stats = DensifyStats(accum.clone().to(DEV), denom.clone().to(DEV), radii.clone().to(DEV))
    res = densify_and_prune(fp, stats, {"gaussian_to_time": t_val.to(DEV), "gaussian_to_time_ind": t_ind.to(DEV)},
                            max_grad, min_opacity, extent, max_screen_size, percent_dense, N, z=z.to(DEV))
    st = DZ.State({k: v.clone() for k, v in vals.items()}, {k: v.clone() for k, v in m1.items()},
                  {k: v.clone() for k, v in m2.items()}, accum.clone(), denom.clone(), radii.clone(),
                  {"gaussian_to_time": t_val.clone(), "gaussian_to_time_ind": t_ind.clone()})
    n_clone, n_sel = DZ.densify_and_prune(st, max_grad, min_opacity, extent, max_screen_size, percent_dense, N, z)
    assert n_clone > 100 and n_sel > 100 and (res.n_clone, res.n_split) == (n_clone, n_sel)
    Pn = st.P
    assert Pn != P and res.fp.shapes["xyz"][0] == Pn and res.fp.step_count == 41
    for k in fp.names:
        o, n = res.fp.offsets[k]
        got = res.fp[k].detach().cpu()
        if k in ("xyz", "scaling"):
            rel_ok(got, st.params[k], tol=2e-6, what="densify " + k)
        else:
            assert torch.equal(got, st.params[k]), k
        assert torch.equal(res.fp.exp_avg[o:o + n].cpu().view_as(st.exp_avg[k]), st.exp_avg[k]), k
        assert torch.equal(res.fp.exp_avg_sq[o:o + n].cpu().view_as(st.exp_avg_sq[k]), st.exp_avg_sq[k]), k
    assert float(res.fp.flat_grad.abs().sum()) == 0.0
    for k, v in st.per_point.items():
        assert torch.equal(res.per_point[k].cpu(), v), k
    assert torch.equal(res.stats.xyz_gradient_accum.cpu(), st.accum) and torch.equal(res.stats.max_radii2D.cpu(), st.max_radii)
    assert res.n_pruned == P + n_clone + N * n_sel - Pn


def test_fused_adam_matches_torch():
    from rodygs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-2, eps=1e-15)
    pd, m, v = p.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        pt.grad = gr * step
        opt.step()
        gd = (gr * step).to(DEV)
        _lib.check(L.rdg_adam_step(n, pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-2, 0.9, 0.999,
                                   1e-15, step, _lib.stream_ptr()), "adam")
    rel_ok(pd, pt.detach(), tol=1e-6, what="adam")


@pytest.mark.parametrize("C_,H,W", [(3, 48, 64), (3, 211, 333), (1, 16, 16), (3, 1080, 1920)])
def test_fused_photometric_loss(C_, H, W):
    """HIP fused L1 + D-SSIM vs the torch restatement (itself pinned to the reference's golden in the CPU suite)."""
    from rodygs_amd.losses import fused_photometric_loss, photometric_loss
    g = torch.Generator().manual_seed(H * W)
    a = torch.rand(C_, H, W, generator=g)
    b = (a + 0.1 * torch.randn(C_, H, W, generator=g)).clamp(0, 1)
    a1 = a.clone().to(DEV).requires_grad_(True)
    lf = fused_photometric_loss(a1, b.to(DEV), 0.2)
    (lf * 1.7).backward()
    a2 = a.clone().requires_grad_(True)
    lt = photometric_loss(a2, b, 0.2)
    (lt * 1.7).backward()
    assert abs(lf.item() - lt.item()) <= 2e-5 * abs(lt.item())
    rel_ok(a1.grad, a2.grad, tol=1e-4, what="d_image")
    if (C_, H, W) == (3, 48, 64):
        gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
        lg = fused_photometric_loss(torch.from_numpy(gold["a"]).to(DEV), torch.from_numpy(gold["b"]).to(DEV), 0.2)
        want = 0.8 * float(gold["l1"]) + 0.2 * (1.0 - float(gold["ssim"]))
        assert abs(lg.item() - want) <= 2e-5 * abs(want)


def test_train_step_runs_and_reduces_loss():
    """The minimal dynamic train step bench.py times (deform -> raster -> fused loss -> backward -> fused Adam)."""
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)
    ds = DynamicScene(sc, num_frames=8, device=DEV)
    ds.make_ground_truth(tgt, range(8))
    losses = [float(ds.train_step(s, perm=list(range(8)))) for s in range(40)]
    assert all(np.isfinite(losses))
    assert np.mean(losses[-8:]) < np.mean(losses[:8])
    assert float(ds.cam_q.grad.abs().sum()) > 0 and ds.net.head_w1.grad is not None


def test_train_loop_with_densification():
    """Build-plan item 8: the train step with densify-and-prune in the loop -- statistics gathered from the
    rasterizer's means2D gradient and radii, the flat bucket rebuilt (Adam moments carried for survivors), sinks and
    birth indices re-pointed, and the optimisation carries on."""
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)
    ds = DynamicScene(sc, num_frames=8, device=DEV)
    ds.make_ground_truth(tgt, range(8))
    ds.track_densification()
    losses = [float(ds.train_step(s_, perm=list(range(8)))) for s_ in range(24)]
    assert float(ds.stats.denom.sum()) > 0 and float(ds.stats.max_radii2D.max()) > 0
    m_before = ds.fp.exp_avg_sq.abs().sum().item()
    info = ds.densify(max_grad=2e-5, min_opacity=0.05, percent_dense=0.002)
    assert info["cloned"] > 0 and info["split"] > 0 and info["pruned"] > 0 and info["P"] == ds.fp.shapes["xyz"][0] != 20000
    assert ds.time_ind.shape[0] == info["P"] and 0 < ds.fp.exp_avg_sq.abs().sum().item() <= m_before
    assert float(ds.stats.denom.sum()) == 0.0
    losses += [float(ds.train_step(s_, perm=list(range(8)))) for s_ in range(24, 48)]
    assert all(np.isfinite(losses)) and np.mean(losses[-8:]) < np.mean(losses[:8])
    assert float(ds.fp["xyz"].grad.abs().sum()) > 0 and ds.m2.grad.shape[0] == info["P"]


def test_deterministic_train_step_is_bit_reproducible():
    """RDG_DETERMINISTIC end to end: the photometric train step bench.py times (MLP -> deformation -> rasterizer -> loss
    -> backward -> Adam, SH Adam inside backward, one densification) run twice from the same state ends with the SAME
    BITS in every parameter and both Adam moments -- no float atomic is left on that path (compositing backward: per-
    instance partial rows + ordered reduction; dB reduction: per-(wave, birth index) slots + ordered sum)."""
    import rodygs_amd.rasterizer as R
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)

    def run(det):
        old = R.DETERMINISTIC
        R.DETERMINISTIC = det
        try:
            torch.manual_seed(1234)                       # split samples of the densification
            ds = DynamicScene(sc, num_frames=8, device=DEV, spatial_order=True)
            ds.make_ground_truth(tgt, range(8))
            ds.track_densification()
            for s_ in range(20):
                ds.train_step(s_, perm=list(range(8)))
            ds.densify(max_grad=2e-5, min_opacity=0.05, percent_dense=0.002)
            for s_ in range(20, 40):
                ds.train_step(s_, perm=list(range(8)))
            torch.cuda.synchronize()
        finally:
            R.DETERMINISTIC = old
        return [t.clone() for t in (ds.fp.flat, ds.fp.exp_avg, ds.fp.exp_avg_sq, ds.sp.flat, ds.sp.exp_avg,
                                    ds.sp.exp_avg_sq)]

    a, b = run(True), run(True)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and torch.equal(x, y), f"deterministic train step: buffer {i} differs between two runs"
    c = run(False)     # the atomic path takes the same decisions at the densification (rounding noise only up to there)
    assert bool(torch.isfinite(c[0]).all())


def test_graph_replay_of_the_train_step_is_bit_identical_to_the_eager_step():
    """trainstep.GraphedStep: the photometric train step captured ONCE as a hipGraph and replayed with the per-step
    values (frame's embedding rows, ground truth, Adam bias corrections, frame index) refreshed in device memory --
    against the same steps launched kernel by kernel from Python.  Deterministic backward on both sides, so every
    parameter, both Adam moments (Gaussians, MLP, camera poses) and the losses must agree BIT FOR BIT."""
    import rodygs_amd.rasterizer as R
    from rodygs_amd.trainstep import DynamicScene, GraphedStep
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)
    old = R.DETERMINISTIC
    R.DETERMINISTIC = True
    try:
        def fresh():
            ds = DynamicScene(sc, num_frames=8, device=DEV, spatial_order=True)
            ds.make_ground_truth(tgt, range(8))
            return ds
        perm, n = [0, 3, 5, 6, 1], 23
        a = fresh()
        la = [a.train_step(s_, perm=perm) for s_ in range(n)]
        b = fresh()
        gs = GraphedStep(b, perm, warmup=2)                 # two eager steps, then steps 2 .. n-1 as replays
        lb = [gs.step().clone() for _ in range(n - 2)]
        assert gs.check() > 0 and gs.next_step == n and b.fp.step_count == a.fp.step_count == n
        gs.close()
        torch.cuda.synchronize()
    finally:
        R.DETERMINISTIC = old
    for x, y in zip(la[2:], lb):
        assert torch.equal(x, y)
    for name in ("flat", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(a.fp, name), getattr(b.fp, name)), "Gaussian bucket " + name
        assert torch.equal(getattr(a.sp, name), getattr(b.sp, name)), "MLP + pose bucket " + name
    # and the scene keeps training eagerly after the graph is dropped
    assert torch.isfinite(b.train_step(n, perm=perm))


def test_train_step_and_graph_replay_with_nan_filled_workspaces():
    """Every ``torch.empty`` filled with NaN / 0xFF (torch.utils.deterministic.fill_uninitialized_memory): a kernel that
    reads a workspace before writing it, or a zero-fill that does not happen where the stream order says, turns the step
    into NaN.  (How the memset nodes of a captured hipGraph were caught racing the kernels next to them: the library
    zero-fills with its own kernel since.)  Eager photometric and full-loss steps stay finite; the float-atomic graph
    replay follows the eager twin step by step."""
    import torch.utils.deterministic as TD
    from rodygs_amd.trainstep import DynamicScene, GraphedStep
    prev = (torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled(),
            TD.fill_uninitialized_memory)
    torch.use_deterministic_algorithms(True, warn_only=True)
    TD.fill_uninitialized_memory = True
    try:
        sc = O.synthetic_scene(30000, 640, 360, 3, seed=5)
        tgt = O.synthetic_scene(8000, 640, 360, 3, seed=6)
        perm = [0, 3, 5, 6, 1, 7]

        def fresh(**kw):
            ds = DynamicScene(sc, num_frames=8, device=DEV, spatial_order=True, **kw)
            ds.make_ground_truth(tgt, range(8))
            return ds

        def finite(ds):
            torch.cuda.synchronize()
            return all(bool(torch.isfinite(f[k]).all()) and bool(torch.isfinite(f[k].grad).all())
                       for f in (ds.fp, ds.sp) for k in f.names)

        a = fresh()
        la = [float(a.train_step(s_, perm=perm)) for s_ in range(14)]
        assert finite(a) and all(np.isfinite(la))
        b = fresh()
        gs = GraphedStep(b, perm, warmup=2)
        lb = [float(gs.step()) for _ in range(12)]
        assert finite(b) and gs.check() > 0
        gs.close()
        assert np.allclose(lb, la[2:], rtol=2e-3, atol=0), (la[2:], lb)       # float atomics: same curve, not the same bits
        c = fresh(full_losses=True)
        lc = [float(c.train_step(s_, perm=perm)) for s_ in range(6)]          # step 0 and 5 are rigidity steps
        assert finite(c) and all(np.isfinite(lc))
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
        TD.fill_uninitialized_memory = prev[2]


def test_full_loss_train_step_runs_and_reduces_loss():
    """Config-5 loss set in the loop: photometric + Pearson depth (global + local) + motion regularisers + rigidity on
    the HIP K-NN every 5th step; gradients of several losses accumulate into the same flat segments."""
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(20000, 320, 256, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 256, 3, seed=6)
    ds = DynamicScene(sc, num_frames=8, device=DEV, full_losses=True)
    ds.make_ground_truth(tgt, range(8))
    losses = [float(ds.train_step(s_, perm=list(range(8)))) for s_ in range(30)]
    assert all(np.isfinite(losses))
    rig = [l for i, l in enumerate(losses) if i % 5 == 0]
    plain = [l for i, l in enumerate(losses) if i % 5 != 0]
    assert np.mean(plain[-8:]) < np.mean(plain[:8]) and rig[-1] < rig[0]
    assert float(ds.fp["motion_coeff"].grad.abs().sum()) > 0 and float(ds.sp["cam_t"].grad.abs().sum()) > 0


def test_overlapped_exchange_path_equals_single_launch_path():
    """The frame-DP branch of the train step (bucketed asynchronous all-reduce over RCCL + Adam applied piece by piece)
    on a 1-rank process group must reproduce the single-launch optimiser step bit for bit from the same gradients:
    the sum over one rank is the identity, so any difference would come from the bucketing / per-piece logic.
    (Whole steps are not compared bitwise: the rasterizer backward accumulates with float atomics.)"""
    import socket
    import torch.distributed as dist
    from rodygs_amd.dp import BucketedAllReduce
    from rodygs_amd.losses import fused_photometric_loss
    from rodygs_amd.trainstep import DynamicScene, fused_adam_
    sc = O.synthetic_scene(6000, 160, 120, 3, seed=15)
    tgt = O.synthetic_scene(1500, 160, 120, 3, seed=16)
    ds = DynamicScene(sc, num_frames=4, device=DEV)
    ds.make_ground_truth(tgt, range(4))
    ds.train_step(0, perm=[2])                                  # non-trivial Adam moments
    out, _ = ds.render(2)
    fused_photometric_loss(out[0], ds.gt[2], 0.2).backward()    # fresh gradients in both flat buckets
    bufs = [ds.fp.flat, ds.fp.exp_avg, ds.fp.exp_avg_sq, ds.sp.flat, ds.sp.exp_avg, ds.sp.exp_avg_sq]
    grads0 = (ds.fp.flat_grad.clone(), ds.sp.flat_grad.clone())
    snap = [b.clone() for b in bufs]
    step0 = ds.fp.step_count
    fused_adam_(ds.fp, row_lr=ds.row_lr, extra=(ds.sp,))
    want = [b.clone() for b in bufs]
    for b, s0 in zip(bufs, snap):
        b.copy_(s0)
    ds.fp.step_count = step0

    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV, torch.cuda.current_device()))
    real = BucketedAllReduce.active
    BucketedAllReduce.active = staticmethod(lambda: True)
    try:
        ds.sync.ready("features")
        ds.sync.finish()
        pieces, first = [], True
        for names in ds.sync.drain():
            pieces.append(names)
            if names is None:
                fused_adam_(ds.fp, names=(), extra=(ds.sp,), advance=first)
            else:
                fused_adam_(ds.fp, row_lr=ds.row_lr, names=names, advance=first)
            first = False
        torch.cuda.synchronize()
        for b, w in zip(bufs, want):
            assert torch.equal(b, w)
        assert torch.equal(ds.fp.flat_grad, grads0[0]) and torch.equal(ds.sp.flat_grad, grads0[1])
        assert ds.fp.step_count == step0 + 1 and ds.sp.step_count == step0 + 1
        # and the whole frame-DP step as bench.py drives it at N > 1: the SH piece is issued from inside backward
        # (autograd thread), the rest after it
        step1 = ds.fp.step_count
        before = ds.fp.flat.clone()
        dp_losses = [float(ds.train_step(s_, 0, 2, perm=[1, 2])) for s_ in range(3)]
        torch.cuda.synchronize()
    finally:
        BucketedAllReduce.active = staticmethod(real)
        dist.destroy_process_group()
    assert all(np.isfinite(dp_losses)) and ds.fp.step_count == step1 + 3 and ds.sp.step_count == step1 + 3
    assert float((ds.fp.flat - before).abs().max()) > 0
    assert pieces == [["features"], ["xyz"], ["scaling", "rotation", "opacity", "motion_coeff"], None]
    del bufs


@pytest.mark.parametrize("use_sinks", [False, True])
def test_fused_activations_match_reference_getters(use_sinks):
    """activate_gaussians vs the torch formulas of the reference getters (rodygs_static.py:82-105) + deformation add."""
    import torch.nn.functional as F
    from rodygs_amd.model_ops import activate_gaussians
    g = torch.Generator().manual_seed(11)
    P, K = 3001, 16
    raw = dict(xyz=torch.randn(P, 3, generator=g), scaling=torch.randn(P, 3, generator=g) - 2,
               rotation=torch.randn(P, 4, generator=g), opacity=torch.randn(P, 1, generator=g),
               f_dc=torch.randn(P, 1, 3, generator=g), f_rest=torch.randn(P, K - 1, 3, generator=g))
    dxyz, drot = 0.1 * torch.randn(P, 3, generator=g), 0.1 * torch.randn(P, 4, generator=g)
    ws = [torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g),
          torch.randn(P, 1, generator=g), torch.randn(P, K, 3, generator=g)]

    def torch_ref(r, dx, dr):
        return (r["xyz"] + dx, torch.exp(r["scaling"]), F.normalize(r["rotation"]) + dr, torch.sigmoid(r["opacity"]),
                torch.cat((r["f_dc"], r["f_rest"]), dim=1))

    rc = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    dxc, drc = dxyz.clone().requires_grad_(True), drot.clone().requires_grad_(True)
    outs_c = torch_ref(rc, dxc, drc)
    sum((o * w).sum() for o, w in zip(outs_c, ws)).backward()
    rg = {k: v.clone().to(DEV).requires_grad_(True) for k, v in raw.items()}
    dxg, drg = dxyz.clone().to(DEV).requires_grad_(True), drot.clone().to(DEV).requires_grad_(True)
    sinks = {k: torch.full_like(v, 7.0) for k, v in rg.items()} if use_sinks else None
    outs_g = activate_gaussians(rg["xyz"], dxg, rg["scaling"], rg["rotation"], drg, rg["opacity"], rg["f_dc"],
                                rg["f_rest"], grad_sinks=sinks)
    sum((o * w.to(DEV)).sum() for o, w in zip(outs_g, ws)).backward()
    for o_g, o_c, n in zip(outs_g, outs_c, ("means3D", "scales", "rots", "opac", "shs")):
        rel_ok(o_g, o_c, tol=2e-6, what=n)
    for k in raw:
        got = sinks[k] if use_sinks else rg[k].grad
        rel_ok(got, rc[k].grad, tol=5e-6, what="d_" + k)
        if use_sinks:
            assert rg[k].grad is None
    rel_ok(dxg.grad, dxc.grad, tol=1e-6, what="d_dxyz"); rel_ok(drg.grad, drc.grad, tol=1e-6, what="d_drot")


def test_gs_properties_static_plus_dynamic_without_cat():
    """gs_properties (static ‖ dynamic written into one set of buffers) vs the reference's expression: getters,
    deformation add, torch.cat (rodygs.py:68-113), values and gradients of every raw parameter and of the deltas."""
    import torch.nn.functional as F
    from rodygs_amd.model_ops import gs_properties
    g = torch.Generator().manual_seed(23)
    Ps, Pd, K = 1777, 2049, 16
    def cloud(P):
        return dict(xyz=torch.randn(P, 3, generator=g), scaling=torch.randn(P, 3, generator=g) - 2,
                    rotation=torch.randn(P, 4, generator=g), opacity=torch.randn(P, 1, generator=g),
                    f_dc=torch.randn(P, 1, 3, generator=g), f_rest=torch.randn(P, K - 1, 3, generator=g))
    st, dy = cloud(Ps), cloud(Pd)
    dxyz, drot = 0.1 * torch.randn(Pd, 3, generator=g), 0.1 * torch.randn(Pd, 4, generator=g)
    P = Ps + Pd
    ws = [torch.randn(P, 3, generator=g), torch.randn(P, 1, generator=g), torch.randn(P, 3, generator=g),
          torch.randn(P, 4, generator=g), torch.randn(P, K, 3, generator=g)]

    def leaves(dev):
        return ({k: v.clone().to(dev).requires_grad_(True) for k, v in st.items()},
                {k: v.clone().to(dev).requires_grad_(True) for k, v in dy.items()},
                dxyz.clone().to(dev).requires_grad_(True), drot.clone().to(dev).requires_grad_(True))

    s0, d0, x0, r0 = leaves("cpu")
    ref = (torch.cat([s0["xyz"], d0["xyz"] + x0]), torch.cat([torch.sigmoid(s0["opacity"]), torch.sigmoid(d0["opacity"])]),
           torch.cat([torch.exp(s0["scaling"]), torch.exp(d0["scaling"])]),
           torch.cat([F.normalize(s0["rotation"]), F.normalize(d0["rotation"]) + r0]),
           torch.cat([torch.cat([s0["f_dc"], s0["f_rest"]], 1), torch.cat([d0["f_dc"], d0["f_rest"]], 1)]))
    sum((a * w).sum() for a, w in zip(ref, ws)).backward()
    s1, d1, x1, r1 = leaves(DEV)
    out = gs_properties(s1, d1, x1, r1)
    sum((a * w.to(DEV)).sum() for a, w in zip(out, ws)).backward()
    for a, b, nm in zip(out, ref, ("xyz", "opacity", "scaling", "rotation", "features")):
        rel_ok(a, b, tol=2e-6, what="gs_properties " + nm)
    for k in st:
        rel_ok(s1[k].grad, s0[k].grad, tol=2e-6, what="static d_" + k)
        rel_ok(d1[k].grad, d0[k].grad, tol=2e-6, what="dynamic d_" + k)
    rel_ok(x1.grad, x0.grad, tol=1e-6, what="d_translation")
    rel_ok(r1.grad, r0.grad, tol=1e-6, what="d_rotation_delta")


def test_pose_view_matrix_matches_reference_camera():
    """pose_view_matrix vs the torch restatement of FixedCameraTorch.world_view_transform (pinned to the reference by
    tests/golden/camera_golden.npz), forward and gradients, including non-unit quaternions."""
    from rodygs_amd.model_ops import pose_view_matrix
    from rodygs_amd.trainstep import world_view_transform
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera_golden.npz"))
    q = torch.stack([torch.from_numpy(gold[f"q{i}"]) for i in range(4)]).float()
    t = torch.stack([torch.from_numpy(gold[f"t{i}"]) for i in range(4)]).float()
    for f in range(4):
        ref = torch.from_numpy(gold[f"w2c{f}"])
        qc, tc = q.clone().requires_grad_(True), t.clone().requires_grad_(True)
        vc = world_view_transform(qc[f], tc[f])
        rel_ok(vc, ref, tol=1e-6, what="torch restatement vs reference golden")
        w = torch.randn(4, 4, generator=torch.Generator().manual_seed(f))
        (vc.transpose(0, 1) * w).sum().backward()
        qg, tg = q.clone().to(DEV).requires_grad_(True), t.clone().to(DEV).requires_grad_(True)
        vg = pose_view_matrix(qg, tg, f)
        rel_ok(vg, ref.t(), tol=1e-6, what="view (glm storage)")
        (vg * w.to(DEV)).sum().backward()
        rel_ok(qg.grad, qc.grad, tol=1e-5, what="d_quat"); rel_ok(tg.grad, tc.grad, tol=1e-5, what="d_trans")


@pytest.mark.parametrize("NR,width", [(1, 128), (16, 128), (37, 128), (101, 128), (300, 128), (33, 64), (20, 40), (50, 32)])
def test_mfma_mlp_matches_torch(NR, width):
    """rdg_mlp_forward/backward (v_mfma_f32_16x16x4_f32) vs the torch expression of the same network, at several batch sizes and
    widths (whole MFMA tiles and ragged ones)."""
    from rodygs_amd import deform
    g = torch.Generator().manual_seed(NR)
    net = deform.MLPBasisNetwork(width, 16, 26, False).to(DEV)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1)).to(DEV))
    x = torch.randn(NR, 53, generator=g).to(DEV)
    w = torch.randn(NR, 16, 7, generator=g).to(DEV)
    res = {}
    for fused in (True, False):
        deform.FUSED_MLP = fused
        try:
            net.zero_grad(set_to_none=True)
            out = net.motion_basis(x)
            (out * w).sum().backward()
            res[fused] = (out.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
        finally:
            deform.FUSED_MLP = True
    rel_ok(res[True][0], res[False][0], tol=1e-5, what="mlp out")
    for n in res[False][1]:
        rel_ok(res[True][1][n], res[False][1][n], tol=2e-5, what="d_" + n)


def test_fused_adam_rows_matches_two_torch_groups():
    """Row-structured Adam (features [P,16,3]: DC at lr, rest at lr/20) == torch Adam with two parameter groups."""
    from rodygs_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    P, K = 1237, 16
    feats, gr = torch.randn(P, K, 3, generator=g), torch.randn(P, K, 3, generator=g)
    dc = feats[:, :1].clone().requires_grad_(True)
    rest = feats[:, 1:].clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [dc], "lr": 2.5e-3}, {"params": [rest], "lr": 2.5e-3 / 20}], eps=1e-15)
    pd = feats.clone().to(DEV)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 4):
        dc.grad, rest.grad = (gr[:, :1] * step).clone(), (gr[:, 1:] * step).clone()
        opt.step()
        gd = (gr * step).to(DEV)
        _lib.check(L.rdg_adam_step_rows(pd.numel(), pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), K * 3, 3,
                                        2.5e-3, 2.5e-3 / 20, 0.9, 0.999, 1e-15, step, _lib.stream_ptr()), "adam rows")
    want = torch.cat([dc.detach(), rest.detach()], dim=1)
    rel_ok(pd - feats.to(DEV), want - feats, tol=1e-5, what="adam rows update")


@pytest.mark.parametrize("NV", [1, 3, 5, 8, 16])
def test_multi_view_getter_matches_single_view_getter(NV):
    """rdg_dyn_getter_views_* (one launch for the camera times of a whole step) against the single-view fused getter
    run once per view: forward bit for bit, backward = the sum over the views."""
    from rodygs_amd import _lib
    from rodygs_amd.deform import _birth_order, dynamic_gaussians
    L = _lib.lib()
    g = torch.Generator().manual_seed(5 + NV)
    P, Tu, stride = 5003, 12, 5120
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(DEV)   # noqa: E731
    xyz, scaling, rotation, opacity, coeff = rnd(P, 3), 0.3 * rnd(P, 3), rnd(P, 4), rnd(P, 1), 0.2 * rnd(P, 16)
    ti = torch.randint(0, Tu, (P,), generator=g).to(DEV)
    table, bt = 0.1 * rnd(Tu, 16, 7), 0.1 * rnd(NV, 16, 7)
    bases_all = torch.cat([table.unsqueeze(0).expand(NV, -1, -1, -1), bt.unsqueeze(1)], dim=1).contiguous()
    gm, gs_, gr, go = rnd(NV, stride, 3), rnd(NV, stride, 3), rnd(NV, stride, 4), rnd(NV, stride, 1)
    f32 = dict(dtype=torch.float32, device=DEV)
    m3, ro = torch.zeros(NV, stride, 3, **f32), torch.zeros(NV, stride, 4, **f32)
    sc, op = torch.zeros(P, 3, **f32), torch.zeros(P, 1, **f32)
    assert L.rdg_dyn_getter_views_supported(16, Tu, NV)
    _lib.check(L.rdg_dyn_getter_views_forward(P, Tu, NV, stride, coeff.data_ptr(), ti.data_ptr(), bases_all.data_ptr(),
                                              5.0, xyz.data_ptr(), scaling.data_ptr(), rotation.data_ptr(),
                                              opacity.data_ptr(), m3.data_ptr(), sc.data_ptr(), ro.data_ptr(),
                                              op.data_ptr(), _lib.stream_ptr()), "views fwd")
    d = {k: torch.zeros_like(t) for k, t in (("xyz", xyz), ("scaling", scaling), ("rotation", rotation),
                                             ("opacity", opacity), ("coeff", coeff))}
    d_bases = torch.zeros_like(bases_all)
    order, inv, _ = _birth_order(ti)
    sws = torch.empty(L.rdg_deform_sorted_views_ws_bytes(P, NV), dtype=torch.uint8, device=DEV)
    _lib.check(L.rdg_dyn_getter_views_backward(P, Tu, NV, stride, coeff.data_ptr(), ti.data_ptr(), bases_all.data_ptr(),
                                               5.0, scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(),
                                               gm.data_ptr(), gs_.data_ptr(), gr.data_ptr(), go.data_ptr(),
                                               d["xyz"].data_ptr(), d["scaling"].data_ptr(), d["rotation"].data_ptr(),
                                               d["opacity"].data_ptr(), d["coeff"].data_ptr(), d_bases.data_ptr(),
                                               order.data_ptr(), inv.data_ptr(), sws.data_ptr(), _lib.stream_ptr()),
               "views bwd")
    want = {k: torch.zeros_like(t) for k, t in d.items()}
    for v in range(NV):
        leaves = [t.clone().requires_grad_(True) for t in (xyz, scaling, rotation, opacity, coeff)]
        bv = bases_all[v].clone().requires_grad_(True)
        o = dynamic_gaussians(*leaves, ti, bv, 5.0)
        assert torch.equal(o[0], m3[v, :P]) and torch.equal(o[2], ro[v, :P])
        assert torch.equal(o[1], sc) and torch.equal(o[3], op)
        torch.autograd.backward(o, [gm[v, :P], gs_[v, :P], gr[v, :P], go[v, :P]])
        for k, t in zip(want, leaves):
            want[k] += t.grad
        rel_ok(d_bases[v], bv.grad, tol=2e-5, what=f"views d_bases[{v}]")
    for k in want:
        rel_ok(d[k], want[k], tol=2e-5, what="views d_" + k)


def _replica_and_shards(P, W_img, H_img, frames, world, seed=21):
    from rodygs_amd.sharded import ShardedDynamicScene
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(P, W_img, H_img, 3, seed=seed)
    tgt = O.synthetic_scene(max(P // 4, 500), W_img, H_img, 3, seed=seed + 1)
    ds = DynamicScene(sc, num_frames=frames, device=DEV)
    ds.make_ground_truth(tgt, range(frames))
    ds.train_step(0, perm=[1])                                   # non-trivial parameters and Adam moments
    shards = [ShardedDynamicScene.from_replica(ds, r, world, exchange=object()) for r in range(world)]
    return ds, shards


@pytest.mark.parametrize("P,world", [(20000, 4), (5003, 3), (9000, 8), (4100, 16), (37, 8), (5, 8)])
def test_sharded_step_matches_replicated_frame_dp(P, world):
    """Gaussian-sharded frame-DP (rodygs_amd/sharded.py) against the replicated all-reduce formulation it replaces:
    `world` virtual ranks inside this process, each owning a slice of the cloud and rendering one camera.  The
    records that cross the exchange must reproduce the replicated render bit for bit (same loss, same tile lists);
    the gradients summed over cameras agree to float-atomics noise; and Adam from identical gradients is bitwise."""
    from rodygs_amd.losses import fused_photometric_loss
    from rodygs_amd.sharded import run_virtual_step
    from rodygs_amd.trainstep import fused_adam_
    frames = 12
    ds, shards = _replica_and_shards(P, 320, 240, frames, world)
    perm = list(range(frames))
    step = 3
    # replicated reference: one camera after the other, gradient buckets summed (what the all-reduce produces)
    acc, acc_sp, ref_losses = torch.zeros_like(ds.fp.flat_grad), torch.zeros_like(ds.sp.flat_grad), []
    for r in range(world):
        f = perm[(step * world + r) % frames]
        out, _ = ds.render(f)
        loss = fused_photometric_loss(out[0], ds.gt[f], 0.2)
        loss.backward()
        ref_losses.append(float(loss))
        acc += ds.fp.flat_grad
        acc_sp += ds.sp.flat_grad
    # sharded: stop before Adam to compare gradients, then finish the step
    from rodygs_amd import sharded as S
    real_update = S.ShardedDynamicScene.phase_update
    S.ShardedDynamicScene.phase_update = lambda self: None
    try:
        losses = run_virtual_step(shards, step, perm)
    finally:
        S.ShardedDynamicScene.phase_update = real_update
    assert [float(x) for x in losses] == ref_losses              # forward is bit-exact through the record exchange
    per = shards[0].per
    for k in ds.fp.names:
        o, m = ds.fp.offsets[k]
        want = acc[o:o + m].view(ds.fp.shapes[k])
        got = torch.cat([sh.fp[k].grad for sh in shards])
        assert got.shape == want.shape
        rel_ok(got, want, tol=2e-4, what="sharded d_" + k)
    for sh in shards:
        rel_ok(sh.sp.flat_grad, acc_sp, tol=2e-4, what="sharded small bucket")
        assert sh.lo == min(sh.rank * per, P) and sh.n == min(per, max(P - sh.lo, 0))      # (5, 8): three empty slices
    # Adam on the slices == Adam on the replica when fed the same gradients
    ds.fp.flat_grad.copy_(acc)
    ds.sp.flat_grad.copy_(shards[0].sp.flat_grad)
    for sh in shards:
        for k in ds.fp.names:
            o, m = ds.fp.offsets[k]
            sh.fp[k].grad.copy_(acc[o:o + m].view(ds.fp.shapes[k])[sh.lo:sh.lo + sh.n])
        sh.phase_update()
    fused_adam_(ds.fp, row_lr=ds.row_lr, extra=(ds.sp,))
    for k in ds.fp.names:
        got, want = torch.cat([sh.fp[k].detach() for sh in shards]), ds.fp[k].detach()
        assert torch.equal(got, want), (k, int((got != want).sum()), float((got - want).abs().max()),
                                        (got != want).reshape(got.shape[0], -1).any(1).nonzero().flatten()[:8].tolist())
    for sh in shards:
        assert torch.equal(sh.sp.flat, ds.sp.flat)


def test_sharded_training_reduces_loss_like_replicated():
    """A few dozen sharded steps (4 virtual ranks) train: the loss falls, every rank keeps identical MLP / pose
    parameters, and the parameters stay close to the replicated run fed the same cameras."""
    from rodygs_amd.sharded import run_virtual_step
    frames, world = 8, 4
    ds, shards = _replica_and_shards(12000, 256, 192, frames, world, seed=31)
    perm = list(range(frames))
    hist = []
    for step in range(1, 25):
        hist.append(np.mean([float(x) for x in run_virtual_step(shards, step, perm)]))
    assert all(np.isfinite(hist)) and np.mean(hist[-6:]) < np.mean(hist[:6])
    for sh in shards[1:]:
        assert torch.equal(sh.sp.flat, shards[0].sp.flat)
    assert shards[0].visible_count() > 0


@pytest.mark.parametrize("step", [5, 6])
def test_sharded_full_loss_step_matches_replicated(step):
    """The config-5 loss set inside the sharded step (3 virtual ranks): Pearson depth on the camera rank, motion L1 /
    sparsity on the owner's slice, the basis regulariser on the replicated table and -- on step 5 -- RigidityLoss on an
    all-gathered copy of the cloud with its gradient reduce-scattered back.  With every rank drawing from its own
    seeded random stream the summed gradients must equal those of the replicated trainer over the same cameras."""
    from rodygs_amd.sharded import ShardedDynamicScene, run_virtual_step
    from rodygs_amd import sharded as S
    from rodygs_amd.trainstep import DynamicScene
    P, world, frames = 9001, 3, 6
    sc = O.synthetic_scene(P, 320, 256, 3, seed=61)
    tgt = O.synthetic_scene(2500, 320, 256, 3, seed=62)
    ds = DynamicScene(sc, num_frames=frames, device=DEV, full_losses=True)
    ds.make_ground_truth(tgt, range(frames))
    ds.train_step(1, perm=[1])
    shards = [ShardedDynamicScene.from_replica(ds, r, world, exchange=object()) for r in range(world)]
    perm = list(range(frames))
    acc, acc_sp, ref_losses = torch.zeros_like(ds.fp.flat_grad), torch.zeros_like(ds.sp.flat_grad), []
    for r in range(world):
        torch.manual_seed(100 + r)
        loss = ds._full_loss(step, perm[(step * world + r) % frames])
        loss.backward()
        ref_losses.append(float(loss.detach()))
        acc += ds.fp.flat_grad
        acc_sp += ds.sp.flat_grad
    for r, sh in enumerate(shards):
        sh.seed_rng(100 + r)
    real_update = S.ShardedDynamicScene.phase_update
    S.ShardedDynamicScene.phase_update = lambda self: None
    try:
        losses = run_virtual_step(shards, step, perm)
    finally:
        S.ShardedDynamicScene.phase_update = real_update
    assert abs(sum(float(x) for x in losses) - sum(ref_losses)) <= 2e-5 * abs(sum(ref_losses))
    for k in ds.fp.names:
        o, m = ds.fp.offsets[k]
        rel_ok(torch.cat([sh.fp[k].grad for sh in shards]), acc[o:o + m].view(ds.fp.shapes[k]), tol=5e-4,
               what=f"sharded full-loss d_{k} (step {step})")
    for sh in shards:
        rel_ok(sh.sp.flat_grad, acc_sp, tol=5e-4, what="sharded full-loss small bucket")


def test_sharded_training_with_densification():
    """Densify-and-prune inside the sharded loop: every rank gathers the screen-space statistics of ITS Gaussians over
    all cameras of each step (no collective), densifies its own slice, and the ranks only agree on the new row stride.
    The statistics equal what the replicated trainer accumulates over the same cameras; training carries on."""
    from rodygs_amd.losses import fused_photometric_loss
    from rodygs_amd.sharded import run_virtual_densify, run_virtual_step
    frames, world = 8, 4
    ds, shards = _replica_and_shards(12000, 256, 192, frames, world, seed=51)
    perm = list(range(frames))
    for sh in shards:
        sh.track_densification()
    ds.track_densification()
    # one step on both formulations, statistics compared before any parameter moves apart
    for r in range(world):
        f = perm[(1 * world + r) % frames]
        out, m2 = ds.render(f)
        fused_photometric_loss(out[0], ds.gt[f], 0.2).backward()      # the replica's statistics: updated inside backward
    run_virtual_step(shards, 1, perm)
    for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
        got = torch.cat([getattr(sh.stats, name) for sh in shards])
        rel_ok(got, getattr(ds.stats, name), tol=2e-4, what="sharded densify stats " + name)
    hist = [np.mean([float(x) for x in run_virtual_step(shards, s_, perm)]) for s_ in range(2, 20)]
    infos = run_virtual_densify(shards, max_grad=2e-5, min_opacity=0.05, percent_dense=0.002)
    assert sum(i["cloned"] for i in infos) > 0 and sum(i["split"] for i in infos) > 0
    assert len({sh.stride for sh in shards}) == 1 and shards[0].stride % 256 == 0
    assert infos[0]["P"] == sum(sh.n for sh in shards) != 12000 and max(sh.n for sh in shards) <= shards[0].stride
    assert all(sh.time_ind.shape[0] == sh.n == sh.fp.shapes["xyz"][0] for sh in shards)
    assert all(float(sh.stats.denom.sum()) == 0.0 for sh in shards)
    hist += [np.mean([float(x) for x in run_virtual_step(shards, s_, perm)]) for s_ in range(20, 44)]
    assert all(np.isfinite(hist)) and np.mean(hist[-6:]) < np.mean(hist[:6])
    assert all(float(sh.fp["xyz"].grad.abs().sum()) > 0 for sh in shards)


def test_sharded_step_over_a_process_group():
    """`train_step` with the real collectives (all_to_all_single x2 + all_reduce over RCCL) on a 1-rank group: the
    exchange is then the identity and the step must equal the virtual single-rank step."""
    import socket
    import torch.distributed as dist
    from rodygs_amd.sharded import DistExchange, ShardedDynamicScene, run_virtual_step
    ds, (a,) = _replica_and_shards(8000, 256, 192, 6, 1, seed=41)
    b = ShardedDynamicScene.from_replica(ds, 0, 1, exchange=DistExchange())
    if not dist.is_initialized():
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(DEV, torch.cuda.current_device()))
    try:
        la = run_virtual_step([a], 2, list(range(6)))[0]
        lb = b.train_step(2, list(range(6)))
        torch.cuda.synchronize()
        assert float(la) == float(lb)
        rel_ok(b.fp.flat_grad, a.fp.flat_grad, tol=2e-4, what="1-rank group gradients")
        full = b.gather_params()
        assert torch.equal(full["xyz"], b.fp["xyz"].detach())
        from rodygs_amd.checkpoint import export_state_dict, flat_params_from_state_dict
        sd = b.export_state_dict(iteration=3)                          # reference checkpoint layout of the whole cloud
        want = export_state_dict(b.fp, 3, b.sh_degree, b.spatial_lr_scale, deform_network=b.net,
                                 feature_lr_rest=b.row_lr["features"][2])
        for k in want["model"]:
            if k != "_deform_network":
                assert torch.equal(sd["model"][k], want["model"][k]), k
        assert sd["model"]["_timestep"].shape[0] == b.P_total and sd["camera"]["R_c2ws_quat"].shape == (6, 4)
        st0, st1 = sd["optim"]["optimizer"]["state"], want["optim"]["optimizer"]["state"]
        assert all(torch.equal(st0[i]["exp_avg_sq"], st1[i]["exp_avg_sq"]) for i in st1)
        back = flat_params_from_state_dict(sd, {k: b.fp.lr[k] for k in b.fp.names}, DEV)
        assert torch.equal(back["motion_coeff"].detach(), b.fp["motion_coeff"].detach())
        b.track_densification()
        for s_ in range(3, 9):
            b.train_step(s_, list(range(6)))
        info = b.densify(max_grad=2e-5, min_opacity=0.05, percent_dense=0.002)
        assert info["P"] == b.n == b.P_total and b.stride >= b.n
        assert np.isfinite(float(b.train_step(9, list(range(6)))))
        # config-5 loss set through the real all_gather / reduce_scatter (rigidity on step 10)
        from rodygs_amd.trainstep import DynamicScene
        sc = O.synthetic_scene(6000, 256, 192, 3, seed=43)
        dsf = DynamicScene(sc, num_frames=6, device=DEV, full_losses=True)
        dsf.make_ground_truth(O.synthetic_scene(1500, 256, 192, 3, seed=44), range(6))
        c = ShardedDynamicScene.from_replica(dsf, 0, 1, exchange=DistExchange())
        vals = [float(c.train_step(s_, list(range(6)))) for s_ in range(9, 13)]
        torch.cuda.synchronize()
        assert all(np.isfinite(vals)) and float(c.fp["motion_coeff"].grad.abs().sum()) > 0
    finally:
        dist.destroy_process_group()


def _two_process_worker(rank, world, port, outdir):
    """One rank of a REAL multi-process sharded run: both processes share cuda:0, the process group is gloo and the
    exchange is staged through host memory (HostStagedExchange)."""
    import os as _os
    import torch.distributed as dist
    _os.environ["MASTER_ADDR"], _os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import rasterizer_oracle as Or
    from rodygs_amd.sharded import HostStagedExchange, ShardedDynamicScene
    from rodygs_amd.trainstep import DynamicScene
    torch.cuda.set_device(0)
    out = {}
    for full in (False, True):
        sc = Or.synthetic_scene(6001, 256, 192, 3, seed=71)
        ds = DynamicScene(sc, num_frames=6, device="cuda", full_losses=full)
        ds.make_ground_truth(Or.synthetic_scene(1500, 256, 192, 3, seed=72), range(6))
        ss = ShardedDynamicScene.from_replica(ds, rank, world, exchange=HostStagedExchange())
        ss.seed_rng(500 + rank)
        if full:
            ss.a2a_chunks = 3           # the owner stage in row chunks, each chunk's records exchanged on the side stream
        losses = [float(ss.train_step(s_, list(range(6)))) for s_ in range(4, 7)]      # step 5: rigidity (full)
        sd = ss.export_state_dict(iteration=7)
        out[full] = {"losses": losses, "params": {k: v.cpu() for k, v in ss.gather_params().items()},
                     "sp": ss.sp.flat.cpu(), "ckpt_xyz": sd["model"]["_xyz"].cpu(),
                     "ckpt_m": sd["optim"]["optimizer"]["state"][0]["exp_avg"].cpu()}
        if not full:
            ss.track_densification()
            for s_ in range(7, 13):
                ss.train_step(s_, list(range(6)))
            info = ss.densify(max_grad=2e-5, min_opacity=0.05, percent_dense=0.002)
            out["densify"] = (info, ss.n, ss.stride, ss.counts, float(ss.train_step(13, list(range(6)))))
    torch.save(out, _os.path.join(outdir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_in_two_real_processes():
    """The sharded `train_step` as two operating-system processes (one rank each, sharing this GPU, gloo group, exchange
    staged through the host) against the same two ranks run as virtual ranks in this process: identical first-step
    losses, the same parameters up to float-atomics noise after three steps (photometric and config-5 loss sets), and a
    collective densification that leaves both ranks with the same stride and slice table."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    from rodygs_amd.sharded import ShardedDynamicScene, run_virtual_step
    from rodygs_amd.trainstep import DynamicScene
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_two_process_worker, args=(2, port, d), nprocs=2, join=True)
        got = [torch.load(f"{d}/p{r}.pt", weights_only=False) for r in range(2)]
    for full in (False, True):
        sc = O.synthetic_scene(6001, 256, 192, 3, seed=71)
        ds = DynamicScene(sc, num_frames=6, device=DEV, full_losses=full)
        ds.make_ground_truth(O.synthetic_scene(1500, 256, 192, 3, seed=72), range(6))
        shards = [ShardedDynamicScene.from_replica(ds, r, 2, exchange=object()) for r in range(2)]
        for r, sh in enumerate(shards):
            sh.seed_rng(500 + r)
        want = [[float(x) for x in run_virtual_step(shards, s_, list(range(6)))] for s_ in range(4, 7)]
        for r in range(2):
            # same forward, same draws (the loss reduction itself sums with float atomics: last-bit differences)
            assert abs(got[r][full]["losses"][0] - want[0][r]) <= 1e-6 * abs(want[0][r]), (full, r)
            assert np.allclose(got[r][full]["losses"], [w[r] for w in want], rtol=2e-3), (full, r)
            assert torch.equal(got[0][full]["sp"], got[r][full]["sp"])               # replicated bucket stays in step
        for k, v in got[0][full]["params"].items():
            ref = torch.cat([sh.fp[k].detach() for sh in shards]).cpu()
            assert v.shape == ref.shape and torch.equal(v, got[1][full]["params"][k])
            if k == "xyz":                                      # the exported checkpoint holds the same gathered cloud
                assert torch.equal(got[0][full]["ckpt_xyz"], v) and torch.equal(got[1][full]["ckpt_xyz"], v)
                assert torch.equal(got[0][full]["ckpt_m"], got[1][full]["ckpt_m"]) and got[0][full]["ckpt_m"].shape == v.shape
            dlt = (v - ref).abs()
            assert float(dlt.mean()) <= 2e-4 * (float(ref.abs().mean()) + 1e-3), (full, k, float(dlt.mean()))
    i0, i1 = got[0]["densify"], got[1]["densify"]
    assert i0[2] == i1[2] and i0[3] == i1[3] and i0[3] == [i0[1], i1[1]] and i0[0]["P"] == i0[1] + i1[1]
    assert np.isfinite(i0[4]) and np.isfinite(i1[4])


def test_owner_stage_in_row_chunks_is_bit_identical():
    """The chunked owner stage (rdg_preprocess_forward_views_rows + the getter on row ranges), which the pipelined
    all-to-all #1 runs, writes exactly the records, radii and tile counts of the one-launch owner stage -- ragged last
    shard, three virtual ranks, chunk counts that do and do not divide the shard."""
    from rodygs_amd.sharded import ShardedDynamicScene
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(5003, 256, 192, 3, seed=91)
    ds = DynamicScene(sc, num_frames=6, device=DEV)
    ds.make_ground_truth(O.synthetic_scene(1200, 256, 192, 3, seed=92), range(6))
    shards = [ShardedDynamicScene.from_replica(ds, r, 3, exchange=object()) for r in range(3)]
    for sh in shards:
        assert sh.chunk_ranges(1) == [(0, sh.stride)] and sh.chunk_ranges(64)[-1][1] == sh.stride
        sh.phase_owner_forward(7, list(range(6)))
        torch.cuda.synchronize()
        want = (sh.geom_own.clone(), sh.radii_own.clone(), sh.m3.clone(), sh.ro.clone(), sh.sc.clone(), sh.op.clone())
        for chunks in (2, 3, 5):
            for t in (sh.geom_own, sh.radii_own, sh.m3, sh.ro, sh.sc, sh.op):
                t.zero_()
            sh.phase_owner_forward(7, list(range(6)), chunks=chunks)
            torch.cuda.synchronize()
            got = (sh.geom_own, sh.radii_own, sh.m3, sh.ro, sh.sc, sh.op)
            for a, b in zip(got, want):
                assert torch.equal(a, b), (sh.rank, chunks)


def _two_process_allreduce_worker(rank, world, port, outdir):
    import os as _os
    import torch.distributed as dist
    _os.environ["MASTER_ADDR"], _os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import rasterizer_oracle as Or
    from rodygs_amd.trainstep import DynamicScene
    torch.cuda.set_device(0)
    sc = Or.synthetic_scene(6001, 256, 192, 3, seed=71)
    ds = DynamicScene(sc, num_frames=6, device="cuda")
    ds.make_ground_truth(Or.synthetic_scene(1500, 256, 192, 3, seed=72), range(6))
    losses = [float(ds.train_step(s_, rank, world, list(range(6)))) for s_ in range(4, 8)]
    torch.cuda.synchronize()
    torch.save({"losses": losses, "flat": ds.fp.flat.cpu(), "sp": ds.sp.flat.cpu()}, _os.path.join(outdir, f"a{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_frame_dp_in_two_real_processes():
    """The replicated formulation (bucketed, overlapped all-reduce + piecewise Adam) as two real processes sharing this
    GPU over gloo: after four steps both ranks must hold bit-identical parameters (they applied the same summed
    gradients), and those agree with the Gaussian-sharded formulation of the same job up to float-atomics noise."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    from rodygs_amd.sharded import ShardedDynamicScene, run_virtual_step
    from rodygs_amd.trainstep import DynamicScene
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_two_process_allreduce_worker, args=(2, port, d), nprocs=2, join=True)
        a, b = [torch.load(f"{d}/a{r}.pt", weights_only=False) for r in range(2)]
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["sp"], b["sp"])
    assert a["losses"] != b["losses"]                                    # ... while rendering different cameras
    sc = O.synthetic_scene(6001, 256, 192, 3, seed=71)
    ds = DynamicScene(sc, num_frames=6, device=DEV)
    ds.make_ground_truth(O.synthetic_scene(1500, 256, 192, 3, seed=72), range(6))
    shards = [ShardedDynamicScene.from_replica(ds, r, 2, exchange=object()) for r in range(2)]
    want = [[float(x) for x in run_virtual_step(shards, s_, list(range(6)))] for s_ in range(4, 8)]
    assert np.allclose(a["losses"], [w[0] for w in want], rtol=2e-3) and np.allclose(b["losses"], [w[1] for w in want], rtol=2e-3)
    for k in ds.fp.names:
        o, m = ds.fp.offsets[k]
        ref = torch.cat([sh.fp[k].detach() for sh in shards]).cpu().reshape(-1)
        dlt = (a["flat"][o:o + m] - ref).abs()
        assert float(dlt.mean()) <= 2e-4 * (float(ref.abs().mean()) + 1e-3), (k, float(dlt.mean()))


def test_test_time_pose_optimisation_recovers_camera():
    """The evaluator's PoseOptimizer (eval.py:342-420) on the HIP rasterizer: frozen Gaussians, a held-out camera whose
    pose is only known approximately (initialised from the calibrated pose of the nearest train frame), Adam on the
    camera-to-world quaternion + translation through dL/dviewmatrix.  The pose error must shrink and the PSNR rise."""
    import math
    from rodygs_amd.checkpoint import psnr
    from rodygs_amd.pose_optimizer import PoseOptimizer
    from rodygs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    sc = O.synthetic_scene(4000, 192, 144, 3, seed=81)
    dev = torch.device(DEV)
    gs = {k: sc[k].to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    rs = GaussianRasterizationSettings(144, 192, sc["tanfovx"], sc["tanfovy"], torch.zeros(3, device=dev), 1.0,
                                       sc["projmatrix"].to(dev), 3, False, False, True, True)

    def render(vm):
        return GaussianRasterizer(rs)(means3D=gs["means3D"], means2D=torch.zeros_like(gs["means3D"]), shs=gs["shs"],
                                      opacities=gs["opacities"], scales=gs["scales"], rotations=gs["rotations"],
                                      viewmatrix=vm)[0]

    def c2w(angle, shift):
        c, s_ = math.cos(angle), math.sin(angle)
        m = torch.eye(4)
        m[:3, :3] = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]])
        m[:3, 3] = torch.tensor(shift)
        return m

    train_gt = torch.stack([c2w(0.02 * i, [0.15 * i, 0.0, 0.0]) for i in range(-3, 4)])
    calibrated = train_gt.clone()                     # the trained poses of those frames (here: exact)
    test_gt = c2w(0.031, [0.21, 0.03, -0.02])         # between train frames 1 and 2 of the right half
    with torch.no_grad():
        rgb = render(torch.inverse(test_gt).t().contiguous().to(dev))
    po = PoseOptimizer(calibrated.to(dev), train_gt.to(dev), render, camera_lr=2e-3, num_opts=200)
    cam = po(test_gt.to(dev), rgb)
    with torch.no_grad():
        w2c0 = torch.inverse(calibrated[4]).to(dev)                    # the start: nearest train frame (index 4)
        p0 = float(psnr(rgb, render(w2c0.t().contiguous())))
        p1 = float(psnr(rgb, render(cam.viewmatrix())))
        est = torch.inverse(cam.world_view_transform).cpu()
    e0 = float(torch.norm(calibrated[4][:3, 3] - test_gt[:3, 3]))
    e1 = float(torch.norm(est[:3, 3] - test_gt[:3, 3]))
    assert float(po.history[-1]) < 0.05 * float(po.history[0]), (float(po.history[0]), float(po.history[-1]))
    assert p1 > p0 + 8.0 and e1 < 0.25 * e0, (p0, p1, e0, e1)


def test_pose_optimisation_trajectory_matches_the_oracle():
    """The evaluator's test-time pose optimisation (/root/reference/src/evaluator/eval.py:342-420: Adam(lr, eps 1e-15) on a
    LearnableCamera's quaternion + translation, l2 loss on the rendered image, Gaussians frozen) run step for step
    through the HIP rasterizer and through the oracle from the same start: 60 steps, the loss curve and the pose
    parameters after every step must coincide (the only gradient in play is dL/dviewmatrix)."""
    import math
    from rodygs_amd.pose_optimizer import LearnablePose, l2_loss
    from rodygs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    W, H, steps, lr = 128, 96, 60, 2e-3
    sc = O.synthetic_scene(2500, W, H, 3, seed=83)
    dev = torch.device(DEV)
    gs = {k: sc[k].to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    rs = GaussianRasterizationSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3, device=dev), 1.0,
                                       sc["projmatrix"].to(dev), 3, False, False, True, True)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 3)

    def render_hip(vm):
        return GaussianRasterizer(rs)(means3D=gs["means3D"], means2D=torch.zeros_like(gs["means3D"]), shs=gs["shs"],
                                      opacities=gs["opacities"], scales=gs["scales"], rotations=gs["rotations"],
                                      viewmatrix=vm)[0]

    def render_oracle(vm):
        return O.rasterize(sc["means3D"], torch.zeros(sc["means3D"].shape[0], 3), sc["opacities"], vm, st, shs=sc["shs"],
                           scales=sc["scales"], rotations=sc["rotations"])[0]

    def c2w(angle, shift):
        c, s_ = math.cos(angle), math.sin(angle)
        m = torch.eye(4)
        m[:3, :3] = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]])
        m[:3, 3] = torch.tensor(shift)
        return m

    target = torch.inverse(c2w(0.031, [0.21, 0.03, -0.02]))
    start = torch.inverse(c2w(0.02, [0.15, 0.0, 0.0]))
    with torch.no_grad():
        rgb = render_oracle(target.t().contiguous())

    def optimise(render, device, view_of):
        cam = LearnablePose(start[:3, :3].clone(), start[:3, 3].clone()).to(device)
        opt = torch.optim.Adam(cam.parameters(), lr=lr, eps=1e-15)
        gt = rgb.to(device)
        losses, traj = [], []
        for _ in range(steps):
            loss = l2_loss(render(view_of(cam)), gt)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(loss))
            traj.append(torch.cat([cam.R_c2w_quat.detach().cpu(), cam.T_c2w.detach().cpu()]))
        return torch.tensor(losses, dtype=torch.float64), torch.stack(traj).double()

    lh, th = optimise(render_hip, dev, lambda cam: cam.viewmatrix())                       # HIP pose op + rasterizer
    lo, to = optimise(render_oracle, "cpu", lambda cam: cam.world_view_transform.t().contiguous())
    assert float(lo[-1]) < 0.2 * float(lo[0])                                              # the optimisation did its job
    moved = float((to[-1] - to[0]).abs().max())
    assert moved > 10 * lr
    assert float(((lh - lo).abs() / lo).max()) <= 2e-3, ((lh - lo).abs() / lo).max()
    assert float((th - to).abs().max()) <= 2e-2 * moved, (float((th - to).abs().max()), moved)
    assert float((th[:10] - to[:10]).abs().max()) <= 1e-5                                  # early steps: bit-level agreement


def test_sh_adam_in_backward_equals_separate_optimiser_step():
    """rdg_preprocess_backward_adam (the SH features stepped inside the per-Gaussian backward kernel, dL/dshs never
    written) against rdg_preprocess_backward + rdg_adam_step_multi on the gradient it writes, from the SAME gradient rows:
    parameters and both moments bit for bit, every other output identical."""
    from rodygs_amd import _lib
    from rodygs_amd.rasterizer import GaussianRasterizationSettings, _c_settings
    L = _lib.lib()
    P, W, H, K = 5003, 256, 176, 16
    sc = O.synthetic_scene(P, W, H, 3, seed=91)
    dev = torch.device(DEV)
    t = {k: sc[k].to(dev).contiguous() for k in ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix",
                                                   "projmatrix")}
    rs = GaussianRasterizationSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3, device=dev), 1.0,
                                       t["projmatrix"], 3, False, False, True, True)
    cs = _c_settings(rs, P, K)
    u8, f32 = dict(dtype=torch.uint8, device=dev), dict(dtype=torch.float32, device=dev)
    n_tiles, cap = ((W + 15) // 16) * ((H + 15) // 16), 40 * P
    geom = torch.empty(L.rdg_geom_bytes(P), **u8)
    binning = torch.empty(L.rdg_binning_bytes(cap, n_tiles), **u8)
    image = torch.empty(L.rdg_image_bytes(H, W), **u8)
    outs = [torch.empty(c, H, W, **f32) for c in (3, 1, 3, 1)]
    radii, nren = torch.empty(P, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    bg, st = torch.zeros(3, **f32), _lib.stream_ptr()
    _lib.check(L.rdg_rasterize_forward(C.byref(cs), bg.data_ptr(), t["means3D"].data_ptr(), t["shs"].data_ptr(), None,
                                       t["opacities"].data_ptr(), t["scales"].data_ptr(), t["rotations"].data_ptr(), None,
                                       t["viewmatrix"].data_ptr(), t["projmatrix"].data_ptr(), geom.data_ptr(),
                                       binning.data_ptr(), cap, image.data_ptr(), *[o.data_ptr() for o in outs],
                                       radii.data_ptr(), nren.data_ptr(), st), "fwd")
    assert 0 < int(nren.item()) <= cap
    g_color = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    gws = torch.empty(L.rdg_grad_bytes(P), **u8)
    _lib.check(L.rdg_composite_backward(C.byref(cs), bg.data_ptr(), geom.data_ptr(), binning.data_ptr(), cap,
                                        image.data_ptr(), g_color.data_ptr(), None, None, None, gws.data_ptr(), st), "cbwd")

    def outputs():
        return {k: torch.zeros(*shp, **f32) for k, shp in (("m3", (P, 3)), ("m2", (P, 3)), ("op", (P, 1)), ("sc", (P, 3)),
                                                           ("ro", (P, 4)), ("vm", (4, 4)))}

    a, d_sh = outputs(), torch.zeros(P, K, 3, **f32)
    _lib.check(L.rdg_preprocess_backward(C.byref(cs), t["means3D"].data_ptr(), t["shs"].data_ptr(), None,
                                         t["opacities"].data_ptr(), t["scales"].data_ptr(), t["rotations"].data_ptr(), None,
                                         t["viewmatrix"].data_ptr(), t["projmatrix"].data_ptr(), radii.data_ptr(),
                                         geom.data_ptr(), gws.data_ptr(), a["m3"].data_ptr(), a["m2"].data_ptr(),
                                         d_sh.data_ptr(), None, a["op"].data_ptr(), a["sc"].data_ptr(), a["ro"].data_ptr(),
                                         None, a["vm"].data_ptr(), st), "pbwd")
    assert float(d_sh.abs().sum()) > 0
    gen = torch.Generator().manual_seed(2)
    m0, v0 = (0.01 * torch.randn(P, K, 3, generator=gen)).to(dev), (1e-4 * torch.rand(P, K, 3, generator=gen)).to(dev)
    for step in (1, 7):
        pa, ma, va = t["shs"].clone(), m0.clone(), v0.clone()
        seg = (_lib.RdgAdamSeg * 1)()            # the launch the train step uses for its parameter groups
        seg[0].n, seg[0].param, seg[0].grad = pa.numel(), pa.data_ptr(), d_sh.data_ptr()
        seg[0].exp_avg, seg[0].exp_avg_sq = ma.data_ptr(), va.data_ptr()
        seg[0].lr_head, seg[0].lr_tail, seg[0].row_len, seg[0].head_len = 2.5e-3, 2.5e-3 / 20, K * 3, 3
        _lib.check(L.rdg_adam_step_multi(1, seg, 0.9, 0.999, 1e-15, step, st), "adam multi")
        pb, mb, vb, b = t["shs"].clone(), m0.clone(), v0.clone(), outputs()
        _lib.check(L.rdg_preprocess_backward_adam(
            C.byref(cs), t["means3D"].data_ptr(), pb.data_ptr(), t["opacities"].data_ptr(), t["scales"].data_ptr(),
            t["rotations"].data_ptr(), t["viewmatrix"].data_ptr(), t["projmatrix"].data_ptr(), radii.data_ptr(),
            geom.data_ptr(), gws.data_ptr(), b["m3"].data_ptr(), b["m2"].data_ptr(), b["op"].data_ptr(), b["sc"].data_ptr(),
            b["ro"].data_ptr(), b["vm"].data_ptr(), mb.data_ptr(), vb.data_ptr(), 3, 2.5e-3, 2.5e-3 / 20, 0.9, 0.999, 1e-15,
            step, st), "pbwd adam")
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), step
        assert not torch.equal(pb, t["shs"])
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_sh_adam_sink_is_one_shot():
    """The optimizer-in-backward sink updates a saved tensor in place: a second backward through the same graph
    (loss.backward(retain_graph=True) twice, /root/reference/src/trainer/rodygs.py:310) must raise instead of stepping
    already-stepped parameters."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    P = 1500
    sc = O.synthetic_scene(P, 128, 96, 3, seed=14)
    rs = HS.make_settings(sc, 3)
    ins = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    shs = sc["shs"].clone().to(DEV)                         # the "parameter" the kernel steps in place
    m, v = torch.zeros_like(shs), torch.zeros_like(shs)
    sink = {"shs_adam": {"param": shs, "exp_avg": m, "exp_avg_sq": v, "head_len": 3, "lr_head": 2.5e-3,
                         "lr_tail": 2.5e-3 / 20, "betas": (0.9, 0.999), "eps": 1e-15, "step": 1}}
    m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=shs.requires_grad_(True),
                                 opacities=ins["opacities"], scales=ins["scales"], rotations=ins["rotations"],
                                 viewmatrix=ins["viewmatrix"], grad_sinks=sink)
    before = shs.detach().clone()
    loss = out[0].sum()
    loss.backward(retain_graph=True)
    after = shs.detach().clone()
    assert not torch.equal(before, after) and shs.grad is None
    with pytest.raises(RuntimeError, match="already applied the SH Adam step"):
        loss.backward()
    assert torch.equal(shs.detach(), after)                 # nothing was stepped twice


def test_reset_opacity_matches_reference_golden():
    """rdg_reset_opacity on a flat-bucket segment against what the imported reference produced on a real
    torch.optim.Adam (golden G11: reset_opacity + replace_tensor_to_optimizer, then the next Adam step)."""
    from rodygs_amd.densify import reset_opacity_
    from rodygs_amd.dp import FlatParams
    from rodygs_amd.trainstep import fused_adam_
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "optimizer_golden.npz"))
    n = g["reset_logit_in"].shape[0]
    fp = FlatParams({"xyz": ((n, 3), 1e-3), "opacity": ((n, 1), 0.05), "scaling": ((n, 3), 1e-3)}, torch.device(DEV))
    o, cnt = fp.offsets["opacity"]
    with torch.no_grad():
        fp["opacity"].copy_(torch.from_numpy(g["reset_logit_in"]))
        fp.exp_avg[o:o + cnt].copy_(torch.from_numpy(g["reset_m_in"]).reshape(-1))
        fp.exp_avg_sq[o:o + cnt].copy_(torch.from_numpy(g["reset_v_in"]).reshape(-1))
        fp.exp_avg[:o].fill_(0.25)
        fp.exp_avg_sq[o + cnt:].fill_(0.5)
        fp["xyz"].fill_(1.0)
    fp.step_count = int(g["reset_step"])
    p_obj, grad_obj = fp["opacity"], fp["opacity"].grad
    reset_opacity_(fp)
    assert fp["opacity"] is p_obj and fp["opacity"].grad is grad_obj        # nothing re-bound
    rel_ok(fp["opacity"], g["reset_logit_out"], tol=2e-6, what="reset logits")
    assert float(fp.exp_avg[o:o + cnt].abs().max()) == 0.0 and float(fp.exp_avg_sq[o:o + cnt].abs().max()) == 0.0
    assert float(fp.exp_avg[:o].min()) == 0.25 and float(fp.exp_avg_sq[o + cnt:].min()) == 0.5   # neighbours untouched
    assert float(fp["xyz"].min()) == 1.0 and fp.step_count == int(g["reset_step_out"])
    # the optimiser step that follows: moments restart from zero, the step counter (bias correction) continues
    with torch.no_grad():
        fp.flat_grad.zero_()
        fp["opacity"].grad.copy_(torch.from_numpy(g["next_grad"]))
    fused_adam_(fp, names=["opacity"])
    rel_ok(fp["opacity"], g["next_logit"], tol=1e-5, what="logits after the next Adam step")
    rel_ok(fp.exp_avg[o:o + cnt].view(n, 1), g["next_m"], tol=1e-6, what="exp_avg")
    rel_ok(fp.exp_avg_sq[o:o + cnt].view(n, 1), g["next_v"], tol=1e-6, what="exp_avg_sq")


def test_fused_adam_reproduces_the_reference_optimizer_step_from_a_checkpoint():
    """G12 on the GPU: the checkpoint of tests/golden/checkpoint_golden.npz (read by the REFERENCE loader, its optimizer
    built by the reference's trainer code, one torch Adam step taken there) -- the same state imported into the flat
    buckets, the same gradients, ONE fused HIP launch over both buckets (Gaussians + MLP) must land on the parameters
    the reference wrote into its own checkpoint."""
    import importlib.util
    from rodygs_amd import checkpoint as CK
    from rodygs_amd.deform import MLPBasisNetwork
    from rodygs_amd.trainstep import bind_module_to_flat, fused_adam_
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_golden.npz"))
    fp0, net0, sp0, g2t, cams, _ = M.checkpoint_inputs()
    sd = CK.export_state_dict(fp0, 5, 3, M.CKPT_SCALE, net0, g2t, cams, feature_lr_rest=M.CKPT_LR["feature_lr"] / 20.0,
                              deform_state=sp0, deform_lr=M.CKPT_DEFORM["deform_lr_init"])
    fp = CK.flat_params_from_state_dict(sd, dict(fp0.lr), DEV)
    net = MLPBasisNetwork(128, 16, 26, False)
    net.load_state_dict(sd["model"]["_deform_network"])
    sp = bind_module_to_flat(net.to(DEV), M.CKPT_DEFORM["deform_lr_init"], DEV)
    assert CK.restore_deform_state(sd, sp) and sp.step_count == fp.step_count == 5
    T = lambda k: torch.from_numpy(g[k]).to(DEV)   # noqa: E731
    # gradients in the reference's parameter order: xyz, f_dc, f_rest, opacity, scaling, rotation, 70 MLP tensors, coeff
    with torch.no_grad():
        fp["xyz"].grad.copy_(T("grad_0"))
        fp["features"].grad.copy_(torch.cat([T("grad_1"), T("grad_2")], dim=1))
        fp["opacity"].grad.copy_(T("grad_3"))
        fp["scaling"].grad.copy_(T("grad_4"))
        fp["rotation"].grad.copy_(T("grad_5"))
        for j, n in enumerate(CK.reference_mlp_param_names(16)):
            CK._mlp_segment(sp.flat_grad, sp, n).copy_(T(f"grad_{6 + j}"))
        fp["motion_coeff"].grad.copy_(T("grad_76"))
    K = fp.shapes["features"][1]
    fused_adam_(fp, row_lr={"features": (K * 3, 3, M.CKPT_LR["feature_lr"] / 20.0)}, extra=(sp,))
    torch.cuda.synchronize()
    want = {"xyz": T("ref_model._xyz"), "features": torch.cat([T("ref_model._features_dc"), T("ref_model._features_rest")], 1),
            "opacity": T("ref_model._opacity"), "scaling": T("ref_model._scaling"), "rotation": T("ref_model._rotation"),
            "motion_coeff": T("ref_model._motion_coeff")}
    for k, w in want.items():
        assert float((fp[k].detach() - w).abs().max()) <= 2e-6 * max(1.0, float(w.abs().max())), k
    for n in CK.reference_mlp_param_names(16):
        w = T("ref_mlp." + n)
        assert float((CK._mlp_segment(sp.flat, sp, n) - w).abs().max()) <= 2e-6 * max(1.0, float(w.abs().max())), n
    o, n_ = fp.offsets["xyz"]
    assert torch.allclose(fp.exp_avg[o:o + n_].view(-1), T("ref_state_0.exp_avg").reshape(-1), rtol=1e-5, atol=1e-9)
    assert torch.allclose(fp.exp_avg_sq[o:o + n_].view(-1), T("ref_state_0.exp_avg_sq").reshape(-1), rtol=1e-5, atol=1e-12)


def test_adam_lr_override_is_per_segment_and_per_step():
    """The reference re-sets the xyz group's learning rate every iteration (rodygs_static.py:143-149): lr_override
    changes that segment for that launch only; torch.optim.Adam with the same per-group lr is the reference."""
    from rodygs_amd.dp import FlatParams
    from rodygs_amd.trainstep import expon_lr, fused_adam_
    n = 3001
    dev = torch.device(DEV)
    fp = FlatParams({"xyz": ((n, 3), 8e-4), "scaling": ((n, 3), 1e-3), "opacity": ((n, 1), 0.05)}, dev)
    gen = torch.Generator().manual_seed(3)
    ref = {k: torch.nn.Parameter(torch.randn(*fp.shapes[k], generator=gen).to(dev)) for k in fp.names}
    with torch.no_grad():
        for k in fp.names:
            fp[k].copy_(ref[k])
    opt = torch.optim.Adam([{"params": [ref[k]], "lr": fp.lr[k], "name": k} for k in fp.names], lr=0.0, eps=1e-15)
    for it in range(1, 5):
        lr_xyz = expon_lr(it * 5000, 8e-4, 8e-6, 0, 0.01, 30000)
        for grp in opt.param_groups:
            if grp["name"] == "xyz":
                grp["lr"] = lr_xyz
        for k in fp.names:
            gk = torch.randn(*fp.shapes[k], generator=gen).to(dev)
            ref[k].grad = gk.clone()
            with torch.no_grad():
                fp[k].grad.copy_(gk)
        opt.step()
        fused_adam_(fp, lr_override={"xyz": lr_xyz})
        for k in fp.names:
            rel_ok(fp[k], ref[k], tol=2e-6, what=f"{k} after step {it}")
    assert fp.lr["xyz"] == 8e-4                                   # the override never sticks


def test_fused_motion_l1_sparsity_matches_reference_golden():
    """csrc/rdg_motionreg.hip against the values and gradients the imported reference produced for MotionL1Loss and
    MotionSparsityLoss (tests/golden/motion_reg_golden.npz), separately and combined, returned and accumulated."""
    from rodygs_amd.motion_losses import fused_motion_l1_sparsity
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "motion_reg_golden.npz"))
    coeff = torch.from_numpy(g["coeff"]).to(DEV)
    for w1, w2 in ((1.0, 0.0), (0.0, 1.0), (0.01, 0.002)):
        c = coeff.clone().requires_grad_(True)
        loss = fused_motion_l1_sparsity(c, w1, w2)
        loss.backward()
        want = w1 * float(g["l1.loss"]) + w2 * float(g["sparsity.loss"])
        assert abs(float(loss) - want) <= 2e-6 * abs(want) + 1e-9
        wg = w1 * torch.from_numpy(g["l1.d_coeff"]) + w2 * torch.from_numpy(g["sparsity.d_coeff"])
        rel_ok(c.grad, wg, tol=1e-5, what=f"motion reg d_coeff ({w1}, {w2})")
        sink = torch.zeros_like(coeff)
        for rep in (1, 2):                                   # the sink ACCUMULATES
            c2 = coeff.clone().requires_grad_(True)
            (3.0 * fused_motion_l1_sparsity(c2, w1, w2, grad_sink=sink)).backward()
            assert c2.grad is None
            rel_ok(sink, 3.0 * rep * wg, tol=2e-5, what="motion reg sink")


def test_fused_basis_regulariser_matches_reference_golden():
    """rdg_basis_reg (value + gradient of MotionBasisRegularizaiton in three launches) against the imported reference's
    value and d_table for degree 0 ("vanilla" weights) and degree 1 with the "gaussian" frequency weights, and against
    the torch host mirror for degree 2 / a disabled term."""
    from rodygs_amd import motion_losses as ML
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "motion_reg_golden.npz"))

    def run(mod, table):
        class M:
            @staticmethod
            def get_total_motion_table():
                return table
        return mod(M)

    cases = {"basis_d0": ML.MotionBasisRegularizaiton(transl_degree=0),
             "basis_d1_gauss": ML.MotionBasisRegularizaiton(transl_degree=1, rot_degree=1, freq_div_mode="gaussian")}
    for name, mod in cases.items():
        t = torch.from_numpy(g["table"]).to(DEV).requires_grad_(True)
        v = run(mod, t)
        (2.0 * v).backward()
        assert abs(float(v) - float(g[name + ".loss"])) <= 5e-6 * abs(float(g[name + ".loss"]))
        rel_ok(t.grad, 2.0 * torch.from_numpy(g[name + ".d_table"]), tol=2e-5, what=name + " d_table")
    for mod in (ML.MotionBasisRegularizaiton(transl_degree=2, rot_degree=2, freq_div_mode="sigmoid"),
                ML.MotionBasisRegularizaiton(transl_degree=-1, rot_degree=0)):
        tg = torch.from_numpy(g["table"]).to(DEV).requires_grad_(True)
        tc = torch.from_numpy(g["table"]).requires_grad_(True)
        vg, vc = run(mod, tg), run(mod, tc)
        vg.backward(); vc.backward()
        assert abs(float(vg) - float(vc)) <= 5e-6 * abs(float(vc))
        rel_ok(tg.grad, tc.grad, tol=5e-5, what="basis reg vs host mirror")


def test_plain_full_loss_fast_path_equals_general_path():
    """A config-5 step without rigidity runs on the photometric step's fused kernels (fused getter with overwriting
    sinks, per-Gaussian regularisers added after the main backward); its gradients must equal those of the general
    accumulate-everything path from the same state and the same random boxes."""
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(9001, 320, 256, 3, seed=95)
    ds = DynamicScene(sc, num_frames=6, device=DEV, full_losses=True)
    ds.make_ground_truth(O.synthetic_scene(2500, 320, 256, 3, seed=96), range(6))
    ds.train_step(1, perm=[1])
    ds.train_step(5, perm=[2])                                  # one rigidity step, general path
    torch.manual_seed(7)
    la = ds._full_loss(6, 3)
    la.backward()
    ga, gsa = ds.fp.flat_grad.clone(), ds.sp.flat_grad.clone()
    ds.fp.flat_grad.fill_(123.0)                                # the fast path must overwrite everything it owns
    torch.manual_seed(7)
    lb, after = ds._full_loss_plain(3, False)
    lb.backward()
    lb = lb.detach() + after()
    assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(la))
    for k in ds.fp.names:
        o, m = ds.fp.offsets[k]
        rel_ok(ds.fp.flat_grad[o:o + m], ga[o:o + m], tol=2e-4, what="plain full-loss d_" + k)
    rel_ok(ds.sp.flat_grad, gsa, tol=2e-4, what="plain full-loss small bucket")
    hist = [float(ds.train_step(s_, perm=list(range(6)))) for s_ in range(6, 30)]
    assert all(np.isfinite(hist))


def test_sharded_step_full_size_1m_1080p_8_ranks():
    """The bench workload itself (1 M dynamic Gaussians, 1920x1080, 100 frames) as 8 virtual ranks: every camera's loss
    equals the replicated render's bit for bit (modulo the loss reduction's float atomics), and the summed gradients
    agree with the replicated formulation's."""
    from rodygs_amd.losses import fused_photometric_loss
    from rodygs_amd import sharded as S
    from rodygs_amd.trainstep import DynamicScene
    P, world, frames = 1000000, 8, 100
    sc = O.synthetic_scene(P, 1920, 1080, 3, seed=777)
    ds = DynamicScene(sc, num_frames=frames, device=DEV)
    perm = list(range(0, frames, 12))[:8]
    ds.make_ground_truth(O.synthetic_scene(P // 4, 1920, 1080, 3, seed=1234), perm)
    ds.train_step(0, perm=[perm[1]])
    shards = [S.ShardedDynamicScene.from_replica(ds, r, world, exchange=object()) for r in range(world)]
    step = 1
    acc, ref_losses = torch.zeros_like(ds.fp.flat_grad), []
    for r in range(world):
        f = perm[(step * world + r) % len(perm)]
        out, _ = ds.render(f)
        loss = fused_photometric_loss(out[0], ds.gt[f], 0.2)
        loss.backward()
        ref_losses.append(float(loss.detach()))
        acc += ds.fp.flat_grad
    real_update = S.ShardedDynamicScene.phase_update
    S.ShardedDynamicScene.phase_update = lambda self: None
    try:
        losses = [float(x) for x in S.run_virtual_step(shards, step, perm)]
    finally:
        S.ShardedDynamicScene.phase_update = real_update
    assert np.allclose(losses, ref_losses, rtol=1e-6, atol=0)
    assert shards[0].stride == 125184 and shards[0].rows == 8 * 125184 and shards[0].visible_count() > 500000
    for k in ("xyz", "opacity", "motion_coeff", "features"):
        o, m = ds.fp.offsets[k]
        rel_ok(torch.cat([sh.fp[k].grad for sh in shards]), acc[o:o + m].view(ds.fp.shapes[k]), tol=3e-4,
               what="full-size sharded d_" + k)


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_row_order_does_not_change_the_image():
    """Z-curve ordering of the cloud (rodygs_amd/layout.py) is a memory-layout choice: the rendered image and the
    gradients (un-permuted) must equal those of the generator's order -- only summation order differs."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    from rodygs_amd.layout import morton_order
    P = 30000
    sc = O.synthetic_scene(P, 480, 272, 3, seed=19)
    perm = morton_order(sc["means3D"])
    assert sorted(perm.tolist()) == list(range(P))
    rs = HS.make_settings(sc, 3, bg=torch.tensor([0.3, 0.1, 0.2]))
    w = torch.rand(3, 272, 480, generator=torch.Generator().manual_seed(2)).to(DEV)
    outs = []
    for order in (None, perm):
        ins = {}
        for k in NAMES:
            t = sc[k] if (order is None or k == "viewmatrix") else sc[k][order]
            ins[k] = t.clone().contiguous().to(DEV).requires_grad_(True)
        m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                     scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
        (out[0] * w).sum().backward()
        outs.append((out, ins))
    (oa, ia), (ob, ib) = outs
    rel_ok(ob[0], oa[0], tol=2e-5, outliers=OUTLIER_FRAC, what="image under row permutation")
    assert torch.equal(ob[4].cpu(), oa[4].cpu()[perm])
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        rel_ok(ib[k].grad, ia[k].grad[perm.to(DEV)], tol=1e-4, outliers=OUTLIER_FRAC, what="d_" + k + " under row permutation")
    rel_ok(ib["viewmatrix"].grad, ia["viewmatrix"].grad, tol=1e-4, what="d_viewmatrix under row permutation")


def test_bin_mode_follows_the_largest_tile_of_the_previous_frame():
    """A frame whose largest tile list exceeds BIN_RADIX_ABOVE makes the NEXT forward of that (P, H, W) take the radix
    path (its time does not depend on how instances are spread over tiles); same image bit for bit either way, and the
    hint falls back once the lists are short again."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer, rasterizer
    W, H = 320, 240
    sc = O.skewed_scene(W, H, [(5, 6, 40000), (14, 3, 3000)], background=2000, sh_degree_max=3, seed=78)
    P = sc["means3D"].shape[0]
    rs = HS.make_settings(sc, 3)
    ins = {k: sc[k].to(DEV) for k in NAMES}
    key = (P, H, W)
    rasterizer._BIN_HINT.pop(key, None)
    rasterizer._SPLIT_HINT.pop(key, None)
    # the 40 k list would also switch the COMPOSITING of the next frame to the split path (same image to the last bits
    # only: test_split_compositing_of_long_tile_lists); held off here, this test is about the binning algorithm
    monkey = rasterizer.SPLIT_ABOVE
    rasterizer.SPLIT_ABOVE = 10 ** 9

    def fwd(src=ins):
        with torch.no_grad():
            return GaussianRasterizer(rs)(means3D=src["means3D"], means2D=torch.zeros(P, 3, device=DEV), shs=src["shs"],
                                          opacities=src["opacities"], scales=src["scales"], rotations=src["rotations"],
                                          viewmatrix=src["viewmatrix"])
    try:
        a = fwd()                                   # bucket binning (multi-workgroup merge for the 40 k tile)
        assert rasterizer._BIN_HINT.get(key) == 1 and key not in rasterizer._SPLIT_HINT
        b = fwd()                                   # radix path, chosen by the hint
        for i_ in (0, 1, 3):
            assert torch.equal(a[i_], b[i_])
        assert torch.equal(a[4], b[4])
        # lists short again (opacity irrelevant: move the cluster behind the camera) -> back to bucket binning
        far = dict(ins)
        m3 = ins["means3D"].clone()
        m3[:, 2] = torch.where(ins["opacities"][:, 0] < 0.035, -torch.ones_like(m3[:, 2]), m3[:, 2])
        far["means3D"] = m3
        fwd(far)
        assert key not in rasterizer._BIN_HINT
    finally:
        rasterizer.SPLIT_ABOVE = monkey
    # with the threshold back, the 40 k list switches the next frame's compositing to the split path
    fwd()
    assert rasterizer._SPLIT_HINT.get(key) == 1
    c = fwd()
    rasterizer._SPLIT_HINT.pop(key, None)
    rasterizer._BIN_HINT.pop(key, None)
    for i_ in (0, 1, 3):
        rel_ok(c[i_], a[i_], tol=2e-5, outliers=OUTLIER_FRAC, what="split path vs one workgroup per tile")


def test_bench_runs_both_dp_formulations_the_way_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2` in fresh child processes
    (two ranks sharing this one GPU: RDG_ONE_DEVICE=1, collectives through gloo): the default --dp-mode times BOTH
    formulations -- the north_star's replicated cloud + all-reduce and the Gaussian-sharded step -- prints one JSON
    line whose `value` is the faster one, names it in config.parallelism, and lists both under dp_modes."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, RDG_ONE_DEVICE="1", RDG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--points", "20000", "--width", "320", "--height", "240", "--frames", "8", "--gt-frames", "4"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert set(d["dp_modes"]) == {"allreduce", "shard"}
    best = max(d["dp_modes"].items(), key=lambda kv: kv[1]["value"])
    assert abs(best[1]["value"] - d["value"]) < 1e-9 * d["value"]
    assert ("Gaussian-sharded" in d["config"]["parallelism"]) == (best[0] == "shard")
    assert ("all-reduce" in d["config"]["parallelism"]) == (best[0] == "allreduce")
    assert "not a BASELINE config" in d["config"]["workload"]
    assert all(v["hbm_frac"] is None or 0 < v["hbm_frac"] <= 1.0 for v in d["stage_roofline"].values())


def test_plain_bench_command_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 --steps 3` with NO launcher around it (how the driver's N = 1 command line looks with a
    larger N): the process starts the two ranks itself (bench.launch_ranks: a torchrun child, no exec, the parent never
    touches the GPU), relays rank 0's JSON line -- n_gpus == 2, both formulations under dp_modes -- and returns the
    children's status.  Without RDG_ONE_DEVICE the same command on this 1-GPU box must refuse (exit 2), not run one GPU
    and call it two."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--steps", "3", "--warmup", "1", "--points", "20000", "--width", "320", "--height", "240", "--frames", "8",
             "--gt-frames", "4"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RDG_ONE_DEVICE="1", RDG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", *small], env=env, cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and set(d["dp_modes"]) == {"allreduce", "shard"}
    assert d["process_group"]["world_size"] == 2 and d["config"]["one_device"] is True
    if torch.cuda.device_count() < 2:
        env.pop("RDG_ONE_DEVICE")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", *small], env=env, cwd=root,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "needs 2 visible devices" in r.stderr and not r.stdout.strip()


def test_psnr_delta_through_the_real_train_step():
    """BASELINE metric "PSNR delta vs ref" (north_star: within 0.05 dB), at 20 k dynamic Gaussians, 320x240, 500
    optimiser steps with one densification (scripts/psnr_delta.py; PSNR per /root/reference/src/utils/eval_utils.py:36-39).
    Free-running trainings separate chaotically (Adam, eps 1e-15: two runs of one float-atomic binary end 0.1-0.2 dB
    apart), so the statement is split into what can be decided:
      * systematic part -- teacher-forced: at every state of the ORACLE's training the HIP gradient is evaluated too and
        the next states the two gradients lead to are scored; the accumulated difference is deterministic and must be
        within 0.05 dB (measured: ~1e-5 dB);
      * the train step bench.py times, free-running in DETERMINISTIC mode (one reproducible trajectory): within 0.01 dB
        of the oracle at step 100 and 0.1 dB at the densification (all HIP runs within 0.12 dB of each other there),
        SH-Adam-in-backward == separate launch bit for bit;
      * chaotic part -- 4 float-atomic runs: same early gates per run; at the end the oracle (one more draw of the same
        process) within 0.05 dB + 2 standard errors of their mean, the spread taken as at least 0.1 dB."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "psnr_delta", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "psnr_delta.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    res = M.run(points=20000, width=320, height=240, steps=500, frames=8, atomic_runs=4)
    sm = res["summary"]
    for k in sm["hip_runs"] + ["det_fused", "det_unfused", "oracle"]:
        assert res[k]["psnr_end_db"] > res[k]["psnr_start_db"] + 5.0, (k, res[k])     # it really trained
        assert res[k]["densify"]["cloned"] + res[k]["densify"]["split"] > 0, (k, res[k])
        assert res[k]["P_end"] == res["oracle"]["P_end"]
    tf = res["teacher_forced"]
    assert tf["steps"] == 500 and abs(tf["drift_db"]) <= 0.05, tf
    assert tf["abs_sum_db"] <= 0.05, tf                  # even with every per-step difference taken with one sign
    assert sm["det_fused_equals_unfused"], sm
    for k, d in res["delta_db_at_step"]["100"].items():
        assert abs(d) <= 0.01, (k, d, res["delta_db_at_step"])
    at250 = res["delta_db_at_step"]["250"]
    for k, d in at250.items():
        # 250 steps in, the trajectories have begun to separate -- and the ORACLE's own trajectory is not the same on
        # every host (its CPU reductions follow the thread count): on one GPU box every HIP run, the deterministic one
        # included, sat at +0.040 ... +0.057 dB here, on the others at -0.003 +- 0.006.  So per run only the coarse bar,
        # and the HIP runs -- which share everything but the float-atomic order -- must agree among themselves
        assert abs(d) <= 0.1, (k, d, res["delta_db_at_step"])
    assert max(at250.values()) - min(at250.values()) <= 0.12, at250          # measured: up to 0.062 (seed 11, 8 atomic runs)
    # End of the free runs: chaos on BOTH sides.  The oracle's end value is one draw of the same process and moves with the
    # host (24.68 ... 24.80 dB on four boxes of the pool for this seed: its CPU reductions follow the thread count); four
    # runs give a poor estimate of the spread (0.045 ... 0.15 dB seen), so the standard error uses at least the 0.1 dB the
    # process is known to have.  This is a sanity bar for gross errors -- the 0.05 dB statement is the teacher-forced gate.
    n_runs = sm["n_atomic_runs"]
    se = max(sm["hip_std_end_db"], 0.1) * (1.0 + 1.0 / n_runs) ** 0.5
    assert abs(sm["mean_delta_db"]) <= 0.05 + 2.0 * se, sm
    assert sm["hip_std_end_db"] < 0.3, sm


def test_teacher_forced_psnr_drift_detects_a_one_percent_gradient_error():
    """Power of the teacher-forced gate: scaling ONE HIP gradient (dL/dopacity) by 1.01 must show up as a drift orders of
    magnitude above the unbiased kernels' (same 60 steps)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "psnr_delta", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "psnr_delta.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    clean = M.teacher_forced(points=5000, width=160, height=120, steps=60, frames=4)
    biased = M.teacher_forced(points=5000, width=160, height=120, steps=60, frames=4, bias=0.01)
    assert abs(clean["drift_db"]) < 1e-4, clean
    assert abs(biased["drift_db"]) > 50 * max(abs(clean["drift_db"]), 1e-7), (clean["drift_db"], biased["drift_db"])
