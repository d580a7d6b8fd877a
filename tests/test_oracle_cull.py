"""CPU tests of the tight-rectangle rule (RdgRasterSettings.cull = 1 / OracleSettings.cull) against the reference rule, inside
the oracle: the culled tile lists are subsequences of the reference lists, they keep EVERY instance that blends in some
pixel of its tile, and image / final_T / gradients do not move.  The HIP path is held to the oracle's culled key stream
bit for bit by the -m gpu tests; this file is what makes that stream a legitimate stand-in for the reference's."""
import math

import numpy as np
import pytest
import torch

from oracle import rasterizer_oracle as O

ALPHA_MIN = 1.0 / 255.0


def _settings(sc, deg, cull, bg=(0.0, 0.0, 0.0)):
    return O.OracleSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], torch.tensor(bg), 1.0, sc["projmatrix"], deg,
                            cull=cull)


def _geom(sc, deg, cull):
    P = sc["means3D"].shape[0]
    with torch.no_grad():
        g = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], _settings(sc, deg, cull),
                         shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    return g, O.bin_and_sort(g)


def _needles(sc, frac=0.3, seed=5):
    """make a share of the cloud strongly anisotropic and another share nearly transparent"""
    g = torch.Generator().manual_seed(seed)
    P = sc["means3D"].shape[0]
    pick = torch.rand(P, generator=g) < frac
    sc = dict(sc)
    s = sc["scales"].clone()
    s[pick, 0] *= 12.0
    s[pick, 1] *= 0.15
    sc["scales"] = s
    o = sc["opacities"].clone()
    faint = torch.rand(P, generator=g) < 0.2
    o[faint] = 0.002 + 0.01 * torch.rand(int(faint.sum()), 1, generator=g)        # around and below 1/255
    sc["opacities"] = o
    return sc


def _blending_instances(g, b):
    """bool per sorted instance of the REFERENCE lists: some pixel of the tile has power <= 0 and alpha >= 1/255 (the
    oracle's own blend test, `_composite_group`, before any transmittance stop), evaluated in float64 at the float32
    geometry with the bar lowered by 1e-6 relative -- a superset of what any float32 evaluation blends."""
    gx, gy = g["grid"]
    vals, ranges = b["vals_sorted"].astype(np.int64), b["ranges"].astype(np.int64)
    px, py = g["px"].double().numpy(), g["py"].double().numpy()
    con, op = g["conic"].double().numpy(), g["opacity"].double().numpy()
    out = np.zeros(len(vals), dtype=bool)
    ly, lx = np.divmod(np.arange(256), 16)
    for t in range(gx * gy):
        s, e = ranges[t]
        if e <= s:
            continue
        ids = vals[s:e]
        X = (t % gx) * 16 + lx[None, :]
        Y = (t // gx) * 16 + ly[None, :]
        dx, dy = px[ids, None] - X, py[ids, None] - Y
        power = -0.5 * (con[ids, 0:1] * dx * dx + con[ids, 2:3] * dy * dy) - con[ids, 1:2] * dx * dy
        alpha = np.minimum(0.99, op[ids, None] * np.exp(np.minimum(power, 0.0)))
        out[s:e] = ((power <= 1e-9) & (alpha >= ALPHA_MIN * (1 - 1e-6))).any(axis=1)
    return out


def test_ln_from_exact_operations():
    x = np.concatenate([np.float32(0.99) + np.linspace(0, 300, 20001, dtype=np.float32),
                        np.array([0.99, 1.0, 1.4142135, 1.4142137, 2.0, 255.0, 1e-3, 1e6], dtype=np.float32)])
    got = O._ln_f32(x)
    assert got.dtype == np.float32
    assert np.abs(got.astype(np.float64) - np.log(x.astype(np.float64))).max() < 1e-6


@pytest.mark.parametrize("P,W,H,deg,needles", [(1000, 256, 256, 0, False), (6000, 320, 240, 3, False),
                                                (4000, 333, 211, 2, True), (3000, 640, 360, 1, True)])
def test_culled_lists_are_subsequences_that_keep_every_blending_instance(P, W, H, deg, needles):
    sc = O.synthetic_scene(P, W, H, 3, seed=11 + P % 13)
    if needles:
        sc = _needles(sc)
    g0, b0 = _geom(sc, deg, False)
    g1, b1 = _geom(sc, deg, True)
    # what the caller sees of the per-Gaussian stage does not move
    assert torch.equal(g0["radii"], g1["radii"]) and torch.equal(g0["valid"], g1["valid"])
    for k in ("px", "py", "conic", "depth", "rgb"):
        assert torch.equal(g0[k], g1[k])
    assert bool((g1["tiles_touched"] <= g0["tiles_touched"]).all())
    x0, y0, x1, y1 = [r.numpy() for r in g0["rect"]]
    u0, v0, u1, v1 = [r.numpy() for r in g1["rect"]]
    live = g1["tiles_touched"].numpy() > 0
    assert (u0[live] >= x0[live]).all() and (v0[live] >= y0[live]).all()
    assert (u1[live] <= x1[live]).all() and (v1[live] <= y1[live]).all()
    # per tile: the culled list is the reference list with instances REMOVED (order kept) ...
    blend = _blending_instances(g0, b0)
    gx, gy = g0["grid"]
    kept_total = 0
    for t in range(gx * gy):
        s0, e0 = b0["ranges"][t].astype(np.int64)
        s1, e1 = b1["ranges"][t].astype(np.int64)
        ref, cul = b0["vals_sorted"][s0:e0], b1["vals_sorted"][s1:e1]
        keep = np.isin(ref, cul)
        assert np.array_equal(ref[keep], cul), f"tile {t}: not a subsequence"
        # ... and nothing that blends in some pixel of the tile is among the removed
        assert not (blend[s0:e0] & ~keep).any(), f"tile {t}: a blending instance was culled"
        kept_total += len(cul)
    assert kept_total == b1["num_rendered"] <= b0["num_rendered"]
    # tightness: the rule is worth having (it removes a sizeable share of what can be removed at all)
    dead = int((~blend).sum())
    removed = b0["num_rendered"] - b1["num_rendered"]
    assert removed <= dead
    if dead > 200:
        assert removed >= 0.5 * dead, (removed, dead)


@pytest.mark.parametrize("P,W,H,deg,needles", [(1000, 256, 256, 0, False), (3000, 320, 240, 3, True)])
def test_image_and_gradients_do_not_move(P, W, H, deg, needles):
    sc = O.synthetic_scene(P, W, H, 3, seed=3)
    if needles:
        sc = _needles(sc)
    gen = torch.Generator().manual_seed(1)
    wc, wd = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
    names = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
    res = []
    # float64: the float32 oracle's own summation order changes with the list lengths (1e-5 of a needle's gradient); in
    # float64 the two modes must agree to rounding -- the dropped instances contribute exact zeros
    sc = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sc.items()}
    wc, wd = wc.double(), wd.double()
    for cull in (False, True):
        inp = {k: sc[k].clone().requires_grad_(True) for k in names}
        m2 = torch.zeros(P, 3, requires_grad=True, dtype=torch.float64)
        out = O.rasterize(inp["means3D"], m2, inp["opacities"], inp["viewmatrix"], _settings(sc, deg, cull, (0.2, 0.1, 0.3)),
                          shs=inp["shs"], scales=inp["scales"], rotations=inp["rotations"])
        ((out[0] * wc).sum() + 0.1 * (out[1] * wd).sum() + out[3].sum()).backward()
        res.append((out, inp, m2))
    (o0, i0, m0), (o1, i1, m1) = res
    for k in range(4):
        assert torch.allclose(o0[k], o1[k], rtol=0, atol=1e-12 * float(o0[k].detach().abs().max()) + 1e-300), k
    assert torch.equal(o0[4], o1[4])                                   # radii
    a0, a1 = o0[5], o1[5]
    assert torch.allclose(a0["final_T"], a1["final_T"], rtol=0, atol=1e-13)
    # n_contrib counts list positions: the SAME Gaussian is the last contributor of every pixel
    gx = a0["geom"]["grid"][0]

    def last_gaussian(aux):
        nc = aux["n_contrib"].numpy().astype(np.int64)
        ys, xs = np.nonzero(nc)
        tile = (ys // 16) * gx + xs // 16
        start = aux["binning"]["ranges"][tile, 0].astype(np.int64)
        out = np.full(nc.shape, -1, dtype=np.int64)
        out[ys, xs] = aux["binning"]["vals_sorted"][start + nc[ys, xs] - 1]
        return out
    assert np.array_equal(last_gaussian(a0), last_gaussian(a1))
    for k in names:
        sc_ = float(i0[k].grad.abs().max())
        assert float((i0[k].grad - i1[k].grad).abs().max()) <= 1e-11 * sc_ + 1e-300, k
    assert float((m0.grad - m1.grad).abs().max()) <= 1e-11 * float(m0.grad.abs().max())


def test_a_splat_that_cannot_reach_the_alpha_floor_keeps_its_radius_and_gets_no_tile():
    sc = O.synthetic_scene(500, 128, 128, 3, seed=9)
    o = sc["opacities"].clone()
    o[:250] = 0.0038            # 255 * 0.0038 = 0.969 < 0.99: below 1/255 everywhere
    sc["opacities"] = o
    g0, _ = _geom(sc, 0, False)
    g1, _ = _geom(sc, 0, True)
    seen = g0["radii"][:250] > 0
    assert bool(seen.any())
    assert torch.equal(g0["radii"], g1["radii"])
    assert int(g1["tiles_touched"][:250].sum()) == 0 and int(g0["tiles_touched"][:250].sum()) > 0
