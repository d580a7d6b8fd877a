"""CPU tests: known-answer checks of the oracle + integer invariants of its binning."""
import math

import numpy as np
import torch

from oracle import rasterizer_oracle as O


def _cam(W, H, fovx_deg=60.0):
    fovx = math.radians(fovx_deg)
    focal = W / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(H / (2 * focal))
    return dict(tanx=math.tan(fovx / 2), tany=math.tan(fovy / 2), focal=focal,
                proj=O.projection_matrix(0.01, 100.0, fovx, fovy).t().contiguous(), view=torch.eye(4))


def _settings(W, H, cam, bg=(0.0, 0.0, 0.0), deg=0):
    return O.OracleSettings(H, W, cam["tanx"], cam["tany"], torch.tensor(bg), 1.0, cam["proj"], deg)


def test_single_isotropic_gaussian_on_axis():
    W = H = 65  # odd: pixel (32,32) is the principal point since px = ((ndc+1)W-1)/2 = 32 at ndc 0
    cam = _cam(W, H)
    z, sigma, o = 4.0, 0.05, 0.7
    m3 = torch.tensor([[0.0, 0.0, z]])
    col = torch.tensor([[0.2, 0.5, 0.9]])
    out = O.rasterize(m3, torch.zeros(1, 3), torch.tensor([[o]]), cam["view"].t().contiguous(), _settings(W, H, cam),
                      colors_precomp=col, scales=torch.full((1, 3), sigma), rotations=torch.tensor([[1.0, 0, 0, 0]]))
    color, depth, normal, alpha, radii, aux = out
    # alpha at the centre pixel = min(0.99, o); colour = c*alpha; depth = z*alpha
    assert abs(alpha[0, 32, 32].item() - o) < 1e-6
    np.testing.assert_allclose(color[:, 32, 32].numpy(), (col[0] * o).numpy(), rtol=1e-6)
    assert abs(depth[0, 32, 32].item() - z * o) < 1e-5
    # variance = f^2 sigma^2 / z^2 + 0.3; isotropic => mid^2 - det = 0 -> floored at 0.1 -> lambda = var + sqrt(0.1)
    var = (cam["focal"] * sigma / z) ** 2 + 0.3
    assert int(radii[0]) == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    # one pixel off-centre: alpha = o * exp(-0.5 / var)
    assert abs(alpha[0, 32, 33].item() - o * math.exp(-0.5 / var)) < 1e-5
    # normal of an isotropic Gaussian = first axis of R, facing the camera (view-space z component <= 0 ... here x axis)
    assert aux["n_contrib"][32, 32] == 1 and abs(aux["final_T"][32, 32].item() - (1 - o)) < 1e-6


def test_alpha_cap_and_background():
    W = H = 33
    cam = _cam(W, H)
    bg = (0.3, 0.6, 0.1)
    out = O.rasterize(torch.tensor([[0.0, 0.0, 3.0]]), torch.zeros(1, 3), torch.tensor([[1.0]]),
                      cam["view"].t().contiguous(), _settings(W, H, cam, bg=bg),
                      colors_precomp=torch.tensor([[1.0, 1.0, 1.0]]), scales=torch.full((1, 3), 0.2),
                      rotations=torch.tensor([[1.0, 0, 0, 0]]))
    color, _, _, alpha, _, _ = out
    assert abs(alpha[0, 16, 16].item() - 0.99) < 1e-6            # capped
    np.testing.assert_allclose(color[:, 16, 16].numpy(), 0.99 + 0.01 * np.array(bg), rtol=1e-5)
    np.testing.assert_allclose(color[:, 0, 0].numpy(), np.array(bg) * (1 - alpha[0, 0, 0].item())
                               + alpha[0, 0, 0].item(), rtol=1e-4, atol=1e-6)


def test_two_gaussians_composite_front_to_back():
    W = H = 33
    cam = _cam(W, H)
    m3 = torch.tensor([[0.0, 0.0, 5.0], [0.0, 0.0, 3.0]])  # index 1 is nearer -> composited first
    col = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    op = torch.tensor([[0.5], [0.6]])
    out = O.rasterize(m3, torch.zeros(2, 3), op, cam["view"].t().contiguous(), _settings(W, H, cam),
                      colors_precomp=col, scales=torch.full((2, 3), 0.3), rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    color, depth, _, alpha, _, aux = out
    np.testing.assert_allclose(color[:, 16, 16].numpy(), [0.4 * 0.5, 0.6, 0.0], rtol=1e-5, atol=1e-7)
    assert abs(depth[0, 16, 16].item() - (3.0 * 0.6 + 5.0 * 0.4 * 0.5)) < 1e-5
    assert abs(alpha[0, 16, 16].item() - (1 - 0.4 * 0.5)) < 1e-6
    b = aux["binning"]
    centre_tile = (16 // 16) * ((W + 15) // 16) + 16 // 16
    s, e = b["ranges"][centre_tile]
    assert list(b["vals_sorted"][s:e]) == [1, 0]                  # depth order inside the tile


def test_culling_and_binning_invariants():
    sc = O.synthetic_scene(3000, 200, 120, 3, seed=11)
    # push some points behind the camera / near plane
    sc["means3D"][:50, 2] = -1.0
    sc["means3D"][50:60, 2] = 0.1
    st = O.OracleSettings(120, 200, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 3)
    with torch.no_grad():
        g = O.preprocess(sc["means3D"], torch.zeros(3000, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                         scales=sc["scales"], rotations=sc["rotations"])
    assert int(g["radii"][:60].abs().sum()) == 0 and int(g["tiles_touched"][:60].sum()) == 0
    b = O.bin_and_sort(g)
    D = b["num_rendered"]
    assert D == int(g["tiles_touched"].sum()) == len(b["keys_sorted"])
    ks = b["keys_sorted"]
    assert np.all(ks[1:] >= ks[:-1])
    gx, gy = g["grid"]
    assert (ks >> np.uint64(32)).max() < gx * gy
    # ranges partition [0, D) in tile order and every key in a range carries that tile id
    tiles = (ks >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles)[:50]:
        s, e = b["ranges"][t]
        assert np.all(tiles[s:e] == t) and (s == 0 or tiles[s - 1] != t) and (e == D or tiles[e] != t)
    assert int((b["ranges"][:, 1] - b["ranges"][:, 0]).sum()) == D
    # stability: equal keys keep emission (Gaussian index) order
    eq = ks[1:] == ks[:-1]
    assert np.all(b["vals_sorted"][1:][eq] > b["vals_sorted"][:-1][eq])


def test_oracle_gradients_against_float64_finite_differences():
    """The oracle's autograd (the spec for the HIP backward) vs central differences in float64, on the smooth
    part of the pipeline (colour + depth of a small scene; thresholds are far from flipping at eps=1e-6)."""
    torch.manual_seed(0)
    sc = O.synthetic_scene(40, 48, 32, 1, seed=3)
    W, H = 48, 32
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor([0.1, 0.2, 0.3]), 1.0,
                          sc["projmatrix"].double(), 1)
    wc = torch.rand(3, H, W, dtype=torch.float64)

    def f(m3, vm, sh, scl, rot, op):
        c, d, _, a, _, _ = O.rasterize(m3, torch.zeros(40, 3, dtype=torch.float64), op, vm, st, shs=sh, scales=scl,
                                       rotations=rot)
        return (c * wc).sum() + 0.1 * d.sum() + 0.3 * a.sum()

    args = [sc[k].double().clone().requires_grad_(True) for k in
            ("means3D", "viewmatrix", "shs", "scales", "rotations", "opacities")]
    loss = f(*args)
    grads = torch.autograd.grad(loss, args)
    eps = 1e-6
    gen = torch.Generator().manual_seed(1)
    for a, g in zip(args, grads):
        if a.shape == (4, 4):
            idxs = [(r, c) for r in range(4) for c in range(3)]  # last glm column (V[3],V[7],..) is unused
            idxs = [(c, r) for (r, c) in idxs][:6]
        else:
            flat = torch.randint(0, a.numel(), (4,), generator=gen).tolist()
            idxs = [tuple(np.unravel_index(i, a.shape)) for i in flat]
        for idx in idxs:
            with torch.no_grad():
                old = a[idx].item()
                a[idx] = old + eps
                lp = f(*args).item()
                a[idx] = old - eps
                lm = f(*args).item()
                a[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g[idx].item()) <= 2e-4 * max(1.0, abs(fd)), (a.shape, idx, fd, g[idx].item())


def test_oracle_extra_attributes_follow_the_colour_weights():
    """The oracle's ``extra_attrs`` extension: attributes equal to the precomputed colours composite to the colour image
    minus its background term, and a constant attribute of 1 composites to the alpha image."""
    import torch
    from oracle import rasterizer_oracle as O
    sc = O.synthetic_scene(400, 64, 48, 3, seed=3)
    bg = torch.tensor([0.3, 0.2, 0.1])
    st = O.OracleSettings(48, 64, sc["tanfovx"], sc["tanfovy"], bg, 1.0, sc["projmatrix"], 0)
    cols = torch.rand(400, 3, generator=torch.Generator().manual_seed(1))
    attrs = torch.cat([cols, torch.ones(400, 1)], dim=1)
    with torch.no_grad():
        o = O.rasterize(sc["means3D"], torch.zeros(400, 3), sc["opacities"], sc["viewmatrix"], st, colors_precomp=cols,
                        scales=sc["scales"], rotations=sc["rotations"], extra_attrs=attrs)
    ex, T = o[5]["extra"], o[5]["final_T"]
    assert ex.shape == (4, 48, 64)
    assert torch.allclose(ex[:3] + T.unsqueeze(0) * bg.view(3, 1, 1), o[0], atol=1e-6)
    assert torch.allclose(ex[3:4], o[3], atol=1e-6)
