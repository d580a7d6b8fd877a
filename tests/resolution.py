"""Who is right when two float32 implementations differ?  The arbiters a miss of the 1e-4 bar is taken to, column by column
(used by scripts/parity_sweep.py for every case that misses the bar against the float32 oracle, and by the named residue tests of
tests/test_gpu_round5.py).  HIP's distance from the oracle run in FLOAT64 decides; a column outside the bar there is

  "f64"   -- not outside at all: inside the bar against the float64 oracle (the float32 oracle was the one that is off);
  "geom"  -- inside the bar against the float64 oracle evaluated AT THE FLOAT32 GEOMETRY: the per-Gaussian forward (pixel
             centre, conic, depth, colour -- the values the bit-exact tests pin, so both implementations share their rounding)
             taken in float32, compositing forward and backward and the per-Gaussian Jacobian in float64.  conic = adj / det
             with det = a c - b^2 in float32 carries 2e-5 of relative error on a needle-shaped footprint (det / (a c) = 3e-3),
             and the derivative of the image with respect to the centre of a 300-pixel needle moves by 1e-4 with it (strict
             sweep 410000 / 223: the exact derivative at the float32 conic is 1.0e-4 from the all-float64 one; HIP delivers the
             former to 1e-6, the float32 oracle lands 1e-5 from the latter because the per-pixel rounding of its naive exponent
             happens to shift the sum back);
  "f32"   -- within 4x the float32 ORACLE's own distance from the float64 oracle: neither float32 implementation resolves it;
  "f32s"  -- within 4x the float32 oracle's LARGEST distance over all gradient columns of the scene, where that is itself outside
             the bar: a scene of one or a few extreme needles (410000 / 153: ONE Gaussian, cov2D (5103, -5800, 6594), det / (a c) =
             1.6e-4, 156 tiles) whose exponent no float32 evaluation resolves -- w = dx + beta dy carries 2^-24 |beta dy| / |w| ~
             1e-5 per pixel, 1e-4 in G --, so every row sum of BOTH implementations is 1e-4 ... 1e-3 off and which column of the
             float32 oracle happens to land close is luck (its rotation gradient 1.4e-4, its scale gradient 8.4e-4, from the same rows);
  "cond"  -- within the change of the float64 oracle's OWN gradient under a relative perturbation of 2^-21 (four float32 ulps: the
             backward error a float32 algorithm of a thousand operations has) of the inputs, largest of four random draws: the
             column is not determined to 1e-4 by float32 inputs at all (a scale gradient that is the null direction of an
             indefinite dL/dcov2D -- 410000 / 58: half an ulp on the inputs moves it by 2.6e-3; 520000 / 2354: a column 1 750x
             below its neighbours, two ulps move it by 1.1e-3, HIP is 1.3e-3 off, the float32 oracle 2.2e-4);
  "fail"  -- none of these.
Images (and the per-pixel final transmittance the backward replays from) can only be "f64", "f32" or "fail"."""
import torch

from oracle import rasterizer_oracle as O
from oracle.parity import columns

NAMES = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
TOL = 1e-4
F32_FACTOR = 4.0
COND_REL = 2.0 ** -21


def _weights(sc, kw):
    H, W = sc["H"], sc["W"]
    gen = torch.Generator().manual_seed(kw["seed"])
    wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
    wn = torch.randn(3, H, W, generator=gen) * kw.get("normal_loss", 0.0)
    return wc, wd, wa, wn


def _loss(color, depth, normal, alpha, w, kw, dt):
    wc, wd, wa, wn = (t.to(dt) for t in w)
    ls = (color * wc).sum() + (alpha * wa).sum()
    if kw.get("depth_loss", 0.1):
        ls = ls + (depth * wd).sum() * kw.get("depth_loss", 0.1)
    if kw.get("normal_loss", 0.0):
        ls = ls + (normal * wn).sum()
    return ls


def _settings(sc, deg, bg, kw, dt):
    return O.OracleSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], torch.tensor(bg).to(dt), kw.get("scale_modifier", 1.0),
                            sc["projmatrix"].to(dt), deg, enable_cov_grad=kw.get("cov_grad", True),
                            enable_sh_grad=kw.get("sh_grad", True))


def float64_run(sc, deg, bg, kw, perturb_seed=None):
    """The oracle in float64 on the case (run_pair's loss): (images, {name: gradient}).  perturb_seed: every input but the SH
    coefficients multiplied by (1 +- COND_REL) with random signs first."""
    P = sc["means3D"].shape[0]
    g = torch.Generator().manual_seed(perturb_seed or 0)
    d = {}
    for k in NAMES:
        v = sc[k].clone().double()
        if perturb_seed is not None and k != "shs":
            v = v * (1 + COND_REL * (torch.randint(0, 2, v.shape, generator=g).double() * 2 - 1))
        d[k] = v.requires_grad_(True)
    m2 = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    st = _settings(sc, deg, bg, kw, torch.float64)
    o = O.rasterize(d["means3D"], m2, d["opacities"], d["viewmatrix"], st, shs=d["shs"], scales=d["scales"],
                    rotations=d["rotations"])
    ls = _loss(o[0], o[1], o[2], o[3], _weights(sc, kw), kw, torch.float64)
    if ls.requires_grad:
        ls.backward()
    grads = {k: (d[k].grad if d[k].grad is not None else torch.zeros_like(d[k])) for k in NAMES}
    grads["means2D"] = m2.grad if m2.grad is not None else torch.zeros_like(m2)
    return [o[i].detach() for i in range(4)] + [o[5]["final_T"].detach()], grads


def float32_geometry_run(sc, deg, bg, kw, return_geom=False):
    """The float64 oracle AT THE FLOAT32 GEOMETRY (class "geom" above): {name: gradient}."""
    P, H, W = sc["means3D"].shape[0], sc["H"], sc["W"]
    keys = ("px", "py", "conic", "opacity", "rgb", "depth")
    with torch.no_grad():
        g32 = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], _settings(sc, deg, bg, kw, torch.float32),
                           shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    d = {k: sc[k].clone().double().requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    st = _settings(sc, deg, bg, kw, torch.float64)
    g64 = O.preprocess(d["means3D"], m2, d["opacities"], d["viewmatrix"], st, shs=d["shs"], scales=d["scales"],
                       rotations=d["rotations"])
    gm = dict(g32)
    for k in keys:
        gm[k] = g32[k].double().detach().requires_grad_(True)
    gm["normal"] = g32["normal"].double()
    img = O.render_tiles(gm, O.bin_and_sort(g32), st.bg, H, W)
    ls = _loss(img["color"], img["depth"], img["normal"], img["alpha"], _weights(sc, kw), kw, torch.float64)
    zero = {k: torch.zeros_like(v) for k, v in {**d, "means2D": m2}.items()}
    if not ls.requires_grad:
        return zero
    ls.backward()
    outs = [g64[k] for k in keys if gm[k].grad is not None and g64[k].requires_grad]
    gouts = [gm[k].grad for k in keys if gm[k].grad is not None and g64[k].requires_grad]
    leaves = [d[k] for k in NAMES] + [m2]
    got = torch.autograd.grad(outs, leaves, grad_outputs=gouts, allow_unused=True)
    grads = {k: (g if g is not None else zero[k]) for k, g in zip(list(NAMES) + ["means2D"], got)}
    if return_geom:       # + dL/d(px, py, conic, opacity, rgb, depth) at the float32 geometry, and that geometry (debug scripts)
        return grads, {k: gm[k].grad for k in keys}, g32
    return grads


def _col_err(a, ref):
    A, R = columns(torch.as_tensor(a).detach().double().cpu()), columns(torch.as_tensor(ref).detach().double().cpu())
    scale = R.abs().amax(1).clamp_min(1e-300)
    return (A - R).abs().amax(1) / scale, scale


def classify(sc, deg, bg, kw, res, cond_draws=4):
    """res = test_gpu_parity.run_pair(sc, deg, bg, **kw).  Returns (verdict, text): the WORST class any column fell into
    (order f64 < geom < f32 < f32s < cond < fail) and one line per column outside the bar against the float64 oracle."""
    hi, hm2, hout, oi, om2, oout = res
    imgs64, g64 = float64_run(sc, deg, bg, kw)
    rank = {"f64": 0, "geom": 1, "f32": 2, "f32s": 3, "cond": 4, "fail": 5}
    verdict, lines = "f64", []

    def worse(v):
        nonlocal verdict
        if rank[v] > rank[verdict]:
            verdict = v

    for idx, name in ((0, "color"), (1, "depth"), (2, "normal"), (3, "alpha"), (4, "final_T")):
        # (final_T: the product of hundreds of (1 - alpha) on a deep list, values of 1e-3: held to the float64 oracle like an image)
        h_img = hout[6][0] if idx == 4 else hout[idx]
        o_img = oout[5]["final_T"] if idx == 4 else oout[idx]
        eh, _ = _col_err(h_img, imgs64[idx])
        eo, _ = _col_err(o_img, imgs64[idx])
        for j in torch.nonzero(eh > TOL).flatten().tolist():
            v = "f32" if float(eh[j]) <= F32_FACTOR * float(eo[j]) else "fail"
            worse(v)
            lines.append(f"{name} channel {j} [{v}]: HIP {float(eh[j]):.2e}, float32 oracle {float(eo[j]):.2e} from the float64 oracle")
    geom, cond = None, None
    eo_scene = 0.0
    for k in list(NAMES) + ["means2D"]:
        eo, _ = _col_err((om2 if k == "means2D" else oi[k]).grad, g64[k])
        eo_scene = max(eo_scene, float(eo.max()))
    for k in list(NAMES) + ["means2D"]:
        h = (hm2 if k == "means2D" else hi[k]).grad
        o32 = (om2 if k == "means2D" else oi[k]).grad
        eh, scale = _col_err(h, g64[k])
        eo, _ = _col_err(o32, g64[k])
        for j in torch.nonzero(eh > TOL).flatten().tolist():
            txt = (f"d_{k} column {j} (scale {float(scale[j]):.2e}, the tensor's largest {float(scale.max()):.2e}): HIP "
                   f"{float(eh[j]):.2e}, float32 oracle {float(eo[j]):.2e} from the float64 oracle")
            if geom is None:
                geom = float32_geometry_run(sc, deg, bg, kw)
            # held to the float32-geometry arbiter on the float64 oracle's column scale
            A, R = columns(h.detach().double().cpu()), columns(geom[k].double())
            eg = float((A[j] - R[j]).abs().max() / scale[j])
            if eg <= TOL:
                v, txt = "geom", txt + f"; {eg:.2e} from the float64 oracle at the float32 geometry"
            elif float(eh[j]) <= F32_FACTOR * float(eo[j]):
                v = "f32"
            elif eo_scene > TOL and float(eh[j]) <= F32_FACTOR * eo_scene:
                v, txt = "f32s", txt + f"; {eg:.2e} at the float32 geometry; the float32 oracle's largest distance in this scene {eo_scene:.2e}"
            else:
                if cond is None:
                    cond = {}
                    for s in range(1, cond_draws + 1):
                        _, gp = float64_run(sc, deg, bg, kw, perturb_seed=s)
                        for kk in gp:
                            e, _ = _col_err(gp[kk], g64[kk])
                            cond[kk] = torch.maximum(cond[kk], e) if kk in cond else e
                cj = float(cond[k][j])
                v = "cond" if float(eh[j]) <= cj else "fail"
                txt += f"; {eg:.2e} at the float32 geometry; inputs perturbed by 2^-21 move the float64 gradient by {cj:.2e}"
            worse(v)
            lines.append(f"[{v}] " + txt)
    return verdict, "; ".join(lines)
