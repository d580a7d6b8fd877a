"""CPU tests of the comparison rules themselves (oracle/parity.py): the full-frame report must pass an implementation
against itself, must flag a one-entry error off the flip candidates, and must grant the allowance only to what a witnessed
pixel flip explains."""
import torch

from oracle import rasterizer_oracle as O
from oracle.parity import column_stats, flip_mask, full_frame_report

NAMES = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")


def _oracle_frame(P=1500, W=96, H=64, deg=2, seed=3):
    sc = O.synthetic_scene(P, W, H, deg, seed=seed)
    ins = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(P, 3, requires_grad=True)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor([0.1, 0.2, 0.3]), 1.0, sc["projmatrix"], deg)
    out = O.rasterize(ins["means3D"], m2, ins["opacities"], ins["viewmatrix"], st, shs=ins["shs"], scales=ins["scales"],
                      rotations=ins["rotations"])
    g = torch.Generator().manual_seed(1)
    (out[0] * torch.rand(3, H, W, generator=g)).sum().add((out[1] * torch.rand(1, H, W, generator=g)).sum()).backward()
    aux = out[5]
    fr = {"images": {"color": out[0].detach(), "depth": out[1].detach(), "alpha": out[3].detach()},
          "final_T": aux["final_T"], "n_contrib": aux["n_contrib"], "radii": out[4], "D": aux["binning"]["num_rendered"],
          "grads": {**{k: ins[k].grad for k in NAMES}, "means2D": m2.grad}}
    return fr, aux["binning"], (W + 15) // 16


def _copy(fr):
    return {"images": {k: v.clone() for k, v in fr["images"].items()}, "final_T": fr["final_T"].clone(),
            "n_contrib": fr["n_contrib"].clone(), "radii": fr["radii"].clone(), "D": fr["D"],
            "grads": {k: v.clone() for k, v in fr["grads"].items()}}


def test_full_frame_report_rules():
    fr, binning, gx = _oracle_frame()
    rep = full_frame_report(_copy(fr), fr, binning["vals_sorted"], binning["ranges"], gx)
    assert rep["ok"] and rep["witnessed_flips"] == 0 and rep["flip_candidate_gaussians"] == 0
    assert max(rep["grad_max_rel_per_tensor"].values()) == 0.0
    # one gradient entry 5e-4 of its column's scale off, no flip anywhere: a violation
    bad = _copy(fr)
    g = bad["grads"]["opacities"]
    row = int(fr["grads"]["opacities"].abs().argmax())
    g[row] += 5e-4 * fr["grads"]["opacities"].abs().max()
    rep = full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx)
    assert not rep["ok"] and any(v.startswith("d_opacities") for v in rep["violations"])
    # the same error on a Gaussian that stands in the list of a pixel with a WITNESSED flip: allowed (below the cap)
    ranges = binning["ranges"].astype("int64")
    t = int((ranges[:, 1] - ranges[:, 0]).argmax())
    gid = int(binning["vals_sorted"][ranges[t, 0]])
    y, x = (t // gx) * 16, (t % gx) * 16
    bad = _copy(fr)
    assert int(bad["n_contrib"][y, x]) >= 1
    bad["n_contrib"][y, x] += 1
    bad["grads"]["opacities"][gid] += 5e-4 * fr["grads"]["opacities"].abs().max()
    rep = full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx)
    assert rep["witnessed_flips"] == 1 and rep["flip_candidate_gaussians"] >= 1 and rep["ok"], rep
    # ... but not beyond the cap, and not for another Gaussian
    bad["grads"]["opacities"][gid] += 0.5 * fr["grads"]["opacities"].abs().max()
    assert not full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx)["ok"]
    # the loss's own kink: the same error is allowed for the Gaussians of a pixel named in loss_kink -- the pixel is not
    bad = _copy(fr)
    bad["grads"]["opacities"][gid] += 5e-4 * fr["grads"]["opacities"].abs().max()
    kink = torch.zeros(fr["final_T"].shape, dtype=torch.bool)
    assert not full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx, loss_kink=kink)["ok"]
    kink[y, x] = True
    rep = full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx, loss_kink=kink)
    assert rep["ok"] and rep["loss_kink_pixels"] == 1 and rep["witnessed_flips"] == 0
    bad["images"]["color"][0, y, x] += 1e-3
    assert not full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx, loss_kink=kink)["ok"]
    # a pixel off the bar without a flip
    bad = _copy(fr)
    bad["images"]["color"][1, 5, 7] += 1e-3
    rep = full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx)
    assert not rep["ok"] and rep["witnessed_flips"] == 0
    # radii / D
    bad = _copy(fr)
    bad["radii"][0] += 1
    bad["D"] += 1
    rep = full_frame_report(bad, fr, binning["vals_sorted"], binning["ranges"], gx)
    assert not rep["radii_equal"] and not rep["D_equal"] and len(rep["violations"]) == 2


def test_flip_mask_and_column_stats():
    fT = torch.tensor([[0.5, 0.25], [1.0, 1e-5]])
    nc = torch.tensor([[3, 4], [0, 9]], dtype=torch.int32)
    assert int(flip_mask(fT, nc, fT, nc).sum()) == 0
    assert int(flip_mask(fT * torch.tensor([[1.0, 1.004], [1.0, 1.0]]), nc, fT, nc).sum()) == 1       # 1 - alpha >= 1/255
    assert int(flip_mask(fT, nc + torch.tensor([[0, 0], [0, 1]], dtype=torch.int32), fT, nc).sum()) == 1
    a = torch.zeros(300, 3)
    b = torch.zeros(300, 3)
    b[:, 0] = torch.linspace(-1, 1, 300)
    a[:, 0] = b[:, 0] + 2e-4
    st = column_stats(a, b)
    assert st["columns"] == 3 and st["entries_over_bar"] == 300 and abs(st["max_rel"] - 2e-4) < 1e-6


def test_bench_cpu_leg_feeds_the_oracle_the_frames_own_bits_and_its_result_passes_the_report():
    """bench.py's cpu_baseline leg: the oracle train step runs on EXACTLY the rasterizer inputs of the frame (the activations
    are differentiated, their values are not used), keeps the whole frame's image / per-pixel state / gradients, and the
    parity entry built from them is the full-frame report (here: the oracle against a second run of itself)."""
    import importlib.util
    import os
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("bench_for_parity", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sc = O.synthetic_scene(800, 80, 48, 3, seed=5)
    gt = torch.rand(3, 48, 80, generator=torch.Generator().manual_seed(1))
    t1, o1 = bench._cpu_train_step(sc, gt, 3, 1e9)
    t2, o2 = bench._cpu_train_step(sc, gt, 3, 1e9)
    assert not t1["extrapolated"] and t1["tiles_done"] == t1["n_tiles"] == 15
    # against the plain oracle call on the same tensors: same radii, D, image
    ins = {k: sc[k].clone() for k in NAMES}
    st = O.OracleSettings(48, 80, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 3)
    with torch.no_grad():
        ref = O.rasterize(ins["means3D"], torch.zeros(800, 3), ins["opacities"], ins["viewmatrix"], st, shs=ins["shs"],
                          scales=ins["scales"], rotations=ins["rotations"])
    assert torch.equal(ref[4], o1["radii"]) and ref[5]["binning"]["num_rendered"] == o1["D"]
    assert torch.equal(ref[0], o1["images"]["color"]) and torch.equal(ref[1], o1["images"]["depth"])
    assert torch.equal(ref[5]["final_T"], o1["final_T"]) and torch.equal(ref[5]["n_contrib"], o1["n_contrib"])
    o1h = {k: o1[k] for k in ("images", "final_T", "n_contrib", "radii", "D", "grads")}
    o1h["loss"] = o1["loss"]
    rep = bench._parity(o1h, o2, gt)
    assert rep["ok"] and rep["witnessed_flips"] == 0 and set(rep["grad_max_rel_per_tensor"]) == set(NAMES) | {"means2D"}
    assert "skipped" in bench._parity(None, o2, gt) and "skipped" in bench._parity(o1h, None, gt)
    n, model = bench._host_cpu()
    assert n == os.cpu_count()


def test_the_frozen_sweep_rules_hash_is_current():
    """tests/test_gpu_round6.py::test_randomised_sweep_under_frozen_rules holds a hash of everything that decides a sweep
    verdict; checked here on the CPU as well, so that an edited rule without the edited constant fails before any GPU run."""
    import importlib
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    sweep_run = importlib.import_module("sweep_run")
    src = open(os.path.join(here, "test_gpu_round6.py")).read()
    assert f'SWEEP_RULES_HASH = "{sweep_run.rules_hash()}"' in src
