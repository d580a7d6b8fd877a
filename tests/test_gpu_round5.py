"""-m gpu tests added in round 5: the densify path against the reference's own densify-and-prune (golden G13), the named
residue of the round-4 anisotropic sweeps, the train loop with periodic densification through the graph path."""
import numpy as np
import pytest
import torch

from oracle import rasterizer_oracle as O  # noqa: F401
from test_gpu_parity import DEV, rel_ok
from test_oracle_golden import densify_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_densify_and_prune_matches_the_reference_trainers_own_run(tag):
    """rodygs_amd.densify.densify_and_prune (one composed source-row list, rdg_gather_rows per buffer, rdg_split_children)
    against G13 -- the output of DynTrainer.densify_and_prune RUN by tests/golden/make_golden.py on the reference's own model
    and optimizer (/root/reference/src/trainer/rodygs_static.py:170-315, rodygs_dynamic.py:150-197, utils.py:36-95): same
    rows in the same order, gathered rows and both moments bit-identical, the split children's positions and scales within
    2e-6 (device exp / log), statistics zeroed, time arrays following their Gaussians, step counter kept."""
    from rodygs_amd.densify import DensifyStats, densify_and_prune
    from rodygs_amd.dp import FlatParams
    ins, outs, a, z, steps = densify_case(tag)
    P = ins["params"]["xyz"].shape[0]
    spec = {k: (tuple(v.shape), 1e-3) for k, v in ins["params"].items()}
    fp = FlatParams(spec, DEV)
    with torch.no_grad():
        for k in fp.names:
            o, n = fp.offsets[k]
            fp[k].copy_(ins["params"][k])
            fp.exp_avg[o:o + n].copy_(ins["exp_avg"][k].reshape(-1))
            fp.exp_avg_sq[o:o + n].copy_(ins["exp_avg_sq"][k].reshape(-1))
    fp.step_count = 5
    stats = DensifyStats(ins["accum"].clone().to(DEV), ins["denom"].clone().to(DEV), ins["max_radii"].clone().to(DEV))
    res = densify_and_prune(fp, stats, {k: v.to(DEV) for k, v in ins["per_point"].items()}, a["max_grad"], a["min_opacity"],
                            a["extent"], a["max_screen_size"], a["percent_dense"], a["N"], z=z.to(DEV))
    Pn = outs["params"]["xyz"].shape[0]
    assert res.fp.shapes["xyz"][0] == Pn and res.fp.step_count == 5 and res.n_split * a["N"] == z.shape[0]
    assert res.n_pruned == P + res.n_clone + a["N"] * res.n_split - Pn
    for k in fp.names:
        o, n = res.fp.offsets[k]
        got = res.fp[k].detach().cpu()
        if k in ("xyz", "scaling"):
            rel_ok(got, outs["params"][k], tol=2e-6, what="densify " + k)
            # survivors are gathered rows (they keep their moments: non-zero here): exact
            kept = outs["exp_avg"]["xyz"].abs().sum(dim=1) != 0
            assert int(kept.sum()) > 50 and torch.equal(got[kept], outs["params"][k][kept]), k
        else:
            assert torch.equal(got, outs["params"][k]), k
        assert torch.equal(res.fp.exp_avg[o:o + n].cpu().view_as(outs["exp_avg"][k]), outs["exp_avg"][k]), k
        assert torch.equal(res.fp.exp_avg_sq[o:o + n].cpu().view_as(outs["exp_avg_sq"][k]), outs["exp_avg_sq"][k]), k
    for k, v in outs["per_point"].items():
        assert torch.equal(res.per_point[k].cpu(), v), k
    assert torch.equal(res.stats.xyz_gradient_accum.cpu(), outs["accum"]) and torch.equal(res.stats.denom.cpu(), outs["denom"])
    assert torch.equal(res.stats.max_radii2D.cpu(), outs["max_radii"])


# ---- the residue of round 4's strict anisotropic sweeps, named (profiles/r04_parity_sweep.txt: the 3 FAIL of 400) ----------------

@pytest.mark.parametrize("mode", ["radix+deterministic", "bucket+atomic"])
@pytest.mark.parametrize("case,expect", [(58, {"d_scales": "geom"}), (223, {"d_means3D": "geom", "d_means2D": "geom"}),
                                         (153, {})])
def test_the_three_sweep_misses_of_round_4_against_their_arbiters(case, expect, mode):
    """410000 / 58, 153, 223 of the anisotropic profile (pancakes and needles, tests/sweep_cases.py), radix binning +
    deterministic backward -- round 4's sweep listed them as misses of the 1e-4 bar against the float64 oracle.  What they are
    (tests/resolution.py, scripts/dbg_scale_grad.py):
      * 58 (a scale-gradient column 400x below its neighbours, HIP 5e-4) and 223 (ONE Gaussian, a needle hundreds of pixels long,
        dL/dmean 1.05e-4): the exact derivative AT THE FLOAT32 GEOMETRY -- the per-Gaussian forward both implementations share bit
        for bit -- is itself 4.4e-4 / 1.1e-4 from the all-float64 one (conic = adj / det with det = a c - b^2 in float32), and HIP
        is within 1e-4 of THAT: class "geom", asserted here explicitly;
      * 153 (ONE Gaussian, a 400-pixel needle): every image and gradient of BOTH float32 implementations is 2-5e-4 from the
        float64 oracle; HIP must stay within 4x the float32 oracle's own distance (class "f32") or the float32 geometry.
    No column may be a plain "fail", in either binning / backward mode."""
    import resolution
    from rodygs_amd import rasterizer as R
    from sweep_cases import sweep_case_aniso
    from test_gpu_parity import run_pair
    sc, deg, bg, kw = sweep_case_aniso(410000, case)
    keep = (R._FORCE_RADIX, R.DETERMINISTIC)
    R._FORCE_RADIX, R.DETERMINISTIC = (mode == "radix+deterministic"), (mode == "radix+deterministic")
    try:
        res = run_pair(sc, deg, bg, **kw)
    finally:
        R._FORCE_RADIX, R.DETERMINISTIC = keep
    assert torch.equal(res[2][4].cpu(), res[5][4]), "radii"
    verdict, txt = resolution.classify(sc, deg, bg, kw, res)
    assert verdict != "fail", txt
    for col, cls in expect.items():
        hit = [ln for ln in txt.split("; [") if col + " column" in ln]
        assert all(ln.startswith(cls) or ln.startswith("[" + cls) for ln in hit), (col, cls, txt)
