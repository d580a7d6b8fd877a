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
