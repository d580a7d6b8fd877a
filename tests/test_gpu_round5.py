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


# ---- the train loop with periodic densification (bench.py --loop) -----------------------------------------------------------------

def _densify_inputs(P=6000, seed=91):
    from rodygs_amd.densify import DensifyStats
    from rodygs_amd.dp import FlatParams
    g = torch.Generator().manual_seed(seed)
    spec = {"xyz": ((P, 3), 1e-3), "features": ((P, 16, 3), 1e-3), "scaling": ((P, 3), 1e-3), "rotation": ((P, 4), 1e-3),
            "opacity": ((P, 1), 1e-3), "motion_coeff": ((P, 1, 16), 1e-3)}
    fp = FlatParams(spec, DEV)
    with torch.no_grad():
        fp.flat.copy_(torch.randn(fp.numel, generator=g))
        fp["scaling"].copy_(torch.log(torch.rand(P, 3, generator=g) * 0.08 + 0.005 + (torch.rand(P, 1, generator=g) > 0.97) * 1.0))
        fp["opacity"].copy_(torch.randn(P, 1, generator=g) * 2.5)
        fp.exp_avg.copy_(torch.randn(fp.numel, generator=g))
        fp.exp_avg_sq.copy_(torch.rand(fp.numel, generator=g))
    fp.step_count = 7
    denom = torch.randint(0, 4, (P, 1), generator=g).float()
    stats = DensifyStats((torch.rand(P, 1, generator=g) * 0.0006 * denom).to(DEV), denom.to(DEV), (torch.rand(P, generator=g) * 40).to(DEV))
    pp = {"time_ind": torch.randint(0, 30, (P,), generator=g).to(DEV)}
    return fp, stats, pp, torch.randn(2 * P, 3, generator=g).to(DEV)


@pytest.mark.parametrize("max_screen_size", [None, 20])
@pytest.mark.parametrize("spatial_order", [False, True])
def test_one_readback_densify_equals_the_general_form(max_screen_size, spatial_order):
    """densify_and_prune as the loop pays for it -- masks, final prune and the five counts formed on the device and read back
    ONCE, the source-row list built without further waits, the Z-curve order composed into the ONE gather per buffer -- against
    the general (decision-replay) form fed the decisions the fast form reports: the same rows, moments, statistics, birth indices
    and counts, bit for bit; a device-tensor threshold equals the float."""
    from rodygs_amd.densify import DensifyStats, densify_and_prune
    fp, stats, pp, z = _densify_inputs()
    args = (0.0002, 0.05, 5.0, max_screen_size, 0.01, 2)
    a = densify_and_prune(fp, stats, pp, *args, z=z, spatial_order=spatial_order)
    st2 = DensifyStats(stats.xyz_gradient_accum.clone(), stats.denom.clone(), stats.max_radii2D.clone())
    b = densify_and_prune(fp, st2, pp, *args, z=z, spatial_order=spatial_order, decisions=a.decisions)       # general path
    c = densify_and_prune(fp, st2, pp, torch.tensor(0.0002, device=DEV), *args[1:], z=z, spatial_order=spatial_order,
                          want_decisions=False)
    assert a.n_clone > 100 and a.n_split > 100 and a.n_pruned > a.n_split and c.decisions is None
    for r in (b, c):
        assert (r.n_clone, r.n_split, r.n_pruned) == (a.n_clone, a.n_split, a.n_pruned) and r.fp.step_count == 7
        assert torch.equal(r.fp.flat, a.fp.flat) and torch.equal(r.fp.exp_avg, a.fp.exp_avg)
        assert torch.equal(r.fp.exp_avg_sq, a.fp.exp_avg_sq) and torch.equal(r.per_point["time_ind"], a.per_point["time_ind"])
        assert float(r.stats.denom.sum()) == 0.0 and r.stats.max_radii2D.shape[0] == a.fp.shapes["xyz"][0]
    if spatial_order:
        from rodygs_amd.layout import morton_codes
        codes = morton_codes(a.fp["xyz"].detach())
        assert bool((codes[1:] >= codes[:-1]).all())


@pytest.mark.parametrize("graph", [False, True])
def test_train_loop_with_periodic_densification(graph):
    """The loop bench.py --loop times: steps with the statistics in the backward kernel, densify_and_prune every 12 steps with a
    device-side quantile threshold (no read-back), the rasterizer's hints carried across the row surgery (the first forward of
    the new cloud runs in deferred mode: no wait for its instance count), and -- graph=True -- a GraphedStep re-captured on
    every new cloud.  The cloud grows, the loss falls, nothing overflows."""
    from rodygs_amd import rasterizer
    from rodygs_amd.trainstep import DynamicScene, GraphedStep
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)
    frames = list(range(8))
    ds = DynamicScene(sc, num_frames=8, device=DEV, spatial_order=True)
    ds.make_ground_truth(tgt, frames)
    ds.track_densification()
    step, losses, traj = 0, [], [ds.P]
    for _ in range(3):
        losses.append(float(ds.train_step(step, perm=frames)))
        step += 1
    keep = rasterizer.DEFERRED_OVERFLOW_CHECK
    rasterizer.DEFERRED_OVERFLOW_CHECK = True
    try:
        for seg in range(4):
            if graph:
                rasterizer.DEFERRED_OVERFLOW_CHECK = False
                ds.raster_state.poll_overflow(block=True)
                gs = GraphedStep(ds, frames, warmup=1, first_step=step)
                step = gs.next_step
                for _ in range(11):
                    losses.append(gs.step())
                gs.check()
                step = gs.next_step
                gs.close()
                losses[-11:] = [float(v) for v in losses[-11:]]
                rasterizer.DEFERRED_OVERFLOW_CHECK = True
            else:
                for _ in range(12):
                    losses.append(float(ds.train_step(step, perm=frames)))
                    step += 1
            assert float(ds.stats.denom.sum()) > 0
            g_mean = (ds.stats.xyz_gradient_accum / ds.stats.denom.clamp_min(1)).reshape(-1)
            key_old = (ds.P, ds.H, ds.W)
            info = ds.densify(max_grad=torch.quantile(g_mean, 0.9), min_opacity=0.01, want_decisions=False)
            traj.append(info["P"])
            assert info["decisions"] is None and info["cloned"] + info["split"] > 0
            assert (ds.P, ds.H, ds.W) in ds.raster_state.capacity_hint and key_old not in ds.raster_state.capacity_hint
            assert ds.time_ind.shape[0] == ds.P == ds.fp.shapes["xyz"][0] and float(ds.stats.denom.sum()) == 0.0
        for _ in range(6):
            losses.append(float(ds.train_step(step, perm=frames)))
            step += 1
        ds.raster_state.poll_overflow(block=True)          # raises if any deferred frame outgrew its carried capacity
    finally:
        rasterizer.DEFERRED_OVERFLOW_CHECK = keep
    assert traj[-1] > traj[0] and len(set(traj)) == len(traj)
    assert all(np.isfinite(losses)) and np.mean(losses[-8:]) < np.mean(losses[:8])


def test_two_raster_states_in_one_process_differ_in_their_modes():
    """The mode switches ride on the RasterState: one state renders through radix binning with the deterministic backward
    (two backward passes: the same bits), another through the defaults, in the same process with the module attributes
    untouched -- same image bit for bit, gradients to float-atomic noise."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    from rodygs_amd import rasterizer as R
    from test_gpu_parity import NAMES
    sc = O.synthetic_scene(6000, 320, 200, 2, seed=12)
    rs = HS.make_settings(sc, 2, bg=torch.tensor([0.1, 0.2, 0.3]))
    gw = torch.rand(3, 200, 320, device=DEV)

    def run(state):
        ins = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
        m2 = torch.zeros(6000, 3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(rs, state=state)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                                  scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
        (out[0] * gw).sum().backward()
        return out[0].detach(), {k: v.grad.clone() for k, v in ins.items()}

    det = R.RasterState(deterministic=True, force_radix=True)
    plain = R.RasterState()
    assert (R.DETERMINISTIC, R._FORCE_RADIX) == (False, False) or pytest.skip("module defaults overridden by the environment")
    i1, g1 = run(det)
    i0, g0 = run(plain)
    i2, g2 = run(det)
    assert det.det_ws is not None and plain.det_ws is None
    assert torch.equal(i1, i2) and torch.equal(i1, i0)
    for k in NAMES:
        assert torch.equal(g1[k], g2[k]), k                       # deterministic state: bit-reproducible
        rel_ok(g0[k], g1[k], tol=2e-5, outliers=1e-4, cap=2e-3, what="atomic vs deterministic d_" + k)


def test_radix_binning_with_a_production_sized_workspace_is_bit_exact():
    """Radix binning with a capacity of 2^22 pairs or more -- what the hosts pass at 1 M Gaussians (4 P + 4096) -- runs the
    4096-pair-tile instantiations of the sort on uint32 keys (below that the 1024-pair ones; tests that size their workspace
    to D + 17 only reach them at 4 M / 4K): keys, order, ranges and D against the oracle at 100 k / 1080p with a 5 M-pair
    workspace; and bucket binning in the same workspace."""
    import hip_stages as HS
    from test_gpu_parity import NAMES, orbit_view
    P, W, H = 100000, 1920, 1080
    sc = O.synthetic_scene(P, W, H, 3, seed=777)
    sc["viewmatrix"] = orbit_view(4.0, -2.0, (0.3, -0.2, 0.5))
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 3)
    with torch.no_grad():
        geom = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                            scales=sc["scales"], rotations=sc["rotations"])
    ref = O.bin_and_sort(geom)
    for bin_mode in (1, 0):
        hs = HS.run_stages(sc, 3, capacity=5000000, bin_mode=bin_mode)
        assert hs["D"] == ref["num_rendered"] < 5000000
        assert np.array_equal(hs["keys_sorted"], ref["keys_sorted"]) and np.array_equal(hs["vals_sorted"], ref["vals_sorted"])
        assert np.array_equal(hs["ranges"], ref["ranges"]), bin_mode


def test_morton_codes_kernel_equals_the_framework_expression():
    """rdg_morton_codes (one launch inside every densification) against layout.morton_codes' float64 torch expression on the
    host: the same codes, so the row order of a cloud does not depend on where it was computed."""
    from rodygs_amd.layout import morton_codes, morton_order
    g = torch.Generator().manual_seed(4)
    for n in (1, 7, 100003):
        x = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 0.2, 40.0]) + torch.tensor([1.0, -5.0, 11.0])
        if n > 7:
            x[::13] = x[0]                                  # equal cells: the order must stay stable
        for bits in (10, 16, 21):
            assert torch.equal(morton_codes(x.to(DEV), bits).cpu(), morton_codes(x, bits)), (n, bits)
        assert torch.equal(morton_order(x.to(DEV)).cpu(), morton_order(x))


def test_reference_iteration_keeps_the_stale_gradient_semantics():
    """trainstep.ReferenceIteration -- static sub-step, then dynamic sub-step, each on the concatenated cloud
    (/root/reference/src/trainer/rodygs.py:157-179, 198-369) -- against the reference's own flow written with framework ops:
    torch.cat of the two clouds' getters (rodygs.py:68-113), the deformation as `coeff @ (B(t) - table[birth])`, the pose as
    FixedCameraTorch.world_view_transform, the same rasterizer and loss, gradients accumulating in plain leaf tensors.  After
    the static sub-step both clouds carry its gradient; the static trainer steps and clears ITS gradients only; after the
    dynamic sub-step the dynamic cloud (and the MLP, and the coefficients) carry the SUM of both frames' gradients, the
    static cloud the second frame's alone.  Densification statistics: each sub-step updates its own slice."""
    import copy
    import torch.nn.functional as F
    from rodygs_amd import GaussianRasterizer
    from rodygs_amd.losses import photometric_loss
    from rodygs_amd.trainstep import ReferenceIteration, world_view_transform
    W, H, T = 208, 144, 6
    ri = ReferenceIteration(O.synthetic_scene(3000, W, H, 3, seed=3), O.synthetic_scene(4000, W, H, 3, seed=4), num_frames=T,
                            device=DEV, spatial_order=True)
    ri.make_ground_truth(O.synthetic_scene(2000, W, H, 3, seed=5), range(T))
    names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")
    leaf = lambda t: t.detach().clone().requires_grad_(True)                                   # noqa: E731
    s = {k: leaf(ri.fp_s[k]) for k in names}
    d = {k: leaf(ri.fp_d[k]) for k in names + ("motion_coeff",)}
    cam = {k: leaf(ri.sp_cam[k]) for k in ("cam_q", "cam_t")}
    net = copy.deepcopy(ri.net)
    for p_ in net.parameters():
        p_.grad = None

    kept = {}

    def flow(frame, static):
        allb = net.motion_basis(ri.emb_rows[frame])
        allb.retain_grad()
        kept["allb"] = allb
        table, bt = allb[:-1], allb[-1]
        delta = (d["motion_coeff"].reshape(-1, 1, 16) @ (bt.unsqueeze(0) - table[ri.time_ind])).squeeze(1)
        dxyz, drot = delta[:, :3] * ri.spatial_lr_scale, delta[:, 3:]
        xyz = torch.cat([s["xyz"], d["xyz"] + dxyz])
        opacity = torch.cat([torch.sigmoid(s["opacity"]), torch.sigmoid(d["opacity"])])
        scaling = torch.cat([torch.exp(s["scaling"]), torch.exp(d["scaling"])])
        rot = torch.cat([F.normalize(s["rotation"]), F.normalize(d["rotation"]) + drot])
        feats = torch.cat([torch.cat([s["f_dc"], s["f_rest"]], 1), torch.cat([d["f_dc"], d["f_rest"]], 1)])
        vm = world_view_transform(cam["cam_q"][frame], cam["cam_t"][frame]).t().contiguous()
        if not static:
            vm = vm.detach()
        m2 = torch.zeros(xyz.shape[0], 3, device=DEV, requires_grad=True)
        out = GaussianRasterizer(ri.settings(static))(means3D=xyz, means2D=m2, shs=feats, opacities=opacity, scales=scaling,
                                                      rotations=rot, viewmatrix=vm)
        photometric_loss(out[0], ri.gt[frame], 0.2).backward()

    def compare(tag):
        for k in names:
            rel_ok(ri.fp_s[k].grad, s[k].grad, tol=2e-4, outliers=1e-4, cap=5e-3, what=f"{tag} static d_{k}")
        for k in names + ("motion_coeff",):
            rel_ok(ri.fp_d[k].grad, d[k].grad, tol=2e-4, outliers=1e-4, cap=5e-3, what=f"{tag} dynamic d_{k}")
        # The MLP is held at its OUTPUT: the gradient of the motion bases [T + 1, 16, 7] (birth-time table, then B(t)).  Its rows
        # sum to zero (dB(t) = -sum_u dB_table[u]), the frame's embedding row equals a table row, and the MLP's parameter
        # gradients are what is left of that cancellation: 1e-5 of the rows' size, so that a 4e-7 relative difference between
        # two float32 evaluations of the bases' gradient reads 5e-2 on them (scripts/probes/ri_mlp_grad_probe.py; the same
        # upstream gradient through both networks gives identical bits).  Accumulated over the sub-steps like the parameters'.
        rel_ok(acc["ri"], acc["flow"], tol=1e-5, what=f"{tag} gradient of the motion bases")
        for k in ("cam_q", "cam_t"):
            g2 = cam[k].grad if cam[k].grad is not None else torch.zeros_like(cam[k])
            rel_ok(ri.sp_cam[k].grad, g2, tol=5e-4, what=f"{tag} d_{k}")

    acc = {}

    def both(frame, which):
        ri.forward_backward(frame, which)                # (retain the bases' gradient of this pass before it runs)
        flow(frame, which == "static")

    real_props = ri.properties

    def props_keep(frame):
        out = real_props(frame)
        ri._last_allb.retain_grad()
        return out

    ri.properties = props_keep
    both(1, "static")
    acc["ri"], acc["flow"] = ri._last_allb.grad.clone(), kept["allb"].grad.clone()
    assert float(ri.fp_d["xyz"].grad.abs().sum()) > 0 and float(ri.sp_cam["cam_q"].grad[1].abs().sum()) > 0
    compare("after the static sub-step:")
    assert float(ri.stats["static"].denom.sum()) > 0 and float(ri.stats["dynamic"].denom.sum()) == 0
    stale = ri.fp_d["xyz"].grad.clone()
    ri.step("static")
    assert float(ri.fp_s.flat_grad.abs().sum()) == 0 and torch.equal(ri.fp_d["xyz"].grad, stale)   # the dynamic gradients stay
    with torch.no_grad():                                        # the flow's static trainer: same step, own gradients cleared
        for k in names:
            s[k].copy_(ri.fp_s[k])
            s[k].grad = None
        for k in cam:
            cam[k].copy_(ri.sp_cam[k])
            cam[k].grad = None
    both(4, "dynamic")
    acc["ri"], acc["flow"] = ri._last_allb.grad.clone(), kept["allb"].grad.clone()
    for k in names:
        s[k].grad = s[k].grad if s[k].grad is not None else torch.zeros_like(s[k])
    compare("after the dynamic sub-step (stale + fresh):")
    assert float((ri.fp_d["xyz"].grad - stale).abs().sum()) > 0 and float(ri.sp_cam["cam_q"].grad.abs().sum()) == 0
    assert float(ri.stats["dynamic"].denom.sum()) > 0
    before = ri.fp_d["xyz"].detach().clone()
    ri.step("dynamic")
    assert float(ri.fp_d.flat_grad.abs().sum()) == 0 and float(ri.fp_s["xyz"].grad.abs().sum()) > 0    # now the static ones stay
    assert not torch.equal(before, ri.fp_d["xyz"].detach())
    l0 = [float(x) for x in ri.iteration(0, list(range(T)))]
    for it in range(1, 12):
        l1 = [float(x) for x in ri.iteration(it, list(range(T)))]
    assert all(np.isfinite(l0 + l1))


def test_graph_branches_switch_gives_the_same_step(monkeypatch):
    """RDG_GRAPH_BRANCHES=1 (opt-in; measured as a loss, profiles/r05_experiments.txt 6): the pose-gradient chain of backward forked
    onto RdgRasterSettings.aux_stream as a branch of the captured graph.  The replayed steps must be the eager steps: the
    pose gradient of the last step agrees to 1e-4 and the loss of the following step to 5e-3 (four Adam steps amplify float-atomic noise)."""
    from rodygs_amd.trainstep import DynamicScene, GraphedStep
    sc = O.synthetic_scene(15000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(4000, 320, 240, 3, seed=6)
    frames = list(range(6))

    def scene():
        ds = DynamicScene(sc, num_frames=6, device=DEV)
        ds.make_ground_truth(tgt, frames)
        for s_ in range(3):
            ds.train_step(s_, perm=frames)
        return ds
    a, b = scene(), scene()
    for s_ in range(3, 7):
        a.train_step(s_, perm=frames)
    monkeypatch.setenv("RDG_GRAPH_BRANCHES", "1")
    gs = GraphedStep(b, frames, warmup=1, first_step=3)
    assert b.raster_state.aux_stream is not None and b.pose_sinks.get("aux") is b.raster_state
    while gs.next_step < 7:
        gs.step()
    gs.check()
    torch.cuda.synchronize()
    rel_ok(b.sp["cam_q"].grad, a.sp["cam_q"].grad, tol=1e-4, what="pose gradient (rotation) of the last replayed step")
    rel_ok(b.sp["cam_t"].grad, a.sp["cam_t"].grad, tol=1e-4, what="pose gradient (translation)")
    # (parameters are not compared entry by entry: Adam with eps 1e-15 turns the float-atomic noise of a near-zero gradient into a
    # full learning-rate step of either sign; what the four steps did to the cloud is compared through the next loss)
    la, lb = float(a.train_step(7, perm=frames)), float(gs.step())
    assert abs(la - lb) <= 5e-3 * abs(la), (la, lb)
    gs.close()
    assert b.raster_state.aux_stream is None and "aux" not in b.pose_sinks
