"""-m gpu tests added in round 4: the densification statistics inside the native step, the graph replay of the step the
reference really runs (statistics, learning-rate schedule, sticky overflow record), the caller-owned rasterizer state,
and the hardened parity evidence (both binning algorithms x both backward modes, the two sweep cases that missed 1e-4)."""
import threading

import numpy as np
import pytest
import torch

from oracle import rasterizer_oracle as O
from test_gpu_parity import DEV, NAMES, check_pair, orbit_view, rel_ok, run_pair

pytestmark = pytest.mark.gpu


# ---- densification statistics (/root/reference/src/trainer/rodygs.py:316-341, rodygs_static.py:317-319) ------------------

def _reference_stats_update(accum, denom, max_radii, viewspace_grad, radii, row0, rows):
    """The reference's own expression on the slice of the concatenated cloud it keeps for the sub-step."""
    sl = slice(row0, row0 + rows)
    radii_s = radii[sl]
    visibility_filter = (radii > 0)[sl]
    grad_densification = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True)[sl]
    max_radii[visibility_filter] = torch.max(max_radii[visibility_filter], radii_s[visibility_filter].to(max_radii.dtype))
    accum[visibility_filter] += grad_densification[visibility_filter]
    denom[visibility_filter] += 1


@pytest.mark.parametrize("mode", ["atomic", "deterministic"])
@pytest.mark.parametrize("row0,rows", [(0, 6000), (0, 2500), (2500, 3500)])     # whole cloud / static part / dynamic part
def test_densification_statistics_inside_backward_match_the_reference_expression(mode, row0, rows):
    """RdgRasterSettings.densify_* (grad_sinks["densify"]): the per-Gaussian backward kernel updates max_radii2D,
    xyz_gradient_accum and denom itself.  Four frames from different cameras (a Gaussian is visible in some of them only);
    against the reference's boolean-mask expression evaluated on the returned dL/dmeans2D and radii, and against the
    stand-alone launch (rdg_densify_stats).  A second backward through the same graph must not count the frame twice."""
    import hip_stages as HS
    import rodygs_amd.rasterizer as R
    from rodygs_amd import GaussianRasterizer
    from rodygs_amd.densify import DensifyStats
    P, W, H = 6000, 333, 211
    sc = O.synthetic_scene(P, W, H, 3, seed=91)
    fused = DensifyStats.zeros(rows, DEV)
    alone = DensifyStats.zeros(rows, DEV)
    want = [torch.zeros(rows, 1, device=DEV), torch.zeros(rows, 1, device=DEV), torch.zeros(rows, device=DEV)]
    old = R.DETERMINISTIC
    R.DETERMINISTIC = mode == "deterministic"
    try:
        for f, (ay, ax, t) in enumerate([(0.0, 0.0, (0.0, 0.0, 0.0)), (14.0, -6.0, (0.5, -0.2, 0.8)),
                                         (-17.0, 9.0, (-0.6, 0.3, -0.5)), (30.0, 0.0, (2.5, 0.0, 3.0))]):
            ins = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
            ins["viewmatrix"] = orbit_view(ay, ax, t).to(DEV).requires_grad_(True)
            m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
            sink = fused.sink(row0)
            out = GaussianRasterizer(HS.make_settings(sc, 3, bg=torch.tensor([0.1, 0.0, 0.2])))(
                means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"], scales=ins["scales"],
                rotations=ins["rotations"], viewmatrix=ins["viewmatrix"], grad_sinks={"densify": sink})
            w = torch.rand(3, H, W, device=DEV, generator=torch.Generator(DEV).manual_seed(f))
            loss = (out[0] * w).sum() + 0.05 * out[1].sum()
            loss.backward(retain_graph=(f == 1))
            radii = out[4]
            assert 0 < int((radii[row0:row0 + rows] > 0).sum()) < rows          # some visible, some not
            _reference_stats_update(want[0], want[1], want[2], m2.grad, radii, row0, rows)
            alone.add_frame(m2.grad, radii, row0)
            if f == 1:
                m2.grad = None
                loss.backward()            # the reference's retain_graph pattern: no second count for this frame
    finally:
        R.DETERMINISTIC = old
    torch.cuda.synchronize()
    for st, name in ((fused, "inside backward"), (alone, "rdg_densify_stats")):
        assert torch.equal(st.denom, want[1]), name + ": denom"
        assert torch.equal(st.max_radii2D, want[2]), name + ": max_radii2D"
        rel_ok(st.xyz_gradient_accum, want[0], tol=1e-6, what=name + ": xyz_gradient_accum")
    assert float(want[1].max()) >= 3.0 and float(want[1].min()) == 0.0


def test_train_step_statistics_with_the_optimizer_in_backward_and_a_densification():
    """The statistics through DynamicScene.train_step -- the per-Gaussian backward variant that also applies the SH Adam
    step -- against the reference expression on the step's own dL/dmeans2D and radii; then densify on them and go on."""
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)
    ds = DynamicScene(sc, num_frames=8, device=DEV, spatial_order=True)
    ds.make_ground_truth(tgt, range(8))
    ds.track_densification()
    P = ds.P
    want = [torch.zeros(P, 1, device=DEV), torch.zeros(P, 1, device=DEV), torch.zeros(P, device=DEV)]
    for s_ in range(12):
        ds.train_step(s_, perm=list(range(8)))
        _reference_stats_update(want[0], want[1], want[2], ds.m2.grad, ds._last_radii, 0, P)
    assert torch.equal(ds.stats.denom, want[1]) and torch.equal(ds.stats.max_radii2D, want[2])
    rel_ok(ds.stats.xyz_gradient_accum, want[0], tol=1e-6, what="xyz_gradient_accum over 12 steps")
    info = ds.densify(max_grad=2e-5, min_opacity=0.05, percent_dense=0.002)
    assert info["cloned"] > 0 and info["split"] > 0 and float(ds.stats.denom.sum()) == 0.0
    assert torch.isfinite(ds.train_step(12, perm=list(range(8)))) and float(ds.stats.denom.sum()) > 0


# ---- graph replay of the step the reference runs ---------------------------------------------------------------------

def _twin_scenes(**kw):
    from rodygs_amd.trainstep import DynamicScene
    sc = O.synthetic_scene(20000, 320, 240, 3, seed=5)
    tgt = O.synthetic_scene(5000, 320, 240, 3, seed=6)

    def fresh():
        ds = DynamicScene(sc, num_frames=8, device=DEV, spatial_order=True, **kw)
        ds.make_ground_truth(tgt, range(8))
        return ds
    return fresh


def test_graph_replay_with_statistics_and_a_learning_rate_schedule_is_bit_identical_to_the_eager_step():
    """GraphedStep on the step a real training runs for its first 15-20 k iterations: densification statistics on, the
    xyz learning rate re-set every iteration (rodygs_static.py:143-149: expon_lr) and the SH rate changed once on the way.
    The rates live in device memory (RdgStepScalars.seg_lr_* / sh_lr_*), so the replays follow them; deterministic
    backward on both sides: parameters, both Adam moments, the statistics and the losses agree BIT FOR BIT."""
    import rodygs_amd.rasterizer as R
    from rodygs_amd.trainstep import GraphedStep, expon_lr
    fresh = _twin_scenes()
    old = R.DETERMINISTIC
    R.DETERMINISTIC = True
    try:
        perm, n = [0, 3, 5, 6, 1], 21

        def set_rates(ds, s_):
            ds.fp.lr["xyz"] = expon_lr(s_, 0.0008, 0.0000016 * 5, max_steps=40)
            if s_ >= 9:
                ds.fp.lr["features"] = 0.004

        a = fresh()
        a.track_densification()
        la = []
        for s_ in range(n):
            set_rates(a, s_)
            la.append(a.train_step(s_, perm=perm))
        b = fresh()
        b.track_densification()
        set_rates(b, 0)
        gs = GraphedStep(b, perm, warmup=1)
        lb = []
        for s_ in range(1, n):
            set_rates(b, s_)
            lb.append(gs.step().clone())
        assert gs.check() > 0 and gs.next_step == n
        gs.close()
        torch.cuda.synchronize()
    finally:
        R.DETERMINISTIC = old
    for x, y in zip(la[1:], lb):
        assert torch.equal(x, y)
    for name in ("flat", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(a.fp, name), getattr(b.fp, name)), "Gaussian bucket " + name
        assert torch.equal(getattr(a.sp, name), getattr(b.sp, name)), "MLP + pose bucket " + name
    for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(a.stats, name), getattr(b.stats, name)), name
    assert float(a.stats.denom.max()) >= 4.0
    # a rate override for one step only, as fused_adam_(lr_override=): differs from the un-overridden twin from there on
    c = fresh()
    gc = GraphedStep(c, perm, warmup=1)
    gc.step()
    before = c.fp["opacity"].detach().clone()
    gc.step(lr_override={"opacity": 0.0})
    assert torch.equal(c.fp["opacity"].detach(), before), "lr_override={'opacity': 0} must freeze that group for the step"
    gc.step()
    assert not torch.equal(c.fp["opacity"].detach(), before)
    gc.close()


def test_graph_replay_remembers_an_overflow_in_the_middle_of_a_run():
    """every replay overwrites the same device instance count, so a frame in the MIDDLE of a run that
    outgrew the captured capacity (rendered empty, Adam stepping on zero gradients) went unnoticed unless it was the last
    one.  The forward now folds D into a sticky device maximum (RdgRasterSettings.num_rendered_max): check() raises
    whichever frame it was, and clears the record."""
    import rodygs_amd.rasterizer as R
    from rodygs_amd.trainstep import GraphedStep
    b = _twin_scenes()()
    perm = [0, 3, 5, 6, 1]
    gs = GraphedStep(b, perm, warmup=2)
    for _ in range(3):
        gs.step()
    assert 0 < gs.check() <= gs._cap
    with torch.no_grad():
        b.fp["scaling"].add_(1.4)              # every Gaussian 4x as large for ONE frame: D far above the capacity
    gs.step()
    with torch.no_grad():
        b.fp["scaling"].sub_(1.4)
    for _ in range(4):                         # ordinary frames again: the last replay fits
        gs.step()
    n_last = int(gs._nren[0])
    assert n_last <= gs._cap
    with pytest.raises(R.RasterizerCapacityOverflow):
        gs.check()
    assert b.raster_state.capacity_hint[gs._key] > gs._cap
    for _ in range(2):
        gs.step()
    assert 0 < gs.check() <= gs._cap           # the record was cleared by the check that raised
    gs.close()


def test_graph_replay_of_the_plain_full_loss_steps():
    """The config-5 loss set under GraphedStep: the four steps between two rigidity steps replay a captured graph (Pearson
    depth terms with their random boxes, basis regulariser, motion regularisers after the main backward), the rigidity
    step runs eagerly -- same loss curve as the all-eager twin (float atomics, and the box draws of a replay come from the
    graph's own generator state: compared as curves, not bits)."""
    from rodygs_amd.trainstep import GraphedStep
    fresh = _twin_scenes(full_losses=True)
    perm = list(range(8))
    torch.manual_seed(7)
    a = fresh()
    la = [float(a.train_step(s_, perm=perm)) for s_ in range(1, 25)]
    torch.manual_seed(7)
    b = fresh()
    gs = GraphedStep(b, perm, warmup=2, first_step=1)
    assert gs.next_step == 3
    lb = la[:2] + [float(gs.step()) for _ in range(3, 25)]
    assert gs.check() > 0 and gs.next_step == 25 and b.fp.step_count == a.fp.step_count
    gs.close()
    assert all(np.isfinite(lb))
    plain = [i for i in range(24) if (i + 1) % 5 != 0]
    rig = [i for i in range(24) if (i + 1) % 5 == 0]
    pa, pb = np.array([la[i] for i in plain]), np.array([lb[i] for i in plain])
    assert abs(pb.mean() / pa.mean() - 1.0) < 0.05 and np.allclose(pb, pa, rtol=0.2), (la, lb)
    assert np.allclose([lb[i] for i in rig], [la[i] for i in rig], rtol=0.3), (la, lb)
    assert pb[-6:].mean() < pb[:6].mean()
    assert torch.isfinite(b.train_step(25, perm=perm))          # and eagerly on after the graph is dropped


# ---- caller-owned rasterizer state (reentrant per workspace, no global mutable state) ----------

def _render_once(sc, state, bg=(0.0, 0.0, 0.0), backward=True):
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    P = sc["means3D"].shape[0]
    ins = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(HS.make_settings(sc, 3, bg=torch.tensor(bg)), state=state)(
        means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"], scales=ins["scales"],
        rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
    if backward:
        (out[0].sum() + 0.1 * out[1].sum()).backward()
    return out, ins


def test_two_scenes_of_equal_size_keep_their_own_hints():
    """Two clouds with the same (P, H, W) -- a static and a dynamic model, a training and an evaluation renderer -- one
    with ordinary tile lists, one with lists of several thousand instances: interleaved on their OWN RasterState, neither
    sees the other's capacity / split hints (on the shared default state every other frame would start from the wrong
    capacity and flip the compositing path)."""
    import rodygs_amd.rasterizer as R
    P, W, H = 6000, 320, 240
    small = O.synthetic_scene(P, W, H, 3, seed=12)
    big = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in O.synthetic_scene(P, W, H, 3, seed=13).items()}
    big["scales"] = big["scales"] * 40.0          # nearly every Gaussian on every tile: lists of ~5 k instances
    sa, sb = R.RasterState(), R.RasterState()
    key = (P, H, W)
    R.DEFAULT_STATE.capacity_hint.pop(key, None)
    first = {}
    for it in range(4):
        for name, sc, st in (("small", small, sa), ("big", big, sb)):
            out, _ = _render_once(sc, st)
            d = int(st.last_nren[0][0])
            assert st.capacity_hint[key] == d and st.last_nren[2] >= d
            if it == 0:
                first[name] = (d, out[0].detach().clone())
            else:
                assert d == first[name][0] and torch.equal(out[0].detach(), first[name][1])
    assert first["big"][0] > 8 * first["small"][0]
    assert sb.split_hint.get(key) == 1 and key not in sa.split_hint
    assert key not in R.DEFAULT_STATE.capacity_hint and not sa.pending and not sb.pending
    # the workspace of the small scene was sized from ITS OWN last frame, not from the big one's
    assert sa.last_nren[2] < first["big"][0]


def test_forward_and_backward_on_two_threads():
    """Two threads, each rendering its own scene on its own RasterState (and its own stream), with the library's stage
    timers on: the same images and gradients as the same work done one after the other on one thread."""
    import rodygs_amd.rasterizer as R
    from rodygs_amd import _lib
    scenes = [O.synthetic_scene(4000, 320, 240, 3, seed=31), O.synthetic_scene(4000, 320, 240, 3, seed=32)]
    scenes[1]["viewmatrix"] = orbit_view()
    ref = []
    for sc in scenes:
        out, ins = _render_once(sc, R.RasterState())
        ref.append((out[0].detach().clone(), out[1].detach().clone(), out[4].clone(), ins["shs"].grad.clone()))
    torch.cuda.synchronize()
    results, errors = [None, None], []
    _lib.timing_enable(True)
    _lib.timing_reset()

    def work(i):
        try:
            st = R.RasterState()
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(6):
                    out, ins = _render_once(scenes[i], st)
                stream.synchronize()
            results[i] = (out[0].detach(), out[1].detach(), out[4], ins["shs"].grad)
        except Exception as e:               # noqa: BLE001
            errors.append(e)

    try:
        threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        times = _lib.stage_times()
    finally:
        _lib.timing_enable(False)
    assert not errors, errors
    assert times["render_fwd"][1] == 12 and times["render_bwd"][1] == 12 and times["preprocess_bwd"][1] == 12
    for i in range(2):
        for j in range(3):
            assert torch.equal(results[i][j], ref[i][j]), (i, j)       # the forward is deterministic
        rel_ok(results[i][3], ref[i][3], tol=1e-5, outliers=1e-4, cap=1e-4, what="d_shs on a thread")   # float atomics


# ---- parity evidence the driver runs: both binning algorithms x both backward modes ---------------------------------

def _core_scene(name):
    if name == "c1":
        return O.synthetic_scene(1000, 256, 256, 3, seed=2), 0, (0.0, 0.0, 0.0)
    if name == "ragged6k":
        sc = O.synthetic_scene(6000, 333, 211, 3, seed=4)
        sc["viewmatrix"] = orbit_view()
        return sc, 3, (0.1, 0.2, 0.3)
    sc = O.skewed_scene(320, 240, [(5, 6, 23000), (14, 3, 7900), (9, 11, 4200), (2, 2, 2500)], background=3000,
                        sh_degree_max=3, seed=78, equal_depth_every=5)
    sc["viewmatrix"] = orbit_view(1.0, -0.7, (0.04, -0.02, 0.08))
    return sc, 3, (0.05, 0.1, 0.15)


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("bin_mode", ["bucket", "radix"])
@pytest.mark.parametrize("scene", ["c1", "ragged6k", "skewed23k"])
def test_core_parity_through_both_binning_algorithms_and_both_backward_modes(scene, bin_mode, deterministic):
    """north_star names "tile binning + device radix sort" and SURVEY.md section 5b a deterministic mode: the core cases
    (BASELINE configs[0], a ragged image with a non-identity pose, the skewed scene with a 23 k-instance tile and depth
    ties) through bucket binning AND the LSD radix sort (keys / order / ranges bit-exact vs the oracle), float-atomic AND
    deterministic backward (image + every gradient <= 1e-4 per column) -- in the suite the driver runs."""
    import hip_stages as HS
    import rodygs_amd.rasterizer as R
    sc, deg, bg = _core_scene(scene)
    P, H, W = sc["means3D"].shape[0], sc["H"], sc["W"]
    hs = HS.run_stages(sc, deg, bin_mode=1 if bin_mode == "radix" else 0)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], deg)
    with torch.no_grad():
        g = O.preprocess(sc["means3D"], torch.zeros(P, 3), sc["opacities"], sc["viewmatrix"], st, shs=sc["shs"],
                         scales=sc["scales"], rotations=sc["rotations"])
    b = O.bin_and_sort(g)
    assert hs["D"] == b["num_rendered"]
    for k in ("keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted", "ranges"):
        assert np.array_equal(hs[k], b[k]), (k, bin_mode)
    keep = (R._FORCE_RADIX, R.DETERMINISTIC)
    R._FORCE_RADIX, R.DETERMINISTIC = bin_mode == "radix", deterministic
    try:
        check_pair(run_pair(sc, deg, bg), NAMES)
    finally:
        R._FORCE_RADIX, R.DETERMINISTIC = keep


@pytest.mark.parametrize("seed0,case", [(40000, 104), (40000, 353)])
def test_pose_gradient_of_the_two_deep_list_sweep_cases(seed0, case):
    """The two of 1 100 random scenes of round 3's sweeps (profiles/r03_parity_sweep.txt) whose pose gradient missed the
    bar (1.14e-4 / 1.61e-4): 16-pixel-wide images, per-pixel lists hundreds of splats deep, covariance path of the pose
    gradient and the depth loss on.  Named here so that the suite the driver runs holds them to 1e-4 like everything else."""
    from sweep_cases import sweep_case
    sc, deg, bg, kw = sweep_case(seed0, case)
    check_pair(run_pair(sc, deg, bg, **kw), NAMES)


@pytest.mark.parametrize("profile,seed0,case", [("", 120000, 248), ("aniso", 400000, 349), ("", 310000, 790)])
def test_gradients_of_thin_axes_and_needle_footprints_against_the_float64_oracle(profile, seed0, case):
    """Three cases the strict sweeps of round 4 found (profiles/r04_parity_sweep.txt), where the per-Gaussian backward lost
    accuracy while dL/dcov2D was three numbers contracted with 3x3 matrices: 120000 / 248 (a scale axis that points at the
    camera: d_scales[:, 0], 4e-5 of the other columns, 3.9e-4 off), 400000 / 349 of the anisotropic profile (a 300 x 6 pixel
    needle, det / (a c) = 1.6e-3: d_scales 3e-4 off) and 310000 / 790 (a rotation-gradient column 260x below its neighbours,
    1.06e-4).  Since the backward goes through the projected axes u_k = T r_k (Cauchy-Binet determinant, adjugate without the
    axis' own term) they hold the per-column bar -- here against the oracle run in FLOAT64 without any outlier allowance (the
    float32 oracle is itself outside the bar on the first two)."""
    from sweep_cases import sweep_case, sweep_case_aniso
    from test_gpu_parity import _columns
    sc, deg, bg, kw = (sweep_case_aniso if profile == "aniso" else sweep_case)(seed0, case)
    res = run_pair(sc, deg, bg, **kw)
    hi = res[0]
    P, W, H = sc["means3D"].shape[0], sc["W"], sc["H"]
    gen = torch.Generator().manual_seed(kw["seed"])
    wc, wd, wa = torch.rand(3, H, W, generator=gen), torch.rand(1, H, W, generator=gen), torch.rand(1, H, W, generator=gen)
    wn = torch.randn(3, H, W, generator=gen) * kw["normal_loss"]
    d = {k: sc[k].clone().double().requires_grad_(True) for k in NAMES}
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], torch.tensor(bg).double(), kw["scale_modifier"],
                          sc["projmatrix"].double(), deg, enable_cov_grad=kw["cov_grad"], enable_sh_grad=kw["sh_grad"])
    o = O.rasterize(d["means3D"], torch.zeros(P, 3, dtype=torch.float64, requires_grad=True), d["opacities"], d["viewmatrix"], st,
                    shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
    ls = (o[0] * wc.double()).sum() + (o[3] * wa.double()).sum()
    if kw["depth_loss"]:
        ls = ls + (o[1] * wd.double()).sum() * kw["depth_loss"]
    if kw["normal_loss"]:
        ls = ls + (o[2] * wn.double()).sum()
    ls.backward()
    for k in ("scales", "rotations", "means3D", "opacities"):
        h_, r_ = _columns(hi[k].grad.cpu().double()), _columns(d[k].grad)
        err = (h_ - r_).abs().amax(1) / r_.abs().amax(1).clamp_min(1e-300)
        assert float(err.max()) <= 1e-4, f"d_{k} against the float64 oracle, per column: {[f'{float(e):.2e}' for e in err]}"


# ---- extra_attrs (upstream kwarg, no RoDyGS caller) --------------------------------------------------------------------

@pytest.mark.parametrize("E", [1, 3, 5])
def test_extra_attrs_are_composited_with_the_colour_weights(E):
    """``GaussianRasterizer(...)(..., extra_attrs=[P,E])`` -> ``extra`` [E,H,W] against the oracle compositing the same
    attributes natively (one pass, E more feature columns), with gradients to the attributes and -- through the blending
    weights -- to every other input, summed with the colour loss's (viewmatrix and means2D included)."""
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer
    P, W, H = 3000, 200, 152
    sc = O.synthetic_scene(P, W, H, 3, seed=17)
    sc["viewmatrix"] = orbit_view(5.0, 3.0, (0.2, 0.1, 0.5))
    bg = torch.tensor([0.2, 0.1, 0.3])
    gen = torch.Generator().manual_seed(3)
    attrs = torch.randn(P, E, generator=gen)
    wc, we = torch.rand(3, H, W, generator=gen), torch.randn(E, H, W, generator=gen)
    hi = {k: sc[k].clone().to(DEV).requires_grad_(True) for k in NAMES}
    ha = attrs.clone().to(DEV).requires_grad_(True)
    hm2 = torch.zeros(P, 3, device=DEV, requires_grad=True)
    out = GaussianRasterizer(HS.make_settings(sc, 3, bg=bg))(
        means3D=hi["means3D"], means2D=hm2, shs=hi["shs"], opacities=hi["opacities"], scales=hi["scales"],
        rotations=hi["rotations"], viewmatrix=hi["viewmatrix"], extra_attrs=ha)
    assert out[5].shape == (E, H, W)
    ((out[0] * wc.to(DEV)).sum() + (out[5] * we.to(DEV)).sum()).backward()
    oi = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
    oa = attrs.clone().requires_grad_(True)
    om2 = torch.zeros(P, 3, requires_grad=True)
    st = O.OracleSettings(H, W, sc["tanfovx"], sc["tanfovy"], bg, 1.0, sc["projmatrix"], 3)
    oo = O.rasterize(oi["means3D"], om2, oi["opacities"], oi["viewmatrix"], st, shs=oi["shs"], scales=oi["scales"],
                     rotations=oi["rotations"], extra_attrs=oa)
    ((oo[0] * wc).sum() + (oo[5]["extra"] * we).sum()).backward()
    rel_ok(out[0], oo[0], outliers=2e-5, what="color")
    rel_ok(out[5], oo[5]["extra"], outliers=2e-5, what="extra")
    rel_ok(ha.grad, oa.grad, outliers=2e-5, what="d_extra_attrs")
    for k in NAMES:
        rel_ok(hi[k].grad, oi[k].grad, outliers=2e-5, what="d_" + k)
    rel_ok(hm2.grad, om2.grad, outliers=2e-5, what="d_means2D")
