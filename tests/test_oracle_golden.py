"""CPU tests: pin the oracle (and the host-side mirrors) against golden vectors produced by the IMPORTED
reference (tests/golden/make_golden.py).  These are the only reference-derived pins that exist for this path
(the rasterizer's own arithmetic is in an un-vendored submodule -> "parity unpinned" there)."""
import os

import numpy as np
import pytest
import torch

from oracle import deform_oracle as DO
from oracle import rasterizer_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_sh_basis_matches_reference_eval_sh():
    g = load("sh_golden.npz")
    sh = torch.from_numpy(g["sh"])          # reference layout [n, C, K]
    dirs = torch.from_numpy(g["dirs"])
    shs = sh.permute(0, 2, 1).contiguous()  # rasterizer layout [n, K, C]
    for deg in range(4):
        got = O.eval_sh_rgb(deg, shs, dirs)
        np.testing.assert_allclose(got.numpy(), g[f"deg{deg}"], rtol=1e-5, atol=1e-6)
    # RGB2SH / SH2RGB constants
    rgb = torch.from_numpy(g["rgb"])
    np.testing.assert_allclose(((rgb - 0.5) / O.SH_C0).numpy(), g["rgb2sh"], rtol=1e-6)
    np.testing.assert_allclose((rgb * O.SH_C0 + 0.5).numpy(), g["sh2rgb"], rtol=1e-6)


def test_projection_matrix_matches_reference():
    g = load("camera_golden.npz")
    for i in range(4):
        Pm = O.projection_matrix(0.01, 100.0, float(g[f"fovx{i}"]), float(g[f"fovy{i}"]))
        np.testing.assert_allclose(Pm.numpy(), g[f"proj{i}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(Pm.numpy(), g[f"proj_fn{i}"], rtol=1e-6, atol=1e-7)


def test_view_transform_convention():
    """world_view_transform of FixedCameraTorch: R_w2c = R(q)^T, t = -R_w2c T; the oracle's glm-flat view
    transform of a point must equal W2C @ p."""
    g = load("camera_golden.npz")
    for i in range(4):
        w2c = torch.from_numpy(g[f"w2c{i}"])
        V = w2c.t().contiguous().reshape(16)
        p = torch.tensor([[0.3, -0.2, 1.5]])
        vx = ((V[0] * p[:, 0] + V[4] * p[:, 1]) + V[8] * p[:, 2]) + V[12]
        vy = ((V[1] * p[:, 0] + V[5] * p[:, 1]) + V[9] * p[:, 2]) + V[13]
        vz = ((V[2] * p[:, 0] + V[6] * p[:, 1]) + V[10] * p[:, 2]) + V[14]
        ref = (w2c[:3, :3] @ p[0] + w2c[:3, 3])
        np.testing.assert_allclose(torch.stack([vx, vy, vz], 1)[0].numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
        # campos = -R^T t as derived in the oracle / kernel
        camx = -((V[0] * V[12] + V[1] * V[13]) + V[2] * V[14])
        camy = -((V[4] * V[12] + V[5] * V[13]) + V[6] * V[14])
        camz = -((V[8] * V[12] + V[9] * V[13]) + V[10] * V[14])
        c2w = torch.linalg.inv(w2c)
        np.testing.assert_allclose(torch.stack([camx, camy, camz]).numpy(), c2w[:3, 3].numpy(), rtol=1e-4, atol=1e-5)


def test_covariance_matches_reference_on_unit_quaternions():
    """Reference builds Sigma = R S S^T R^T with a NORMALISED quaternion (general_utils.py:92-127); the rasterizer
    path uses the raw quaternion, so they agree exactly when fed the unit quaternion."""
    g = load("cov_golden.npz")
    cov = O.covariance3d(torch.from_numpy(g["scales"]), float(g["scale_modifier"]), torch.from_numpy(g["rot_unit"]))
    np.testing.assert_allclose(cov.numpy(), g["cov6"], rtol=2e-5, atol=1e-6)
    R = torch.stack(O.rotation_from_raw_quat(torch.from_numpy(g["rot_unit"])), dim=1).reshape(-1, 3, 3)
    np.testing.assert_allclose(R.numpy(), g["R"], rtol=1e-5, atol=1e-6)


def test_time_embedding_matches_reference():
    g = load("deform_golden.npz")
    emb = DO.time_embedding(torch.tensor(float(g["t_now"])))
    # arguments reach pi*2^25: agreement needs the same float32 product and a full-range sin/cos
    np.testing.assert_allclose(emb.numpy(), g["emb_now"], rtol=0, atol=2e-6)
    embs = DO.time_embedding(torch.from_numpy(g["times"]))
    np.testing.assert_allclose(embs.numpy(), g["embs"], rtol=0, atol=2e-6)


def _sd(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}


def test_deformation_oracle_matches_reference_forward_and_grads():
    g = load("deform_golden.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(g).items()}
    coeff = torch.from_numpy(g["coeff"]).requires_grad_(True)
    emb_now = torch.from_numpy(g["emb_now"])
    embs = torch.from_numpy(g["embs"])
    basis_t = DO.motion_basis(sd, emb_now)
    table = DO.motion_basis(sd, embs)
    np.testing.assert_allclose(table.detach().numpy(), g["table"], rtol=1e-4, atol=1e-6)
    tr, ro = DO.gaussian_deformation(coeff, torch.from_numpy(g["time_ind"]), basis_t, table, float(g["spatial"]))
    np.testing.assert_allclose(tr.detach().numpy(), g["trans"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ro.detach().numpy(), g["rot"], rtol=1e-4, atol=1e-5)
    loss = (tr * torch.from_numpy(g["wx"])).sum() + (ro * torch.from_numpy(g["wr"])).sum()
    loss.backward()
    np.testing.assert_allclose(coeff.grad.numpy(), g["d_coeff"], rtol=1e-4, atol=1e-5)
    # the last-layer biases cancel analytically between the forward and inverse motion (pure rounding noise),
    # so errors are judged against the largest parameter gradient as well
    gmax = max(np.abs(g[k]).max() for k in g.files if k.startswith("dsd."))
    for k in g.files:
        if k.startswith("dsd."):
            ref = g[k]
            got = sd[k[4:]].grad.numpy()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-6 * gmax, k


def test_host_mlp_mirror_is_state_dict_compatible_and_matches_reference():
    """rodygs_amd.deform.MLPBasisNetwork (torch host code, runs on CPU too) vs the imported reference class."""
    from rodygs_amd.deform import MLPBasisNetwork
    g = load("deform_golden.npz")
    net = MLPBasisNetwork(128, 16, 26, False, activation="gelu")
    missing = net.load_state_dict(_sd(g), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    emb = net.t_embedder(torch.tensor(float(g["t_now"])))
    assert emb.shape == (53,)
    np.testing.assert_allclose(emb.numpy(), g["emb_now"], atol=2e-6)
    assert net.t_embedder(torch.tensor([0.25])).shape == (53, 1)     # reference stacks along dim 0
    embs = net.batch_embedding(torch.from_numpy(g["times"]))
    np.testing.assert_allclose(embs.numpy(), g["embs"], atol=2e-6)
    table = net.batch_inference(torch.from_numpy(g["embs"]))
    np.testing.assert_allclose(table.detach().numpy(), g["table"], rtol=1e-4, atol=1e-6)


def test_losses_restated_for_bench_match_reference():
    from rodygs_amd.losses import l1_loss, ssim
    g = load("loss_golden.npz")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    np.testing.assert_allclose(l1_loss(a, b).numpy(), g["l1"], rtol=1e-6)
    np.testing.assert_allclose(ssim(a, b).numpy(), g["ssim"], rtol=1e-5)


class _FakeDynModel:
    def __init__(self, xyz, coeff, fdc, table):
        self._xyz, self._motion_coeff, self._features_dc = xyz, coeff, fdc
        self.temporal_motion_table = table
        self.unique_times = list(range(table.shape[0]))

    def get_motion_for_times(self, timesteps, time_indices=None):
        return self.temporal_motion_table[time_indices]


RIGIDITY_CASES = {"coeff": dict(mode=["coeff"]),
                  "coeff_l1_nocolor": dict(mode=["coeff"], sim_metric="l1", color_sim=False),
                  "all": dict(mode=["coeff", "surface", "distance_preserving"], K=8, scale=2)}


def run_rigidity_case(name, dev, knn_points=None, knn_gather=None):
    """The host mirror on the golden inputs with the reference's seeds; returns (loss, grads dict, golden)."""
    import random
    from rodygs_amd.rigidity import RigidityLoss
    g = load("rigidity_golden.npz")
    t = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ("xyz", "transl", "coeff", "fdc", "table")}
    random.seed(99)
    torch.manual_seed(7)
    model = _FakeDynModel(t["xyz"], t["coeff"], t["fdc"], t["table"])
    loss = RigidityLoss(**RIGIDITY_CASES[name], knn_points=knn_points, knn_gather=knn_gather)(model, t["transl"])
    grads = torch.autograd.grad(loss, [t[k] for k in ("xyz", "transl", "coeff", "fdc", "table")], allow_unused=True)
    return loss, dict(zip(("xyz", "transl", "coeff", "fdc", "table"), grads)), g


@pytest.mark.parametrize("name", list(RIGIDITY_CASES))
def test_rigidity_mirror_matches_reference_with_oracle_knn(name):
    """rodygs_amd.rigidity.RigidityLoss (host logic) + the brute-force knn restatement == the imported reference's
    RigidityLoss run with the same knn restatement (tests/golden/make_golden.py G7)."""
    from oracle import knn_oracle as KO
    loss, grads, g = run_rigidity_case(name, "cpu", KO.knn_points_batched, KO.knn_gather)
    assert abs(float(loss) - float(g[name + ".loss"])) <= 1e-6 * abs(float(g[name + ".loss"]))
    for k, gr in grads.items():
        want = torch.from_numpy(g[f"{name}.d_{k}"])
        if gr is None:
            assert want.numel() == 1 and float(want.abs().sum()) == 0.0
            continue
        assert float((gr - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-12, k


@pytest.mark.parametrize("mode", [None, "static", "dynamic"])
def test_depth_loss_oracle_matches_reference(mode):
    """oracle/depth_loss_oracle.py == the imported reference's Global/LocalPearsonDepthLoss (golden G8)."""
    from oracle import depth_loss_oracle as DL
    g = load("depth_loss_golden.npz")
    tag = str(mode)
    gt, motion = torch.from_numpy(g["gt"]), torch.from_numpy(g["motion"])
    mask = None if mode is None else (~motion if mode == "static" else motion)
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    lg = DL.pearson_depth_loss(pred, gt, 1e-6, mask)
    (dg,) = torch.autograd.grad(lg, pred)
    assert abs(float(lg) - float(g[f"global.{tag}.loss"])) <= 1e-6
    assert float((dg - torch.from_numpy(g[f"global.{tag}.d_pred"])).abs().max()) <= 1e-9
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    rows, cols = torch.from_numpy(g[f"local.{tag}.rows"]), torch.from_numpy(g[f"local.{tag}.cols"])
    ll = DL.local_pearson_depth_loss(pred, gt, rows, cols, int(g["box_p"]), len(rows), mask)
    (dl,) = torch.autograd.grad(ll, pred)
    assert abs(float(ll) - float(g[f"local.{tag}.loss"])) <= 1e-6
    assert float((dl - torch.from_numpy(g[f"local.{tag}.d_pred"])).abs().max()) <= 1e-8


@pytest.mark.parametrize("name,build", [
    ("l1", lambda ML: ML.MotionL1Loss()), ("sparsity", lambda ML: ML.MotionSparsityLoss()),
    ("basis_d0", lambda ML: ML.MotionBasisRegularizaiton(transl_degree=0)),
    ("basis_d1_gauss", lambda ML: ML.MotionBasisRegularizaiton(transl_degree=1, rot_degree=1, freq_div_mode="gaussian"))])
def test_motion_regularisers_match_reference(name, build):
    """rodygs_amd.motion_losses == the imported reference's MotionL1Loss / MotionSparsityLoss /
    MotionBasisRegularizaiton (golden G9): value and gradients w.r.t. coefficients and motion table."""
    from rodygs_amd import motion_losses as ML
    g = load("motion_reg_golden.npz")
    coeff = torch.from_numpy(g["coeff"]).requires_grad_(True)
    table = torch.from_numpy(g["table"]).requires_grad_(True)

    class M:
        _motion_coeff = coeff

        @staticmethod
        def get_total_motion_table():
            return table

    v = build(ML)(M)
    gc, gt = torch.autograd.grad(v, [coeff, table], allow_unused=True)
    assert abs(float(v) - float(g[name + ".loss"])) <= 1e-6 * max(1.0, abs(float(g[name + ".loss"])))
    for got, key in ((gc, ".d_coeff"), (gt, ".d_table")):
        want = torch.from_numpy(g[name + key])
        if got is None:
            assert float(want.abs().sum()) == 0.0
        else:
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-12


# ---- G4: the rasterizer fixture (oracle-generated: freezes the spec against joint drift of oracle + kernels) -----

RASTER_INT = ("radii", "tiles_touched", "keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted", "ranges")
RASTER_IMG = ("color", "depth", "normal", "alpha", "final_T")
RASTER_GRAD = ("grad_means3D", "grad_means2D", "grad_shs", "grad_opacities", "grad_scales", "grad_rotations",
               "grad_viewmatrix")


def raster_fixture_inputs(g):
    inp = {k: torch.from_numpy(g["in_" + k]) for k in ("means3D", "shs", "opacities", "scales", "rotations",
                                                        "viewmatrix")}
    cfg = dict(deg=int(g["in_sh_degree"]), bg=tuple(float(v) for v in g["in_bg"]), H=int(g["in_H"]), W=int(g["in_W"]),
               tanx=float(g["in_tanfovx"]), tany=float(g["in_tanfovy"]), proj=torch.from_numpy(g["in_projmatrix"]))
    return inp, cfg


@pytest.mark.parametrize("scene", ["c1", "skewed"])
def test_oracle_reproduces_the_committed_rasterizer_fixture(scene):
    """Every integer of the binning stage bit for bit, every float to 2e-5 of its tensor's max (the compositing sums
    may be ordered differently by a different thread count), n_contrib on all but a discontinuity's handful of pixels."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_rasterizer_golden", os.path.join(G, "make_rasterizer_golden.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    g = load(f"rasterizer_golden_{scene}.npz")
    inp, c = raster_fixture_inputs(g)
    out = M.run_oracle(inp, c["deg"], c["bg"], c["H"], c["W"], c["tanx"], c["tany"], c["proj"], cull=False)
    assert int(out["num_rendered"]) == int(g["num_rendered"])
    for k in RASTER_INT:
        assert np.array_equal(out[k], g[k]), k
    for k in RASTER_IMG + RASTER_GRAD:
        scale = np.abs(g[k]).max() + 1e-30
        assert np.abs(out[k] - g[k]).max() <= 2e-5 * scale, (k, np.abs(out[k] - g[k]).max() / scale)
    assert (out["n_contrib"] != g["n_contrib"]).mean() <= 2e-5
    # the tight rectangles (RdgRasterSettings.cull = 1): their integers are frozen too; every float is the SAME fixture's
    cul = M.run_oracle(inp, c["deg"], c["bg"], c["H"], c["W"], c["tanx"], c["tany"], c["proj"], cull=True)
    assert int(cul["num_rendered"]) == int(g["cull_num_rendered"]) < int(g["num_rendered"])
    assert np.array_equal(cul["radii"], g["radii"])
    for k in ("tiles_touched", "keys_unsorted", "vals_unsorted", "keys_sorted", "vals_sorted", "ranges"):
        assert np.array_equal(cul[k], g["cull_" + k]), k
    for k in RASTER_IMG + RASTER_GRAD:
        scale = np.abs(g[k]).max() + 1e-30
        assert np.abs(cul[k] - g[k]).max() <= 2e-5 * scale, ("cull", k, np.abs(cul[k] - g[k]).max() / scale)
    assert (cul["n_contrib"] != g["cull_n_contrib"]).mean() <= 2e-5
    if scene == "skewed":                           # the fixture really exercises what it is there for
        r = g["ranges"].astype(np.int64)
        n = np.sort(r[:, 1] - r[:, 0])
        assert n[-1] > 8192 and (n[-3:-1] > 1024).all() and int(g["n_contrib"].max()) > 8192
        ks = g["keys_sorted"]
        assert (ks[1:] == ks[:-1]).sum() > 100      # depth ties: order falls back to the Gaussian index


def test_xyz_lr_schedule_matches_reference_get_expon_lr_func():
    """rodygs_amd.trainstep.expon_lr against values the imported reference function produced (golden G11)."""
    from rodygs_amd.trainstep import expon_lr
    g = load("optimizer_golden.npz")
    for tag in ("a", "b"):
        lr_init, lr_final, delay_steps, delay_mult, max_steps = g["lr_kw_" + tag]
        got = [expon_lr(int(s), lr_init, lr_final, int(delay_steps), delay_mult, int(max_steps)) for s in g["lr_steps"]]
        np.testing.assert_allclose(got, g["lr_" + tag], rtol=1e-12)
    assert expon_lr(-1, 1e-3, 1e-5) == 0.0 and expon_lr(5, 0.0, 0.0) == 0.0


def densify_case(tag):
    """G13 (tests/golden/densify_golden.npz: DynTrainer.densify_and_prune RUN by make_golden.py on the CPU): inputs in the
    flat-bucket layout (features = cat(f_dc, f_rest)), the call's arguments, the recorded split draws, and the reference's
    outputs in the same layout."""
    g = load("densify_golden.npz")
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "motion_coeff")

    def bucket(prefix):
        d = {k: torch.from_numpy(g[f"{tag}.{prefix}{k}"]) for k in names}
        d["features"] = torch.cat([d.pop("f_dc"), d.pop("f_rest")], dim=1)
        return d
    ins = dict(params=bucket("in."), exp_avg=bucket("in.exp_avg."), exp_avg_sq=bucket("in.exp_avg_sq."))
    outs = dict(params=bucket("out."), exp_avg=bucket("out.exp_avg."), exp_avg_sq=bucket("out.exp_avg_sq."))
    for d, pre in ((ins, "in."), (outs, "out.")):
        d.update(accum=torch.from_numpy(g[f"{tag}.{pre}accum"]), denom=torch.from_numpy(g[f"{tag}.{pre}denom"]),
                 max_radii=torch.from_numpy(g[f"{tag}.{pre}max_radii"]),
                 per_point={k: torch.from_numpy(g[f"{tag}.{pre}{k}"]) for k in ("gaussian_to_time", "gaussian_to_time_ind")})
    max_grad, min_opacity, extent, mss, percent_dense, N = (float(v) for v in g[f"{tag}.args"])
    args = dict(max_grad=max_grad, min_opacity=min_opacity, extent=extent, max_screen_size=(mss or None),
                percent_dense=percent_dense, N=int(N))
    steps = {k: float(g[f"{tag}.out.step.{k}"]) for k in names}
    return ins, outs, args, torch.from_numpy(g[f"{tag}.z"]), steps


@pytest.mark.parametrize("tag", ["a", "b"])
def test_densify_oracle_matches_the_reference_trainers_own_densify_and_prune(tag):
    """G13 pins oracle/densify_oracle.py with the reference itself: DynTrainer.densify_and_prune
    (/root/reference/src/trainer/rodygs_static.py:280-315 with densify_and_clone / densify_and_split / prune_points /
    densification_postfix, rodygs_dynamic.py:150-197, utils.py:36-95) was RUN on a model and optimizer the reference built
    from a checkpoint; the oracle must reproduce every parameter row, both Adam moments, the statistics and the per-Gaussian
    time arrays BIT FOR BIT (same torch CPU operations in the same order; the split draws are the recorded ones)."""
    from oracle import densify_oracle as DZ
    ins, outs, a, z, steps = densify_case(tag)
    st = DZ.State({k: v.clone() for k, v in ins["params"].items()}, {k: v.clone() for k, v in ins["exp_avg"].items()},
                  {k: v.clone() for k, v in ins["exp_avg_sq"].items()}, ins["accum"].clone(), ins["denom"].clone(),
                  ins["max_radii"].clone(), {k: v.clone() for k, v in ins["per_point"].items()})
    P0 = st.P
    n_clone, n_sel = DZ.densify_and_prune(st, a["max_grad"], a["min_opacity"], a["extent"], a["max_screen_size"],
                                          a["percent_dense"], a["N"], z)
    assert n_clone > 20 and n_sel > 20 and z.shape[0] == a["N"] * n_sel
    assert st.P == outs["params"]["xyz"].shape[0] != P0
    for k, v in outs["params"].items():
        assert torch.equal(st.params[k], v), k
        assert torch.equal(st.exp_avg[k], outs["exp_avg"][k]) and torch.equal(st.exp_avg_sq[k], outs["exp_avg_sq"][k]), k
    assert torch.equal(st.accum, outs["accum"]) and torch.equal(st.denom, outs["denom"])
    assert torch.equal(st.max_radii, outs["max_radii"]) and float(st.max_radii.abs().sum()) == 0.0
    for k, v in outs["per_point"].items():
        assert torch.equal(st.per_point[k], v), k
    assert set(steps.values()) == {5.0}          # the surgery keeps the step counter of every group
    if tag == "b":                               # the world-size prune (0.1 * extent) fired on top of the opacity prune
        big = torch.exp(ins["params"]["scaling"]).max(dim=1).values > 0.1 * a["extent"]
        assert int(big.sum()) > 5
