"""CPU tests: the C-ABI library loads and exports every symbol include/rodygs_hip.h declares (no compute calls
without a GPU), workspace-size functions are sane, and the host layer mirrors the reference surface and FAILS
LOUDLY (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "rodygs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rdg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_lib):
    import ctypes
    from rodygs_amd import _lib
    names = _header_functions()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/rodygs_hip.h but not exported"
    # and the ctypes binding table covers the whole header
    assert set(names) == set(_lib.EXPORTED_SYMBOLS)
    assert hip_lib.rdg_abi_version() == _lib.ABI_VERSION == 7


def test_workspace_sizes(hip_lib):
    assert hip_lib.rdg_geom_bytes(1000) >= 1000 * 64 + 1000 * 4
    assert hip_lib.rdg_geom_bytes(0) > 0
    b1 = hip_lib.rdg_binning_bytes(1 << 20, 8160)
    assert b1 >= (1 << 20) * 24
    assert hip_lib.rdg_binning_bytes(1 << 21, 8160) > b1
    assert hip_lib.rdg_image_bytes(1080, 1920) >= 1080 * 1920 * 8 + 8160 * 8
    assert hip_lib.rdg_grad_bytes(1000) >= 1000 * 64
    assert hip_lib.rdg_sort_tmp_bytes(1000) > 0 and hip_lib.rdg_knn_tmp_bytes(1000) > 0


def _lib_mod():
    from rodygs_amd import _lib
    return _lib


def test_c_struct_matches_header():
    import ctypes
    from rodygs_amd._lib import RdgRasterSettings
    # 18 int32 / float fields, two pointers, the three statistics pointers + (rows, reserved), the sticky-count pointer, the
    # auxiliary stream of the backward's pose chain
    assert ctypes.sizeof(RdgRasterSettings) == 18 * 4 + 16 + 24 + 8 + 8 + 8
    assert ctypes.sizeof(_lib_mod().RdgStepScalars) == 128 == 4 * _lib_mod().STEP_SCALARS_FLOATS
    assert RdgRasterSettings.densify_grad_accum.offset == 88 and RdgRasterSettings.num_rendered_max.offset == 120
    from rodygs_amd import _lib
    assert _lib.lib().rdg_settings_bytes() == ctypes.sizeof(RdgRasterSettings)      # the compiled header's sizeof
    assert RdgRasterSettings.zero_grad_ws.offset == 72
    assert [f[0] for f in RdgRasterSettings._fields_][:8] == [
        "P", "M", "sh_degree", "image_height", "image_width", "tanfovx", "tanfovy", "scale_modifier"]


def test_settings_surface_matches_reference_call_site():
    """Same 12 keyword fields, in the order of /root/reference/src/trainer/renderer.py:50-63."""
    from rodygs_amd import GaussianRasterizationSettings
    import diff_gauss_pose
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "projmatrix", "sh_degree",
        "prefiltered", "debug", "enable_cov_grad", "enable_sh_grad")
    assert diff_gauss_pose.GaussianRasterizationSettings is GaussianRasterizationSettings
    from simple_knn._C import distCUDA2
    import rodygs_amd
    assert distCUDA2 is rodygs_amd.distCUDA2


def _rs():
    from rodygs_amd import GaussianRasterizationSettings
    return GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), 0, False, False, True,
                                         True)


def test_rasterizer_argument_errors_match_upstream_behaviour():
    from rodygs_amd import GaussianRasterizer
    r = GaussianRasterizer(_rs())
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), scales=m, rotations=torch.zeros(4, 4),
          viewmatrix=torch.eye(4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), colors_precomp=m, viewmatrix=torch.eye(4))


def test_no_cpu_fallback_anywhere():
    """The product path must fail loudly on CPU tensors instead of silently computing somewhere else."""
    from rodygs_amd import GaussianRasterizer, distCUDA2, gaussian_deformation
    r = GaussianRasterizer(_rs())
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4),
          viewmatrix=torch.eye(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        distCUDA2(torch.zeros(10, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gaussian_deformation(torch.zeros(4, 16), torch.zeros(4, dtype=torch.int64), torch.zeros(16, 7), None, 1.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rodygs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
    for shim in ("diff_gauss_pose/__init__.py", "simple_knn/_C.py"):
        assert "oracle" not in open(os.path.join(ROOT, shim)).read()


def test_deformation_field_birth_time_keys():
    """gaussian_to_time_ind follows int(trunc(float32(t)*1000)) keys over sorted unique birth times
    (/root/reference/src/model/rodygs_dynamic.py:44,56-77) -- checked on the host logic only (CPU)."""
    from rodygs_amd.deform import DeformationField
    t = torch.tensor([0.30, 0.00, 0.10, 0.30, 0.10, 0.99])
    f = DeformationField(6, t, device="cpu")
    assert f.unique_times == [0, 100, 300, 990] or f.unique_times == [0, 100, 300, 989]
    assert f.gaussian_to_time_ind.tolist() == [2, 0, 1, 2, 1, 3]
    assert f._time_batch_embeddings.shape == (4, 53)
    assert f.get_total_motion_table().shape == (4, 16, 7)
    assert DeformationField.timetokey(0.25) == 250


def test_checkpoint_wire_format_round_trip_and_torch_adam_compat(tmp_path):
    """export -> torch.save((sd, it)) -> load -> FlatParams keeps every tensor and moment; the exported optimizer
    state loads into a torch.optim.Adam built the way the reference builds its groups (rodygs_static.py:106-141);
    the keys are the ones create_from_state_dict / the evaluator read (rodygs_static.py:172-182,
    rodygs_dynamic.py:106-120, evaluator/eval.py:51-78)."""
    import torch
    from rodygs_amd import checkpoint as CK
    from rodygs_amd.deform import MLPBasisNetwork
    from rodygs_amd.dp import FlatParams
    P, K, B, T = 37, 16, 16, 5
    spec = {"xyz": ((P, 3), 1.6e-4), "features": ((P, K, 3), 2.5e-3), "scaling": ((P, 3), 1e-3),
            "rotation": ((P, 4), 1e-3), "opacity": ((P, 1), 5e-2), "motion_coeff": ((P, 1, B), 1.6e-4)}
    fp = FlatParams(spec, "cpu")
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        fp.flat.copy_(torch.randn(fp.numel, generator=g))
        fp.exp_avg.copy_(torch.randn(fp.numel, generator=g))
        fp.exp_avg_sq.copy_(torch.rand(fp.numel, generator=g))
    fp.step_count = 123
    net = MLPBasisNetwork(128, 16, 26, False)
    t_birth = torch.rand(P, generator=g)
    cams = (torch.randn(T, 4, generator=g), torch.randn(T, 3, generator=g))
    sd = CK.export_state_dict(fp, 7000, 3, 5.5, net, t_birth, cams, feature_lr_rest=2.5e-3 / 20)
    assert set(sd) == {"iteration", "active_sh_degree", "model", "optim", "spatial_lr_scale", "camera"}
    assert set(sd["model"]) == {"_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity",
                                "_motion_coeff", "_deform_network", "_timestep"}
    assert sd["model"]["_features_dc"].shape == (P, 1, 3) and sd["model"]["_features_rest"].shape == (P, K - 1, 3)
    assert "basis_xyz.15.basis.2.weight" in sd["model"]["_deform_network"]          # reference per-head key names
    assert set(sd["optim"]) == {"max_radii2D", "xyz_gradient_accum", "denom", "optimizer"}
    assert set(sd["camera"]) == {"R_c2ws_quat", "T_c2ws"}
    path = tmp_path / "dynamic_last.ckpt"
    CK.save_checkpoint(str(path), sd)
    raw = torch.load(str(path), weights_only=False)
    assert isinstance(raw, tuple) and raw[1] == 7000                                  # (state_dict, iteration)
    back = CK.load_checkpoint(str(path))
    fp2 = CK.flat_params_from_state_dict(back, {k: v[1] for k, v in spec.items()}, "cpu")
    assert fp2.step_count == 123
    for k in fp.names:
        o, n = fp.offsets[k]
        o2, n2 = fp2.offsets[k]
        assert torch.equal(fp[k], fp2[k]), k
        assert torch.equal(fp.exp_avg[o:o + n], fp2.exp_avg[o2:o2 + n2]), k
        assert torch.equal(fp.exp_avg_sq[o:o + n], fp2.exp_avg_sq[o2:o2 + n2]), k
    # a torch Adam with the reference's groups (six single-tensor groups, the MLP's 70 tensors as "deform_network", then
    # the motion coefficients: rodygs_static.py:106-141, rodygs_dynamic.py:93-116) accepts the exported optimizer state
    m = back["model"]
    order = [("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
             ("scaling", "_scaling"), ("rotation", "_rotation")]
    params = [torch.nn.Parameter(m[key].clone()) for _, key in order]
    mlp = [torch.nn.Parameter(m["_deform_network"][n].clone()) for n in CK.reference_mlp_param_names(16)]
    coeff = torch.nn.Parameter(m["_motion_coeff"].clone())
    groups = [{"params": [p], "lr": 1e-3, "name": nm} for p, (nm, _) in zip(params, order)]
    groups += [{"params": mlp, "lr": 1e-3, "name": "deform_network"}, {"params": [coeff], "lr": 1e-3, "name": "motion_coeff"}]
    opt = torch.optim.Adam(groups, eps=1e-15)
    opt.load_state_dict(back["optim"]["optimizer"])
    assert [g_["name"] for g_ in opt.param_groups] == [nm for nm, _ in order] + ["deform_network", "motion_coeff"]
    assert len(opt.param_groups[6]["params"]) == 70 and opt.state[coeff]["exp_avg"].shape == (P, 1, B)
    assert abs(opt.param_groups[2]["lr"] - 2.5e-3 / 20) < 1e-12
    assert torch.equal(opt.state[params[1]]["exp_avg"], back["optim"]["optimizer"]["state"][1]["exp_avg"])
    # PSNR: 10 log10(1 / MSE) on clipped images
    a = torch.full((3, 4, 4), 0.5)
    b = a + 0.1
    assert abs(float(CK.psnr(a, b)) - 20.0) < 1e-4
    assert abs(float(CK.psnr(a, a + 2.0)) - float(10 * torch.log10(torch.tensor(1 / 0.25)))) < 1e-5   # clipped to 1


def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    return M


def _reference_layout_optimizer(model_dict):
    """torch.optim.Adam over the tensors of a checkpoint's "model" entry with the reference's eight groups."""
    import torch
    from rodygs_amd import checkpoint as CK
    single = [("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
              ("scaling", "_scaling"), ("rotation", "_rotation")]
    params = {nm: torch.nn.Parameter(model_dict[key].clone()) for nm, key in single}
    mlp_names = CK.reference_mlp_param_names(16)
    mlp = [torch.nn.Parameter(model_dict["_deform_network"][n].clone()) for n in mlp_names]
    params["motion_coeff"] = torch.nn.Parameter(model_dict["_motion_coeff"].clone())
    groups = [{"params": [params[nm]], "lr": 0.0, "name": nm} for nm, _ in single]
    groups += [{"params": mlp, "lr": 0.0, "name": "deform_network"},
               {"params": [params["motion_coeff"]], "lr": 0.0, "name": "motion_coeff"}]
    return torch.optim.Adam(groups, lr=0.0, eps=1e-15), params, dict(zip(mlp_names, mlp))


def test_exported_checkpoint_as_the_reference_loader_reads_it_golden():
    """G12 (tests/golden/make_golden.py checkpoint): the dictionary rodygs_amd.checkpoint.export_state_dict writes was
    saved, loaded and handed to the REFERENCE's DynRoDyGS.create_from_state_dict / sync_gaussian_to_time_ind /
    get_total_motion_table / get_gaussian_deformation / getters, and its optimizer state to the optimizer the
    reference's trainer builds (eight groups, deform_network before motion_coeff), which then took one Adam step and
    wrote its own checkpoint.  Here: (a) what the reference computed from the dictionary equals what the build computes
    from its flat buckets; (b) a torch Adam in the reference's layout, loaded with the exported state and given the same
    gradients, lands on the reference's post-step parameters and moments; (c) the reference-written checkpoint imports
    back into the flat buckets (Gaussians and MLP, values and both moments)."""
    import numpy as np
    import torch
    from oracle import deform_oracle as DO
    from rodygs_amd import checkpoint as CK
    from rodygs_amd.deform import MLPBasisNetwork
    from rodygs_amd.trainstep import bind_module_to_flat
    M = _golden_module()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_golden.npz"))
    fp, net, sp, g2t, cams, _ = M.checkpoint_inputs()
    sd = CK.export_state_dict(fp, 5, 3, M.CKPT_SCALE, net, g2t, cams, feature_lr_rest=M.CKPT_LR["feature_lr"] / 20.0,
                              deform_state=sp, deform_lr=M.CKPT_DEFORM["deform_lr_init"])
    T = lambda k: torch.from_numpy(g[k])   # noqa: E731
    # ---- (a) the model the reference built from the dictionary ----
    keys = torch.trunc(g2t * 1000).to(torch.int64)
    uniq = torch.unique(keys)
    assert uniq.tolist() == g["unique_keys"].tolist()
    time_ind = torch.searchsorted(uniq, keys)
    assert torch.equal(time_ind, T("time_ind"))
    real_times = torch.sort(torch.unique(g2t)).values
    assert torch.equal(real_times, T("real_times"))
    with torch.no_grad():
        table = net.batch_inference(net.batch_embedding(real_times))
        assert float((table - T("table")).abs().max()) <= 2e-6 * float(T("table").abs().max())
        c = fp["motion_coeff"]
        for i, t in enumerate((0.0, 0.37)):
            basis = net.motion_basis(net.t_embedder(torch.tensor(t)).reshape(1, -1)).squeeze(0)
            dxyz, drot = DO.gaussian_deformation(c, time_ind, basis, table, M.CKPT_SCALE)
            # the random MLP weights make the deltas O(100): relative bar
            sx, sr = float(T(f"deform_xyz_{i}").abs().max()), float(T(f"deform_rot_{i}").abs().max())
            assert float((dxyz - T(f"deform_xyz_{i}")).abs().max()) <= 2e-6 * sx, i
            assert float((drot - T(f"deform_rot_{i}")).abs().max()) <= 2e-6 * sr, i
        assert torch.equal(fp["xyz"], T("get_xyz")) and torch.equal(fp["features"], T("get_features"))
        assert torch.allclose(torch.sigmoid(fp["opacity"]), T("get_opacity"), atol=1e-7)
        assert torch.allclose(torch.exp(fp["scaling"]), T("get_scaling"), rtol=1e-6)
        assert torch.allclose(torch.nn.functional.normalize(fp["rotation"]), T("get_rotation"), atol=1e-6)
    assert np.allclose(g["group_lr"], [M.CKPT_LR["position_lr_init"] * M.CKPT_SCALE, M.CKPT_LR["feature_lr"],
                                       M.CKPT_LR["feature_lr"] / 20.0, M.CKPT_LR["opacity_lr"], M.CKPT_LR["scaling_lr"],
                                       M.CKPT_LR["rotation_lr"], M.CKPT_DEFORM["deform_lr_init"],
                                       M.CKPT_DEFORM["motion_coeff_lr"]], rtol=1e-12)
    # ---- (b) the reference's optimizer step from the exported state ----
    opt, params, mlp = _reference_layout_optimizer(sd["model"])
    opt.load_state_dict(sd["optim"]["optimizer"])
    assert [g_["name"] for g_ in opt.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation",
                                                        "deform_network", "motion_coeff"]
    flat = [q for g_ in opt.param_groups for q in g_["params"]]
    assert len(flat) == 77 and g["ref_group_sizes"].tolist() == [1, 1, 1, 1, 1, 1, 70, 1]
    for i, q in enumerate(flat):
        q.grad = T(f"grad_{i}")
    opt.step()
    ref_model = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                 "scaling": "_scaling", "rotation": "_rotation", "motion_coeff": "_motion_coeff"}
    for nm, key in ref_model.items():
        assert torch.allclose(params[nm].detach(), T("ref_model." + key), rtol=0, atol=1e-7), nm
    for n, q in mlp.items():
        assert torch.allclose(q.detach(), T("ref_mlp." + n), rtol=0, atol=1e-7), n
    for i, q in enumerate(flat):
        assert float(opt.state[q]["step"]) == float(g[f"ref_state_{i}.step"]) == 6.0
        assert torch.allclose(opt.state[q]["exp_avg"], T(f"ref_state_{i}.exp_avg"), rtol=0, atol=1e-9), i
        assert torch.allclose(opt.state[q]["exp_avg_sq"], T(f"ref_state_{i}.exp_avg_sq"), rtol=1e-6, atol=1e-12), i
    # ---- (c) the checkpoint the reference wrote, imported into the flat buckets ----
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "deform_network", "motion_coeff"]
    sizes = g["ref_group_sizes"].tolist()
    groups, k = [], 0
    for nm, n_, lr in zip(names, sizes, g["ref_group_lr"].tolist()):
        groups.append({"name": nm, "lr": lr, "params": list(range(k, k + n_))})
        k += n_
    state = {i: {"step": torch.tensor(float(g[f"ref_state_{i}.step"])), "exp_avg": T(f"ref_state_{i}.exp_avg"),
                 "exp_avg_sq": T(f"ref_state_{i}.exp_avg_sq")} for i in range(int(g["ref_n_state"]))}
    ref_sd = {"iteration": int(g["ref_iteration"]), "active_sh_degree": int(g["ref_active_sh_degree"]),
              "model": {k_[len("ref_model."):]: T(k_) for k_ in g.files if k_.startswith("ref_model.")},
              "optim": {"optimizer": {"state": state, "param_groups": groups}}, "spatial_lr_scale": M.CKPT_SCALE}
    ref_sd["model"]["_deform_network"] = {k_[len("ref_mlp."):]: T(k_) for k_ in g.files if k_.startswith("ref_mlp.")}
    assert set(ref_sd["model"]) == set(sd["model"])
    fp2 = CK.flat_params_from_state_dict(ref_sd, dict(fp.lr), "cpu")
    assert fp2.step_count == 6
    feats = torch.cat([params["f_dc"].detach(), params["f_rest"].detach()], dim=1)
    assert torch.allclose(fp2["features"], feats, atol=1e-7) and torch.allclose(fp2["xyz"], params["xyz"].detach(), atol=1e-7)
    o, n = fp2.offsets["motion_coeff"]
    assert torch.allclose(fp2.exp_avg[o:o + n].view(-1), opt.state[params["motion_coeff"]]["exp_avg"].reshape(-1), atol=1e-9)
    net2 = MLPBasisNetwork(128, 16, 26, False)
    net2.load_state_dict(ref_sd["model"]["_deform_network"])
    sp2 = bind_module_to_flat(net2, 0.0016, "cpu")
    assert CK.restore_deform_state(ref_sd, sp2) and sp2.step_count == 6
    for n, q in mlp.items():
        assert torch.allclose(CK._mlp_segment(sp2.flat, sp2, n), q.detach(), atol=1e-7), n
        assert torch.allclose(CK._mlp_segment(sp2.exp_avg, sp2, n), opt.state[q]["exp_avg"], atol=1e-9), n
        assert torch.allclose(CK._mlp_segment(sp2.exp_avg_sq, sp2, n), opt.state[q]["exp_avg_sq"], rtol=1e-6, atol=1e-12), n


def test_eval_pose_helpers_against_reference_golden():
    """Host pieces of the evaluator's test-time pose optimisation (rodygs_amd/pose_optimizer.py) against values produced
    by the imported reference (tests/golden/make_golden.py pose): matrix_to_quaternion incl. near-degenerate pivots,
    search_nearest_two, l2_loss, and the LearnableCamera parameterisation + its world-view matrix."""
    import numpy as np
    import torch
    from rodygs_amd.pose_optimizer import LearnablePose, l2_loss, matrix_to_quaternion, search_nearest_two
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_pose_golden.npz"))
    q = matrix_to_quaternion(torch.from_numpy(g["R"]))
    assert torch.allclose(q, torch.from_numpy(g["quat"]), rtol=0, atol=2e-6)
    assert matrix_to_quaternion(torch.from_numpy(g["R"]).reshape(8, 8, 3, 3)).shape == (8, 8, 4)
    near = search_nearest_two(torch.from_numpy(g["query_pose"]), torch.from_numpy(g["db_poses"]))
    assert near.tolist() == g["nearest"].tolist()
    assert abs(float(l2_loss(torch.from_numpy(g["l2_a"]), torch.from_numpy(g["l2_b"]))) - float(g["l2"])) < 1e-7
    cam = LearnablePose(torch.from_numpy(g["cam_R_w2c"]), torch.from_numpy(g["cam_T_w2c"]))
    assert torch.allclose(cam.R_c2w_quat.detach(), torch.from_numpy(g["cam_quat"]), atol=1e-6)
    assert torch.allclose(cam.T_c2w.detach(), torch.from_numpy(g["cam_t"]), atol=1e-6)
    assert torch.allclose(cam.world_view_transform.detach(), torch.from_numpy(g["cam_w2c"]), atol=1e-6)


def bench_metric(P, W, H):
    import bench
    return bench.metric_label(P, W, H)


def test_committed_bench_line_honours_the_contract():
    """The bench line committed under profiles/ (printed by `python bench.py` on the GPU box) carries every field of
    the measurement contract: the driver's keys, `roofline` for the dominant kernel and `cpu_baseline`; no stage is
    credited with more bytes than 8 TB/s could move in its time; the workload label is derived, not hard-coded."""
    import json
    prof = os.path.join(os.path.dirname(__file__), "..", "profiles")
    newest = os.path.exists(os.path.join(prof, "r05_bench.json"))       # (the byte accounting of the deformation changed in round 5)
    r6 = os.path.exists(os.path.join(prof, "r06_bench.json"))           # (round 6: D / D_composited, sub-records, live counters)
    d = json.load(open(os.path.join(prof, "r06_bench.json" if r6 else "r05_bench.json" if newest else "r03_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert "configs[2]" in d["config"]["workload"] and d["config"]["points"] == 1000000
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and (r["traffic"] is None or r["traffic"] > 0)
    assert isinstance(r["traffic_source"], str) and r["second_roofline"]["bound"] == "valu_issue"
    assert d["metric"] == bench_metric(d["config"]["points"], d["config"]["width"], d["config"]["height"])
    for k in ("deferred_overflow_check", "one_device", "graph_replay", "deterministic_backward", "scene"):
        assert k in d["config"], k
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert "no extrapolation" in c["sample"] and c["c2_full"]["value"] > 0 and len(c["c2_full"]["runs_s"]) == 3
    assert abs(d["value"] - d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]
    for name, st in d["stage_roofline"].items():
        assert st["hbm_frac"] is None or 0.0 < st["hbm_frac"] <= 1.0, (name, st)
    assert 0.0 < d["step_roofline"]["frac_of_8TBps"] <= 1.0
    # the step's byte count is the sum of the per-stage counts bench.stage_bytes prices
    import bench
    cfg = d["config"]
    sb = bench.stage_bytes(cfg["points"], 16, cfg["visible_V"], cfg["num_rendered_D"], cfg["height"], cfg["width"],
                           sh_adam_in_backward=cfg["sh_adam_in_backward"], radix_binning=cfg["binning"] == "radix")
    if newest:
        sb = bench.stage_bytes(cfg["points"], 16, cfg["visible_V"], cfg["num_rendered_D"], cfg["height"], cfg["width"],
                               sh_adam_in_backward=cfg["sh_adam_in_backward"], radix_binning=cfg["binning"] == "radix",
                               densify_stats=cfg["densify_stats"])
        assert sum(sb.values()) == d["step_roofline"]["algorithmic_bytes_per_step"]
        # round 5: the oracle's frame of the cpu_baseline leg is compared with the HIP frame, every tile, outside the timed region
        for name in ("bench_frame", "c2"):
            pc = d["parity_check"][name]
            assert pc["ok"] and pc["radii_equal"] and pc["D_equal"] and not pc["violations"], (name, pc)
            assert max(pc["image_max_rel"].values()) <= 1e-4 and max(pc["grad_max_rel_per_tensor"].values()) <= 1e-4
        assert d["cpu_baseline"]["host"]["cpu_count"] >= d["cpu_baseline"]["cores"] and d["cpu_baseline"]["host"]["cpu_model"]
    if r6:
        # the instances the reference's rectangles hold against the ones this build bins and composites
        assert cfg["tight_tile_rectangles"] and cfg["D_composited"] == cfg["num_rendered_D"] < 0.8 * cfg["D"]
        for name in ("bench_frame", "c2"):
            assert d["parity_check"][name]["cap"] == 5e-3
        # roofline.traffic from THIS run's counters, not a replayed file
        assert d["roofline"]["traffic_source"].startswith("LIVE") and d["roofline"]["traffic"] > d["roofline"]["algorithmic_bytes_per_launch"]
        assert d["render_valu_issue"]["source"].startswith("LIVE") and 0.3 < d["render_valu_issue"]["render_bwd"]["frac_of_issue_ceiling"] <= 1.0
        # what DESIGN.md section 5 claims next to the step, in the same line
        sr = d["sub_records"]
        assert sr["loop"]["capacity_overflows"] == 0 and sr["loop"]["sustained_over_steady"] >= 0.95 and sr["loop"]["steps"] >= 300
        assert sr["reference_iteration"]["0.5M+0.5M"]["ms_per_sub_step"] <= 1.75
        assert sr["reference_iteration"]["0.1M+0.1M"]["ms_per_sub_step"] <= 0.85
        assert sr["graph_100k"]["graph"]["ms_per_step"] > 0 and sr["psnr_delta"]["within_gate"]
        assert abs(d["psnr_delta_db"]) <= 0.05


def test_bench_accounting_follows_the_algorithm_that_runs():
    """bench.stage_bytes: optimizer-in-backward moves the SH share of the Adam bytes into the per-Gaussian backward
    kernel (and removes the dL/dshs write), bucket binning is not priced with the radix formula, sharding shrinks only
    the optimiser; bench.workload_label is derived from the arguments."""
    import bench
    P, K, V, D, H, W = 1000000, 16, 854412, 3475218, 1080, 1920
    plain = bench.stage_bytes(P, K, V, D, H, W)
    fused = bench.stage_bytes(P, K, V, D, H, W, sh_adam_in_backward=True)
    assert plain["adam"] == 28 * 75 * P and fused["adam"] == 28 * 27 * P
    assert fused["preprocess_bwd"] - plain["preprocess_bwd"] == (24 * 48 - 12 * 16) * P
    # the step's total falls by exactly the gradient round trip that no longer happens (write 4 B + read 4 B per float)
    assert sum(plain.values()) - sum(fused.values()) == 8 * 48 * P
    radix = bench.stage_bytes(P, K, V, D, H, W, radix_binning=True)
    assert plain["binning"] == D * 28 + P * 80 + 8160 * 16
    assert radix["binning"] == P * 88 + D * (8 + 20 * 2 + 4) + 8160 * 8      # depth first: 4 passes on P keys, 2 on D pairs
    assert bench.stage_bytes(P, K, V, D, H, W, world=8, sharded=True)["adam"] == 28 * 75 * (P // 8)
    assert "configs[2]" in bench.workload_label(1000000, 1920, 1080, 100, False, 1)
    assert "configs[4]" in bench.workload_label(4000000, 3840, 2160, 100, True, 1)
    assert "configs[2]" not in bench.workload_label(4000000, 3840, 2160, 100, True, 1)
    assert "not a BASELINE config" in bench.workload_label(5000, 640, 480, 10, False, 1)


def test_zcurve_row_order_is_a_permutation_along_the_curve():
    """rodygs_amd.layout: Morton codes interleave the quantised coordinates bit by bit; morton_order is a stable
    permutation; points of one octant of the bounding box come before the next octant's."""
    from rodygs_amd.layout import morton_codes, morton_order
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4096, 3, generator=g) * torch.tensor([4.0, 2.0, 9.0]) - 1.0
    codes = morton_codes(x, bits=10)
    lo, hi = x.min(0).values.double(), x.max(0).values.double()
    q = ((x.double() - lo) / (hi - lo) * 1023).round().long()

    def interleave(a, b, c):
        r = 0
        for i in range(10):
            r |= ((a >> i) & 1) << (3 * i) | ((b >> i) & 1) << (3 * i + 1) | ((c >> i) & 1) << (3 * i + 2)
        return r
    for i in range(0, 4096, 257):
        assert interleave(*q[i].tolist()) == int(codes[i])
    perm = morton_order(x, bits=10)
    assert sorted(perm.tolist()) == list(range(4096))
    assert bool((codes[perm][1:] >= codes[perm][:-1]).all())
    top = (q[perm] >> 9)                                   # octant = most significant bit of every axis
    octant = top[:, 0] + 2 * top[:, 1] + 4 * top[:, 2]
    assert bool((octant[1:] >= octant[:-1]).all())
    assert morton_order(torch.zeros(0, 3)).numel() == 0


def test_bin_mode_hint_has_hysteresis():
    """rasterizer._note_largest_tile: above BIN_RADIX_ABOVE the next frame of that (P, H, W) takes the radix path, and it
    stays there until the largest list falls below BIN_BUCKET_BELOW (no flapping around one threshold)."""
    from rodygs_amd import rasterizer as R
    key = ("test", 1080, 1920)                                   # (P, H, W): 8160 tiles
    R._BIN_HINT.pop(key, None)
    R._note_largest_tile(key, R.BIN_RADIX_ABOVE)
    assert key not in R._BIN_HINT
    R._note_largest_tile(key, R.BIN_RADIX_ABOVE + 1)
    assert R._BIN_HINT[key] == 1
    R._note_largest_tile(key, R.BIN_BUCKET_BELOW + 5)           # between the thresholds: unchanged
    assert R._BIN_HINT[key] == 1
    R._note_largest_tile(key, R.BIN_BUCKET_BELOW - 1)
    assert key not in R._BIN_HINT
    R._note_largest_tile(key, R.BIN_BUCKET_BELOW + 5)           # between the thresholds from below: still bucket
    assert key not in R._BIN_HINT
    # the same for the MEAN list length (dense frames: the per-tile sorts of bucket binning leave their fast form)
    tiles = 8160
    R._note_largest_tile(key, 3000, int(R.BIN_RADIX_MEAN_LIST_ABOVE * tiles) + tiles)
    assert R._BIN_HINT[key] == 1
    R._note_largest_tile(key, 3000, int(0.5 * (R.BIN_RADIX_MEAN_LIST_ABOVE + R.BIN_BUCKET_MEAN_LIST_BELOW) * tiles))
    assert R._BIN_HINT[key] == 1
    R._note_largest_tile(key, 3000, int(R.BIN_BUCKET_MEAN_LIST_BELOW * tiles) - tiles)
    assert key not in R._BIN_HINT


def test_measurement_scripts_and_entry_points_parse_and_stay_off_the_oracle_in_the_product():
    """Every script the profiles / DESIGN numbers come from must at least parse (they only run on the GPU box), and
    nothing under rodygs_amd/ may import the oracle (prompt rule 3: the oracle is test infrastructure)."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "scripts", "*.py")) + [os.path.join(root, "bench.py"),
                                                                os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        ast.parse(open(f).read(), filename=f)
    for f in glob.glob(os.path.join(root, "rodygs_amd", "**", "*.py"), recursive=True):
        tree = ast.parse(open(f).read(), filename=f)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{f} imports the oracle"


def test_bench_visits_the_orbit_in_bit_reversed_order():
    """bench.py's frame order (the deterministic stand-in for the reference's permutation sampler): a permutation, and any
    window of it samples the orbit evenly -- the first four of sixteen are the four quarter points."""
    import bench
    for n in (1, 2, 5, 12, 16, 100):
        assert sorted(bench._spread_order(n)) == list(range(n))
    assert bench._spread_order(16)[:4] == [0, 8, 4, 12] and bench._spread_order(16)[4:8] == [2, 10, 6, 14]


def test_anisotropic_sweep_profile_keeps_the_regular_draws():
    """tests/sweep_cases.py: the anisotropic profile changes the scales only (pancakes / needles from its own random stream);
    every other draw of the case, and the regular profile itself, stay what the named tests and profiles/r04_parity_sweep.txt
    refer to."""
    import torch
    from sweep_cases import sweep_case, sweep_case_aniso
    a, b = sweep_case(400000, 349), sweep_case_aniso(400000, 349)
    assert a[1:3] == b[1:3] and a[3] == b[3]
    for k in a[0]:
        same = torch.equal(a[0][k], b[0][k]) if torch.is_tensor(a[0][k]) else a[0][k] == b[0][k]
        assert same == (k != "scales"), k
    ratio = (b[0]["scales"].amax(1) / b[0]["scales"].amin(1))
    assert float(ratio.median()) > 10.0 and float((a[0]["scales"].amax(1) / a[0]["scales"].amin(1)).median()) < 5.0
    c = sweep_case_aniso(400000, 349)
    assert torch.equal(b[0]["scales"], c[0]["scales"])


def test_bench_gpus_flag_is_never_silently_ignored():
    """bench.py --gpus N (CPU side of the launcher logic): without a launcher and without N visible devices the command
    refuses (exit 2) instead of running one GPU and printing n_gpus = 1; under a launcher whose WORLD_SIZE disagrees with
    --gpus it refuses too; with RDG_ONE_DEVICE=1 it really starts N child ranks (which, on this GPU-less box, stop at
    "needs a GPU" -- the parent hands their failure on; the launcher ends the other rank as soon as the first one has failed, so
    how many of them got to print the message is a race: at least one, through the launcher)."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RDG_ONE_DEVICE")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    run = lambda a, e: subprocess.run([sys.executable, os.path.join(root, "bench.py"), *a], env=e, cwd=root,   # noqa: E731
                                      capture_output=True, text=True, timeout=300)
    r = run(["--gpus", "2", "--steps", "1"], env)
    assert r.returncode == 2 and "needs 2 visible devices" in r.stderr and not r.stdout.strip()
    r = run(["--gpus", "4", "--steps", "1"], dict(env, WORLD_SIZE="2", RANK="0"))
    assert r.returncode != 0 and "they must agree" in r.stderr
    r = run(["--gpus", "2", "--steps", "1"], dict(env, RDG_ONE_DEVICE="1", RDG_DIST_BACKEND="gloo"))
    assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") >= 1 and "local_rank" in r.stderr, r.stderr[-2000:]


def test_mode_switches_ride_on_the_raster_state():
    """DETERMINISTIC / _FORCE_RADIX / _FORCE_BUCKET / DEFERRED_OVERFLOW_CHECK / RENDER_NORMAL: module attributes are the
    process-wide defaults, a RasterState's own setting overrides them -- two trainers in one process can differ."""
    from rodygs_amd import rasterizer as R
    a, b = R.RasterState(), R.RasterState(deterministic=True, force_radix=True, render_normal=False)
    keep = (R.DETERMINISTIC, R._FORCE_RADIX, R.DEFERRED_OVERFLOW_CHECK, R.RENDER_NORMAL)
    try:
        R.DETERMINISTIC, R._FORCE_RADIX, R.DEFERRED_OVERFLOW_CHECK, R.RENDER_NORMAL = False, False, True, True
        assert not a.mode("deterministic") and b.mode("deterministic")
        assert a.mode("deferred_overflow_check") and b.mode("deferred_overflow_check")
        b.deferred_overflow_check = False
        assert not b.mode("deferred_overflow_check") and a.mode("deferred_overflow_check")
        rs = R.GaussianRasterizationSettings(16, 16, 1.0, 1.0, None, 1.0, None, 0, False, False, True, True)
        ca, cb = R._c_settings(rs, 10, 1, a), R._c_settings(rs, 10, 1, b)
        assert (ca.bin_mode, ca.render_normal) == (0, 1) and (cb.bin_mode, cb.render_normal) == (1, 0)
        R._FORCE_RADIX = True
        assert R._c_settings(rs, 10, 1, a).bin_mode == 1 and R._c_settings(rs, 10, 1).bin_mode == 1      # default state follows
        # a frame with a huge tile list flips the binning hint -- unless the state is pinned to bucket binning
        c = R.RasterState(force_bucket=True)
        for st in (a, c):
            st.note_largest_tile((10, 64, 64), 10 ** 6, 10 ** 6)
        assert a.bin_hint and not c.bin_hint
    finally:
        R.DETERMINISTIC, R._FORCE_RADIX, R.DEFERRED_OVERFLOW_CHECK, R.RENDER_NORMAL = keep


def test_flat_params_laid_out_in_given_storage():
    """FlatParams(storage=): the segments are views of caller-owned buffers (what densify.FlatPool hands out), the gradient
    buffer and the alignment padding are cleared, the layout equals the allocating constructor's; too small a storage raises."""
    from rodygs_amd.dp import FlatParams, FlatStorage, flat_numel
    spec = {"xyz": ((37, 3), 1e-3), "features": ((37, 16, 3), 2e-3), "opacity": ((37, 1), 5e-2)}
    n = flat_numel(spec)
    ref = FlatParams(spec, "cpu")
    assert n == ref.numel and n % 64 == 0
    st = FlatStorage(n + 1000, "cpu")
    for b in st.buffers:
        b.fill_(7.0)
    fp = FlatParams(spec, "cpu", storage=st)
    assert fp.offsets == ref.offsets and fp.numel == n and fp.storage is st
    assert fp.flat.data_ptr() == st.buffers[0].data_ptr() and fp["xyz"].grad.data_ptr() == st.buffers[1].data_ptr()
    assert float(fp.flat_grad.abs().sum()) == 0.0
    o, m = fp.offsets["xyz"]
    assert float(fp.flat[o:o + m].min()) == 7.0                     # the rows themselves are the caller's business ...
    assert float(fp.flat[o + m:fp.offsets["features"][0]].abs().sum()) == 0.0      # ... the padding behind them is cleared
    assert float(fp.exp_avg[o + m:fp.offsets["features"][0]].abs().sum()) == 0.0
    assert float(st.buffers[0][n:].min()) == 7.0                    # nothing beyond the layout is touched
    import pytest as _pt
    with _pt.raises(ValueError):
        FlatParams(spec, "cpu", storage=FlatStorage(n - 1, "cpu"))


def test_the_binding_documented_in_integration_md_matches_the_library(hip_lib):
    """INTEGRATION.md shows the ctypes binding a RoDyGS maintainer would write.  A maintainer who copies it must get THE
    struct of the library: the snippet is extracted from the document and EXECUTED up to its `def forward` -- its two guards
    (rdg_abi_version, rdg_settings_bytes) run against the built library -- and its field table is compared with the
    package's own mirror name by name, type by type.  (Round 5's snippet ended one field short: the backward would have read
    `aux_stream` past the end of the caller's struct.)"""
    import ctypes
    import re
    from rodygs_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    snippet = next(b for b in blocks if "class RdgRasterSettings(C.Structure)" in b)
    head = snippet.split("def forward(")[0]
    assert "rdg_settings_bytes" in head and "rdg_abi_version" in head, "the snippet must call the header's two guards"
    os.environ["RDG_LIB_PATH"] = _lib.LIB_PATH
    ns = {}
    exec(compile(head, "INTEGRATION.md", "exec"), ns)         # raises if a guard fails
    doc = ns["RdgRasterSettings"]
    assert ctypes.sizeof(doc) == ctypes.sizeof(_lib.RdgRasterSettings) == hip_lib.rdg_settings_bytes()
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in _lib.RdgRasterSettings._fields_]
    for (n, _t) in doc._fields_:
        assert getattr(doc, n).offset == getattr(_lib.RdgRasterSettings, n).offset, n
    # and the forward of the snippet names arguments in the header's order: 1 struct + 12 pointers + capacity + 8 pointers
    assert "[C.c_void_p] * 12 + [C.c_int64] + [C.c_void_p] * 8" in snippet


def test_birth_order_refresh_in_place_keeps_the_tensors_a_graph_knows():
    """deform.refresh_birth_order_inplace: after the birth indices were modified IN PLACE (a fixed-capacity densification) the
    sorted order, its inverse and the segment starts are recomputed INTO the tensors the cache already holds (a captured graph
    reads them by address), and the cache answers for the tensor's new version."""
    from rodygs_amd import deform
    deform.invalidate_birth_order_cache()
    ti = torch.tensor([3, 0, 2, 0, 1, 3, 2], dtype=torch.int64)
    order, inv, seg = deform._birth_order(ti, 4)
    ptrs = (order.data_ptr(), inv.data_ptr(), seg.data_ptr())
    assert ti[order.long()].tolist() == sorted(ti.tolist()) and seg.tolist() == [0, 2, 3, 5, 7]
    ti[1], ti[5] = 3, 0                                     # in place: same address, new version
    deform.refresh_birth_order_inplace(ti, 4)
    o2, i2, s2 = deform._birth_order(ti, 4)                 # a cache hit on the new version ...
    assert (o2.data_ptr(), i2.data_ptr(), s2.data_ptr()) == ptrs      # ... and the very tensors of before
    assert ti[o2.long()].tolist() == sorted(ti.tolist()) and s2.tolist() == [0, 2, 3, 5, 7]
    assert torch.equal(i2[o2.long()].long(), torch.arange(7))
    deform.invalidate_birth_order_cache()


def test_bench_preflight_reports_and_refuses_without_devices():
    """`bench.py --gpus N --preflight`: one JSON line with the devices, the peer-access matrix, RCCL and the wire bytes of both
    frame-DP formulations; exit 2 when fewer than N devices are visible (this box shows none)."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--preflight"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["preflight"] and d["gpus_requested"] == 8 and d["devices_visible"] == 0 and d["ok"] is False and "why" in d
    w = d["wire_bytes_per_step_at_1M"]
    assert w["allreduce"]["all_reduce_payload_bytes"] == 1000000 * 75 * 4 + (68656 + 700) * 4
    assert w["shard"]["all_to_all_payload_bytes"] < w["allreduce"]["all_reduce_payload_bytes"]


def test_selection_rank_and_birth_order_host_branches():
    """densify.mask_rank / _compact and deform._stable_order on CPU tensors (the branches the CPU suite can run; the library ops
    behind the GPU branches are compared with these in tests/test_gpu_round6.py): rank = position of a row in its selection,
    compacted list = ascending indices of the set rows, birth order = the stable sort."""
    import torch
    from rodygs_amd import deform
    from rodygs_amd.densify import _compact, mask_rank
    g = torch.Generator().manual_seed(3)
    for n in (1, 17, 4097):
        for density in (0.0, 0.3, 1.0):
            m = torch.rand(n, generator=g) < density
            r = mask_rank(m)
            assert r.dtype == torch.int64 and torch.equal(r, torch.cumsum(m, 0) - 1)
            k = int(m.sum())
            idx = _compact(m, k)
            assert torch.equal(idx, m.nonzero().squeeze(1))
            assert k == 0 or torch.equal(r[idx], torch.arange(k))
    t = torch.randint(0, 7, (1000,), generator=g)
    o = deform._stable_order(t, 7)
    assert torch.equal(t[o], torch.sort(t, stable=True).values) and torch.equal(o, torch.argsort(t, stable=True))
