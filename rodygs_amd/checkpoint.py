"""Checkpoint wire format of the reference + PSNR, so that a scene optimised on the flat
buckets can be scored by the reference's evaluator and a reference checkpoint can be loaded into the flat buckets.

Layout written by the reference (``torch.save((state_dict, iteration), ".../dynamic_last.ckpt")``,
/root/reference/src/trainer/rodygs.py:186-196) and read back by ``create_from_state_dict``
(/root/reference/src/model/rodygs_static.py:172-182, rodygs_dynamic.py:106-120) and the evaluator
(/root/reference/src/evaluator/eval.py:51-78):

    state_dict = {
      "iteration", "active_sh_degree", "spatial_lr_scale",
      "model": {"_xyz" [P,3], "_features_dc" [P,1,3], "_features_rest" [P,K-1,3], "_scaling" [P,3], "_rotation" [P,4],
                "_opacity" [P,1],  (dynamic:) "_motion_coeff" [P,1,B], "_deform_network" (MLPBasisNetwork state_dict
                with per-head keys), "_timestep" (birth time per Gaussian)},
      "optim": {"max_radii2D", "xyz_gradient_accum", "denom", "optimizer": torch.optim.Adam.state_dict() with the
                groups xyz, f_dc, f_rest, opacity, scaling, rotation (rodygs_static.py:106-141) and, for a dynamic
                model, deform_network (the MLP's 70 parameter tensors in module order) and motion_coeff, appended in
                that order (/root/reference/src/trainer/rodygs_dynamic.py:93-116)},
      "camera": {"R_c2ws_quat" [T,4], "T_c2ws" [T,3]}      # when the cameras are optimised (datamodule.py:419-424)
    }

The flat buckets keep the SH features as one [P,K,3] tensor; they are split / joined here.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .dp import FlatParams

# reference optimizer groups, in the order ThreeDGSTrainer creates them (rodygs_static.py:106-141); DynTrainer appends
# "deform_network" and then "motion_coeff" (rodygs_dynamic.py:93-116)
_GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_STACKED_HEAD = {("0", "weight"): "head_w1", ("0", "bias"): "head_b1", ("2", "weight"): "head_w2", ("2", "bias"): "head_b2"}


def reference_mlp_param_names(num_basis: int = 16) -> List[str]:
    """``[n for n, _ in MLPBasisNetwork.named_parameters()]`` of the reference class (rodygs_dynamic.py:243-288): the
    three timenet layers, then head by head (basis.0.weight, basis.0.bias, basis.2.weight, basis.2.bias).  This is the
    order of the "deform_network" optimizer group's parameter indices."""
    names = [f"timenet.{i}.{k}" for i in (0, 2, 4) for k in ("weight", "bias")]
    for b in range(num_basis):
        names += [f"basis_xyz.{b}.basis.{layer}.{kind}" for layer in ("0", "2") for kind in ("weight", "bias")]
    return names


def _mlp_segment(buf: torch.Tensor, sp: FlatParams, ref_name: str) -> torch.Tensor:
    """The slice of a small-bucket buffer (values / exp_avg / exp_avg_sq) that belongs to one reference MLP tensor;
    the bucket stores the 16 heads stacked (rodygs_amd/deform.py), the reference one module per head."""
    if ref_name.startswith("timenet."):
        return _segment(buf, sp, ref_name)
    _, b, _, layer, kind = ref_name.split(".")
    return _segment(buf, sp, _STACKED_HEAD[(layer, kind)])[int(b)]


def _segment(buf: torch.Tensor, fp: FlatParams, name: str) -> torch.Tensor:
    o, n = fp.offsets[name]
    return buf[o:o + n].view(fp.shapes[name])


def export_state_dict(fp: FlatParams, iteration: int, active_sh_degree: int, spatial_lr_scale: float,
                      deform_network: Optional[torch.nn.Module] = None, gaussian_to_time: Optional[torch.Tensor] = None,
                      cameras: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, stats=None,
                      feature_lr_rest: Optional[float] = None, deform_state: Optional[FlatParams] = None,
                      deform_lr: float = 0.0016) -> Dict:
    """The reference's checkpoint dictionary from the flat buckets (tensors are detached clones).
    ``deform_state``: the small bucket holding the MLP (rodygs_amd.trainstep.bind_module_to_flat) -- its Adam moments go
    into the "deform_network" group's state; without it the group is written with empty state (as before its first
    step)."""
    def val(name):
        return fp[name].detach().clone()

    feats = val("features")
    model = {"_xyz": val("xyz"), "_features_dc": feats[:, :1].contiguous(), "_features_rest": feats[:, 1:].contiguous(),
             "_scaling": val("scaling"), "_rotation": val("rotation"), "_opacity": val("opacity")}
    # Adam state in torch.optim.Adam.state_dict() form, one single-tensor group per name
    per_group = {
        "xyz": ("xyz", None), "opacity": ("opacity", None), "scaling": ("scaling", None), "rotation": ("rotation", None),
        "f_dc": ("features", slice(0, 1)), "f_rest": ("features", slice(1, None)),
    }
    names = list(_GROUPS)
    dynamic = "motion_coeff" in fp.offsets
    if dynamic:
        model["_motion_coeff"] = val("motion_coeff")
        per_group["motion_coeff"] = ("motion_coeff", None)
    state, groups = {}, []
    hyper = {"betas": (0.9, 0.999), "eps": 1e-15, "weight_decay": 0, "amsgrad": False}

    def one_tensor_group(i, g):
        seg, sl = per_group[g]
        a, b = _segment(fp.exp_avg, fp, seg), _segment(fp.exp_avg_sq, fp, seg)
        if sl is not None:
            a, b = a[:, sl], b[:, sl]
        if fp.step_count > 0:
            state[i] = {"step": torch.tensor(float(fp.step_count)), "exp_avg": a.detach().clone().contiguous(),
                        "exp_avg_sq": b.detach().clone().contiguous()}
        lr = fp.lr[seg] if g != "f_rest" or feature_lr_rest is None else feature_lr_rest
        groups.append({"lr": lr, "name": g, **hyper, "params": [i]})

    for i, g in enumerate(names):
        one_tensor_group(i, g)
    nxt = len(names)
    if dynamic and deform_network is not None:
        # the reference appends the MLP's parameters as ONE group before the motion coefficients
        nb = getattr(deform_network, "num_basis", 16)
        mlp_names = reference_mlp_param_names(nb)
        if deform_state is not None and deform_state.step_count > 0:
            for j, n in enumerate(mlp_names):
                state[nxt + j] = {"step": torch.tensor(float(deform_state.step_count)),
                                  "exp_avg": _mlp_segment(deform_state.exp_avg, deform_state, n).detach().clone(),
                                  "exp_avg_sq": _mlp_segment(deform_state.exp_avg_sq, deform_state, n).detach().clone()}
        groups.append({"lr": deform_lr, "name": "deform_network", **hyper,
                       "params": list(range(nxt, nxt + len(mlp_names)))})
        nxt += len(mlp_names)
    if dynamic:
        one_tensor_group(nxt, "motion_coeff")
    P = fp.shapes["xyz"][0]
    dev = fp.flat.device
    optim = {"max_radii2D": stats.max_radii2D.clone() if stats is not None else torch.zeros(P, device=dev),
             "xyz_gradient_accum": stats.xyz_gradient_accum.clone() if stats is not None else torch.zeros(P, 1, device=dev),
             "denom": stats.denom.clone() if stats is not None else torch.zeros(P, 1, device=dev),
             "optimizer": {"state": state, "param_groups": groups}}
    sd = {"iteration": int(iteration), "active_sh_degree": int(active_sh_degree), "model": model, "optim": optim,
          "spatial_lr_scale": float(spatial_lr_scale)}
    if deform_network is not None:
        model["_deform_network"] = {k: v.detach().clone() for k, v in deform_network.state_dict().items()}
    if gaussian_to_time is not None:
        model["_timestep"] = gaussian_to_time.detach().clone()
    if cameras is not None:
        sd["camera"] = {"R_c2ws_quat": cameras[0].detach().clone(), "T_c2ws": cameras[1].detach().clone()}
    return sd


def save_checkpoint(path: str, state_dict: Dict) -> None:
    """``(state_dict, iteration)`` exactly as rodygs.py:186-196 writes ``static_last.ckpt`` / ``dynamic_last.ckpt``."""
    torch.save((state_dict, state_dict["iteration"]), path)


def load_checkpoint(path: str, map_location=None) -> Dict:
    obj = torch.load(path, map_location=map_location, weights_only=False)
    return obj[0] if isinstance(obj, (tuple, list)) else obj


def flat_params_from_state_dict(sd: Dict, lrs: Dict[str, float], device, restore_optimizer: bool = True) -> FlatParams:
    """FlatParams (xyz, features, scaling, rotation, opacity[, motion_coeff]) from a reference checkpoint dictionary;
    Adam moments and the step count come from ``sd["optim"]["optimizer"]`` when present."""
    m = sd["model"]
    feats = torch.cat([m["_features_dc"], m["_features_rest"]], dim=1)
    tensors = {"xyz": m["_xyz"], "features": feats, "scaling": m["_scaling"], "rotation": m["_rotation"],
               "opacity": m["_opacity"]}
    if "_motion_coeff" in m:
        tensors["motion_coeff"] = m["_motion_coeff"]
    spec = {k: (tuple(v.shape), float(lrs.get(k, 0.0))) for k, v in tensors.items()}
    fp = FlatParams(spec, device)
    with torch.no_grad():
        for k, v in tensors.items():
            fp[k].copy_(v.detach().to(device))
    opt = sd.get("optim", {}).get("optimizer") if restore_optimizer else None
    if opt and opt.get("state"):
        by_name = {g["name"]: opt["state"].get(g["params"][0]) for g in opt["param_groups"]
                   if "name" in g and len(g["params"]) == 1}
        with torch.no_grad():
            for key, buf in (("exp_avg", fp.exp_avg), ("exp_avg_sq", fp.exp_avg_sq)):
                for name in ("xyz", "scaling", "rotation", "opacity", "motion_coeff"):
                    st = by_name.get(name)
                    if st is not None and name in fp.offsets:
                        _segment(buf, fp, name).copy_(st[key].to(device))
                dc, rest = by_name.get("f_dc"), by_name.get("f_rest")
                if dc is not None and rest is not None:
                    _segment(buf, fp, "features").copy_(torch.cat([dc[key], rest[key]], dim=1).to(device))
        steps = [float(st["step"]) for st in by_name.values() if st is not None and "step" in st]
        if steps:
            fp.step_count = int(max(steps))
    return fp


def restore_deform_state(sd: Dict, sp: FlatParams, num_basis: int = 16) -> bool:
    """Adam moments (and step count) of the "deform_network" group of a reference checkpoint -> the small bucket that
    holds the MLP.  Returns False when the checkpoint has no such group or no state for it yet."""
    opt = sd.get("optim", {}).get("optimizer") or {}
    grp = [g for g in opt.get("param_groups", []) if g.get("name") == "deform_network"]
    if not grp:
        return False
    names = reference_mlp_param_names(num_basis)
    idx = grp[0]["params"]
    if len(idx) != len(names):
        raise RuntimeError(f"deform_network group holds {len(idx)} tensors, the MLP has {len(names)}")
    st = opt.get("state", {})
    if any(i not in st for i in idx):
        return False
    with torch.no_grad():
        for i, n in zip(idx, names):
            _mlp_segment(sp.exp_avg, sp, n).copy_(st[i]["exp_avg"].to(sp.flat.device))
            _mlp_segment(sp.exp_avg_sq, sp, n).copy_(st[i]["exp_avg_sq"].to(sp.flat.device))
    sp.step_count = int(float(st[idx[0]]["step"]))
    return True


def psnr(gt_image: torch.Tensor, pred_image: torch.Tensor) -> torch.Tensor:
    """PSNR as the reference's evaluator computes it (/root/reference/src/utils/eval_utils.py:26-39): both images
    clipped to [0,1], value range 1, mean squared error over the whole image: 10 log10(1 / MSE)."""
    g = gt_image.clip(0, 1)
    p = pred_image.clip(0, 1)
    mse = ((g - p) ** 2).reshape(-1).mean()
    return 10.0 * torch.log10(1.0 / mse)
