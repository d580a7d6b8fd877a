"""CPU oracle for densify-and-prune (TEST INFRASTRUCTURE, never shipped or imported by the product path).

Restates, tensor by tensor and in the reference's own order of operations, what
/root/reference/src/trainer/rodygs_static.py:170-319 (+ the dynamic overrides rodygs_dynamic.py:150-197) and
/root/reference/src/trainer/utils.py:15-95 do to the parameters, the Adam moments, the densification statistics
and the per-Gaussian time arrays: clone (cat), split (cat, then mask out the parents), prune (mask).

PARITY: the reference trainer cannot be instantiated here (it needs the rasterizer, a datamodule and CUDA), so this
is a restatement of Python code that is present in /root/reference, checked by reading, not by running it: parity of
this component is pinned to the cited lines, not to reference outputs.  The only deviation: the split samples are
``exp(scaling) * z`` with the standard-normal ``z`` passed in, where the reference calls torch.normal(0, std).
"""
from __future__ import annotations

from typing import Dict

import torch


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """/root/reference/src/utils/general_utils.py:92-115 (normalises the quaternion)."""
    q = r / torch.sqrt((r * r).sum(dim=1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros(q.shape[0], 3, 3, dtype=r.dtype)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class State:
    """params / exp_avg / exp_avg_sq: {name: [P,...]}; stats and per-point arrays as in the trainer."""

    def __init__(self, params, exp_avg, exp_avg_sq, accum, denom, max_radii, per_point):
        self.params, self.exp_avg, self.exp_avg_sq = params, exp_avg, exp_avg_sq
        self.accum, self.denom, self.max_radii, self.per_point = accum, denom, max_radii, per_point

    @property
    def P(self):
        return self.params["xyz"].shape[0]


def _cat(st: State, new: Dict[str, torch.Tensor]):
    """cat_tensors_to_optimizer + densification_postfix (utils.py:36-69, rodygs_static.py:170-180)."""
    for k in st.params:
        st.exp_avg[k] = torch.cat([st.exp_avg[k], torch.zeros_like(new[k])], dim=0)
        st.exp_avg_sq[k] = torch.cat([st.exp_avg_sq[k], torch.zeros_like(new[k])], dim=0)
        st.params[k] = torch.cat([st.params[k], new[k]], dim=0)
    st.accum = torch.zeros(st.P, 1)
    st.denom = torch.zeros(st.P, 1)
    st.max_radii = torch.zeros(st.P)


def _prune(st: State, mask: torch.Tensor):
    """prune_points (rodygs_static.py:303-315, utils.py:72-95)."""
    valid = ~mask
    for k in st.params:
        st.params[k] = st.params[k][valid]
        st.exp_avg[k] = st.exp_avg[k][valid]
        st.exp_avg_sq[k] = st.exp_avg_sq[k][valid]
    st.accum, st.denom, st.max_radii = st.accum[valid], st.denom[valid], st.max_radii[valid]
    st.per_point = {k: v[valid] for k, v in st.per_point.items()}


def densify_and_prune(st: State, max_grad, min_opacity, extent, max_screen_size, percent_dense, N, z, decisions=None):
    """rodygs_static.py:280-301; returns the number of clone / split selections.  ``decisions`` ({"clone", "split",
    "prune"} boolean masks, the layout of rodygs_amd.densify.DensifyResult.decisions) replaces the three threshold
    tests -- used by scripts/psnr_delta.py to hold the set of Gaussians fixed across runs."""
    grads = st.accum / st.denom
    grads[grads.isnan()] = 0.0
    # ---- clone (:244-277) ----
    scaling = torch.exp(st.params["scaling"])
    sel = (torch.norm(grads, dim=-1) >= max_grad) & (scaling.max(dim=1).values <= percent_dense * extent)
    if decisions is not None:
        sel = decisions["clone"]
    new = {k: v[sel] for k, v in st.params.items()}
    st.per_point = {k: torch.cat([v, v[sel]]) for k, v in st.per_point.items()}
    n_clone = int(sel.sum())
    _cat(st, new)
    # ---- split (:182-242) ----
    padded = torch.zeros(st.P)
    padded[:grads.shape[0]] = grads.squeeze()
    scaling = torch.exp(st.params["scaling"])
    sel = (padded >= max_grad) & (scaling.max(dim=1).values > percent_dense * extent)
    if decisions is not None:
        sel = torch.cat([decisions["split"], torch.zeros(st.P - decisions["split"].shape[0], dtype=torch.bool)])
    n_sel = int(sel.sum())
    stds = scaling[sel].repeat(N, 1)
    samples = stds * z[:stds.shape[0]]
    rots = build_rotation(st.params["rotation"][sel]).repeat(N, 1, 1)
    new = {k: v[sel].repeat(N, *([1] * (v.dim() - 1))) for k, v in st.params.items()}
    new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + st.params["xyz"][sel].repeat(N, 1)
    new["scaling"] = torch.log(scaling[sel].repeat(N, 1) / (0.8 * N))
    _cat(st, new)
    st.per_point = {k: torch.cat([v] + [v[sel] for _ in range(N)]) for k, v in st.per_point.items()}
    _prune(st, torch.cat([sel, torch.zeros(N * n_sel, dtype=torch.bool)]))
    # ---- prune (:286-298) ----
    mask = (torch.sigmoid(st.params["opacity"]) < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = st.max_radii > max_screen_size
        big_ws = torch.exp(st.params["scaling"]).max(dim=1).values > 0.1 * extent
        mask = mask | big_vs | big_ws
    if decisions is not None:
        mask = decisions["prune"]
    _prune(st, mask)
    return n_clone, n_sel
