"""CPU restatement of the reference's time-deformation path (TEST INFRASTRUCTURE, never shipped).

Follows /root/reference/src/model/rodygs_dynamic.py:
  * TimestepEmbedder.forward            :202-220   -> time_embedding
  * MLPBasisNetwork.{timenet,basis_xyz} :267-288   -> motion_basis (functional, from a state_dict)
  * MLPBasisNetwork.forward             :308-327   -> coeff @ basis
  * DynRoDyGS.get_gaussian_deformation  :122-138   -> gaussian_deformation (inverse-motion form)
Pinned against the imported reference classes by tests/golden/deform_*.npz (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def time_embedding(t: torch.Tensor, multires: int = 26, log_sampling: bool = False) -> torch.Tensor:
    """[t, sin(t f_0), cos(t f_0), ...] with f = pi * linspace(1, 2^(m-1), m)  (rodygs_dynamic.py:202-220)."""
    if log_sampling:
        freq = 2.0 ** torch.linspace(0.0, multires - 1, multires)
    else:
        freq = torch.linspace(1.0, 2.0 ** (multires - 1), multires)
    freq = freq * math.pi
    t = t.to(torch.float32)
    emb = [t]
    for f in freq:
        emb.append(torch.sin(t * f))
        emb.append(torch.cos(t * f))
    return torch.stack(emb, dim=-1)  # [..., 2m+1]


def motion_basis(sd: dict, t_emb: torch.Tensor, num_basis: int = 16) -> torch.Tensor:
    """timenet (53->128->128->64, GELU) then 16 heads (64->32->7, GELU): [..., 53] -> [..., 16, 7]."""
    h = t_emb
    for i in (0, 2, 4):
        h = F.gelu(F.linear(h, sd[f"timenet.{i}.weight"], sd[f"timenet.{i}.bias"]))
    outs = []
    for b in range(num_basis):
        u = F.gelu(F.linear(h, sd[f"basis_xyz.{b}.basis.0.weight"], sd[f"basis_xyz.{b}.basis.0.bias"]))
        outs.append(F.linear(u, sd[f"basis_xyz.{b}.basis.2.weight"], sd[f"basis_xyz.{b}.basis.2.bias"]))
    return torch.stack(outs, dim=-2)


def gaussian_deformation(coeff: torch.Tensor, time_ind: torch.Tensor, basis_t: torch.Tensor,
                         table: torch.Tensor, spatial_lr_scale: float):
    """(scaled_translation[P,3], rotation_delta[P,4]) = c.(B(t) - B_table[birth])  (rodygs_dynamic.py:122-138)."""
    c = coeff.reshape(coeff.shape[0], -1)                       # [P,16]
    fwd = c @ basis_t                                           # [P,7]
    inv = torch.bmm(c.unsqueeze(1), table[time_ind]).squeeze(1)  # [P,7]
    d = fwd - inv
    return d[:, :3] * spatial_lr_scale, d[:, 3:]
