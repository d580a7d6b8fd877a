"""Minimal RoDyGS dynamic train step around the hot path ("build re-states a minimal train
step for configs 3-5; full trainer out of scope").  It is the caller bench.py times, not a trainer replacement:
no densification, no LR schedule, photometric loss only.

One step = what /root/reference/src/trainer/rodygs.py:198-369 does for a dynamic sub-step on one camera:
  time-deformation (MLP basis in torch + HIP per-Gaussian contraction, rodygs_dynamic.py:122-138)
  -> activations (exp / sigmoid / normalize, rodygs_static.py:82-105)
  -> rasterize (HIP, renderer.py:87-101) -> 0.8 L1 + 0.2 D-SSIM (train_kubric_mrig.yaml:134-146)
  -> backward (HIP) -> [frame-DP: one all-reduce of the flat gradient bucket] -> Adam (fused HIP, eps 1e-15).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from .deform import MLPBasisNetwork, gaussian_deformation
from .dp import FlatParams, allreduce_sum_, frame_for
from .losses import fused_photometric_loss
from .model_ops import activate_gaussians, pose_view_matrix
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """Same convention as /root/reference/src/utils/graphic_utils.py:76-102 (two_s = 2/|q|^2)."""
    r, i, j, k = q[0], q[1], q[2], q[3]
    two_s = 2.0 / (q * q).sum()
    return torch.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)]).reshape(3, 3)


def world_view_transform(q_c2w: torch.Tensor, t_c2w: torch.Tensor) -> torch.Tensor:
    """FixedCameraTorch.world_view_transform (/root/reference/src/data/utils.py:161-170)."""
    R_w2c = quaternion_to_matrix(q_c2w).transpose(0, 1)
    T_w2c = -(R_w2c @ t_c2w)
    top = torch.cat([R_w2c, T_w2c.unsqueeze(1)], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=q_c2w.device, dtype=q_c2w.dtype)
    return torch.cat([top, bottom], dim=0)


def fused_adam_(fp: FlatParams, lr_scale: float = 1.0, betas=(0.9, 0.999), eps=1e-15, row_lr=None) -> None:
    """ONE fused HIP launch for all parameter groups of the flat buffers (rdg_adam_step_multi).
    row_lr: {name: (row_len, head_len, lr_tail)} for segments whose rows mix two learning rates."""
    L = _lib.lib()
    fp.step_count += 1
    segs = (_lib.RdgAdamSeg * len(fp.names))()
    for i, k in enumerate(fp.names):
        o, n = fp.offsets[k]
        b = o * 4
        row_len, head_len, lr_tail = (row_lr or {}).get(k, (1, 1, fp.lr[k]))
        segs[i].n = n
        segs[i].param = fp.flat.data_ptr() + b
        segs[i].grad = fp.flat_grad.data_ptr() + b
        segs[i].exp_avg = fp.exp_avg.data_ptr() + b
        segs[i].exp_avg_sq = fp.exp_avg_sq.data_ptr() + b
        segs[i].lr_head = fp.lr[k] * lr_scale
        segs[i].lr_tail = lr_tail * lr_scale
        segs[i].row_len = row_len
        segs[i].head_len = head_len
    _lib.check(L.rdg_adam_step_multi(len(fp.names), segs, betas[0], betas[1], eps, fp.step_count, _lib.stream_ptr()),
               "rdg_adam_step_multi")


class DynamicScene:
    """1 cloud of dynamic Gaussians + deformation MLP + per-frame learnable camera poses, synthetic data of
    SURVEY.md §8d (seeded), replicated on every rank."""

    def __init__(self, scene: dict, num_frames: int = 100, sh_degree: int = 3, device="cuda", seed: int = 777,
                 spatial_lr_scale: float = 5.0, orbit_deg: float = 15.0):
        g = torch.Generator().manual_seed(seed + 1)
        dev = torch.device(device)
        self.device = dev
        self.W, self.H = scene["W"], scene["H"]
        self.tanfovx, self.tanfovy = scene["tanfovx"], scene["tanfovy"]
        self.sh_degree = sh_degree
        self.T = num_frames
        self.spatial_lr_scale = spatial_lr_scale
        P = scene["means3D"].shape[0]
        self.P = P
        K = scene["shs"].shape[1]
        spec = {
            "xyz": ((P, 3), 0.00016 * spatial_lr_scale),
            # SH features as ONE [P,K,3] tensor (what the rasterizer consumes): no cat forward, no split backward.
            # Row-structured Adam keeps the reference's two groups: f_dc at feature_lr, f_rest at feature_lr/20.
            "features": ((P, K, 3), 0.0025),
            "scaling": ((P, 3), 0.001),
            "rotation": ((P, 4), 0.001),
            "opacity": ((P, 1), 0.05),
            "motion_coeff": ((P, 1, 16), 0.00016),
        }
        fp = FlatParams(spec, dev)
        with torch.no_grad():
            fp["xyz"].copy_(scene["means3D"])
            fp["features"].copy_(scene["shs"])
            fp["scaling"].copy_(torch.log(scene["scales"]))
            fp["rotation"].copy_(scene["rotations"])
            op = scene["opacities"].clamp(1e-4, 1 - 1e-4)
            fp["opacity"].copy_(torch.log(op / (1 - op)))
            fp["motion_coeff"].copy_(0.1 * torch.randn(P, 1, 16, generator=g))
        self.fp = fp
        self.row_lr = {"features": (K * 3, 3, 0.0025 / 20.0)}
        self.time_ind = torch.randint(0, num_frames, (P,), generator=g).to(dev)
        self.net = MLPBasisNetwork(128, 16, 26, False).to(dev)
        self.times = torch.arange(num_frames, dtype=torch.float32) / num_frames
        self.time_batch_embeddings = self.net.batch_embedding(self.times.to(dev))
        self.frame_embeddings = self.time_batch_embeddings  # frame i is rendered at time i/T (same rows)
        # cameras: orbit of +-orbit_deg about the scene centroid (z = 11 on the optical axis)
        cz = 11.0
        qs, ts = [], []
        for i in range(num_frames):
            a = math.radians(orbit_deg) * math.sin(2 * math.pi * i / num_frames)
            # camera-to-world: rotate about y by a around the centroid
            q = torch.tensor([math.cos(a / 2), 0.0, math.sin(a / 2), 0.0])
            c = torch.tensor([-cz * math.sin(a), 0.0, cz - cz * math.cos(a)])
            qs.append(q)
            ts.append(c)
        self.cam_q = torch.stack(qs).to(dev).requires_grad_(True)
        self.cam_t = torch.stack(ts).to(dev).requires_grad_(True)
        self.proj_t = scene["projmatrix"].to(dev).contiguous()   # already P^T (glm storage)
        self.bg = torch.zeros(3, device=dev)
        self.small_opt = torch.optim.Adam([
            {"params": list(self.net.parameters()), "lr": 0.0016},
            {"params": [self.cam_q], "lr": 1e-5},
            {"params": [self.cam_t], "lr": 1e-6}], eps=1e-15, fused=True)
        self.gt = {}

    # ---- pieces of the step ------------------------------------------------------------------------------------
    def settings(self) -> GaussianRasterizationSettings:
        return GaussianRasterizationSettings(self.H, self.W, self.tanfovx, self.tanfovy, self.bg, 1.0, self.proj_t,
                                             self.sh_degree, False, False, True, True)

    def gaussians_at(self, frame: int):
        """DynRoDyGS.get_gaussian_deformation + activations for the frame's time."""
        fp, net = self.fp, self.net
        # ONE pass of the MLP over the T birth-time rows + the frame's own time (row T)
        allb = net.motion_basis(torch.cat([self.time_batch_embeddings, self.frame_embeddings[frame:frame + 1]], dim=0))
        table, basis_t = allb[:-1], allb[-1]
        dxyz, drot = gaussian_deformation(fp["motion_coeff"], self.time_ind, basis_t, table, self.spatial_lr_scale)
        # activations + deformation add + feature concat: 2 HIP launches; the parameter gradients are written by
        # the backward kernel straight into the flat gradient bucket (no AccumulateGrad copies)
        sinks = {k: fp[k].grad for k in ("xyz", "scaling", "rotation", "opacity")}
        xyz, scaling, rot, opacity, _ = activate_gaussians(fp["xyz"], dxyz, fp["scaling"], fp["rotation"], drot,
                                                           fp["opacity"], None, None, grad_sinks=sinks)
        return xyz, opacity, scaling, rot, fp["features"]

    def render(self, frame: int):
        xyz, opacity, scaling, rot, feats = self.gaussians_at(frame)
        vm = pose_view_matrix(self.cam_q, self.cam_t, frame)
        m2 = torch.zeros_like(xyz, requires_grad=True)
        out = GaussianRasterizer(self.settings())(means3D=xyz, means2D=m2, shs=feats, opacities=opacity, scales=scaling,
                                                  rotations=rot, viewmatrix=vm,
                                                  grad_sinks={"shs": self.fp["features"].grad})
        return out, m2

    def make_ground_truth(self, target_scene: dict, frames):
        """GT images = HIP render of a different-seed static cloud from each frame's camera."""
        dev = self.device
        with torch.no_grad():
            for f in frames:
                vm = pose_view_matrix(self.cam_q, self.cam_t, int(f))
                z = torch.zeros_like(target_scene["means3D"])
                out = GaussianRasterizer(self.settings())(
                    means3D=target_scene["means3D"].to(dev), means2D=z.to(dev), shs=target_scene["shs"].to(dev),
                    opacities=target_scene["opacities"].to(dev), scales=target_scene["scales"].to(dev),
                    rotations=target_scene["rotations"].to(dev), viewmatrix=vm)
                self.gt[int(f)] = out[0].clamp(0, 1).clone()

    def train_step(self, step: int, rank: int = 0, world: int = 1, perm=None) -> torch.Tensor:
        perm = perm if perm is not None else list(self.gt.keys())
        frame = frame_for(step, rank, world, perm)
        # only the motion coefficients still arrive through autograd accumulation; every other segment of the
        # flat bucket is overwritten by the activation backward kernel
        self.fp.segment(self.fp.flat_grad, "motion_coeff").zero_()
        self.small_opt.zero_grad(set_to_none=True)
        out, _ = self.render(frame)
        loss = fused_photometric_loss(out[0], self.gt[frame], 0.2)
        loss.backward()
        if world > 1:
            small = [p.grad for p in self.net.parameters() if p.grad is not None]
            for t in (self.cam_q, self.cam_t):
                if t.grad is not None:
                    small.append(t.grad)
            allreduce_sum_(self.fp.flat_grad, small)
        fused_adam_(self.fp, row_lr=self.row_lr)
        self.small_opt.step()
        return loss.detach()
