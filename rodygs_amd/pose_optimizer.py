"""Test-time camera-pose optimisation of the reference evaluator, on the HIP rasterizer's
pose gradient.

Reference: ``PoseOptimizer`` (/root/reference/src/evaluator/eval.py:342-420) -- for a held-out frame, start from the
CALIBRATED train pose whose ground-truth counterpart is nearest to the frame's ground-truth pose
(``search_nearest_two``, src/evaluator/utils.py:15-26), wrap it in a ``LearnableCamera`` (camera-to-world quaternion +
translation as parameters, src/data/utils.py:173-232) and run ``num_opts`` steps of Adam(lr=camera_lr, eps=1e-15) on
``l2_loss(render(camera), rgb)`` (src/utils/loss_utils.py:23-24) with the Gaussians frozen.  The only gradient that
matters is dL/dviewmatrix, which this build's rasterizer returns (``enable_cov_grad`` / ``enable_sh_grad`` gates
included); the pose -> W2C^T map and its backward are one HIP launch each (``model_ops.pose_view_matrix``).

The host pieces (``matrix_to_quaternion``, ``search_nearest_two``, ``l2_loss``, the parameterisation) are pinned by
tests/golden/eval_pose_golden.npz, generated from the imported reference.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """[...,3,3] rotation matrices -> [...,4] quaternions (real part first), the component of largest magnitude made
    positive -- the convention of the reference's helper (src/utils/graphic_utils.py:116-159).

    For each candidate pivot k in (r, i, j, k): 4 q_k^2 = 1 +- m00 +- m11 +- m22, and the other three components follow
    from the off-diagonal sums / differences divided by 4 q_k; the pivot with the largest q_k is the well-conditioned
    one."""
    if matrix.shape[-2:] != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    lead = matrix.shape[:-2]
    m = matrix.reshape(-1, 3, 3)
    d0, d1, d2 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    four_sq = torch.stack([1 + d0 + d1 + d2, 1 + d0 - d1 - d2, 1 - d0 + d1 - d2, 1 - d0 - d1 + d2], dim=1)
    two_abs = torch.sqrt(four_sq.clamp_min(0.0))                       # 2 |q_k|
    pivot = two_abs.argmax(dim=1)
    a = m[:, 2, 1] - m[:, 1, 2]      # 4 r i
    b = m[:, 0, 2] - m[:, 2, 0]      # 4 r j
    c = m[:, 1, 0] - m[:, 0, 1]      # 4 r k
    e = m[:, 1, 0] + m[:, 0, 1]      # 4 i j
    f = m[:, 0, 2] + m[:, 2, 0]      # 4 i k
    g = m[:, 1, 2] + m[:, 2, 1]      # 4 j k
    sq = two_abs * two_abs
    rows = torch.stack([torch.stack([sq[:, 0], a, b, c], dim=1), torch.stack([a, sq[:, 1], e, f], dim=1),
                        torch.stack([b, e, sq[:, 2], g], dim=1), torch.stack([c, f, g, sq[:, 3]], dim=1)], dim=1)
    picked = rows[torch.arange(m.shape[0], device=m.device), pivot]                        # 4 q_pivot * q
    denom = 2.0 * two_abs.gather(1, pivot[:, None]).clamp_min(0.1)
    return (picked / denom).reshape(*lead, 4)


def search_nearest_two(query_pose: torch.Tensor, db_poses: torch.Tensor) -> torch.Tensor:
    """Indices of the two poses of ``db_poses`` [N,4,4] whose translations are nearest to ``query_pose``'s [4,4]."""
    d = torch.norm(query_pose[None, :3, 3] - db_poses[:, :3, 3], dim=1)
    return torch.topk(d, k=2, largest=False).indices


def l2_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return ((network_output - gt) ** 2).mean()


class LearnablePose(nn.Module):
    """The two parameters of the reference's ``LearnableCamera``: camera-to-world quaternion and translation, built from
    a world-to-camera rotation / translation exactly as its constructor does."""

    def __init__(self, R_w2c: torch.Tensor, T_w2c: torch.Tensor):
        super().__init__()
        R_c2w = R_w2c.transpose(0, 1)
        self.R_c2w_quat = nn.Parameter(matrix_to_quaternion(R_c2w).detach().clone())
        self.T_c2w = nn.Parameter((-(R_c2w @ T_w2c)).detach().clone())

    @property
    def world_view_transform(self) -> torch.Tensor:
        """W2C [4,4] in plain torch (any device): the reference's property, for inspection and the golden test."""
        from .motion_losses import quaternion_to_matrix
        Rt = quaternion_to_matrix(self.R_c2w_quat).transpose(0, 1)
        top = torch.cat([Rt, (-(Rt @ self.T_c2w)).unsqueeze(1)], dim=1)
        return torch.cat([top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=top.dtype, device=top.device)], dim=0)

    def viewmatrix(self) -> torch.Tensor:
        """W2C^T (glm storage), differentiable w.r.t. both parameters through the HIP op."""
        from .model_ops import pose_view_matrix
        return pose_view_matrix(self.R_c2w_quat.unsqueeze(0), self.T_c2w.unsqueeze(0), 0)


class PoseOptimizer:
    """``PoseOptimizer(calibrated_poses, uncalibrated_poses, render, camera_lr, num_opts)(gt_pose, rgb)``.

    ``calibrated_poses`` [N,4,4]: the trained camera-to-world poses; ``uncalibrated_poses`` [N,4,4]: the ground-truth
    poses of the same train frames (the evaluator reads them from train_transforms.json); ``render(viewmatrix)``: the
    frozen scene rendered from a W2C^T matrix, returning the image [3,H,W] (e.g. a closure over
    ``GaussianRasterizer(settings)(..., viewmatrix=vm)[0]``)."""

    def __init__(self, calibrated_poses: torch.Tensor, uncalibrated_poses: torch.Tensor,
                 render: Callable[[torch.Tensor], torch.Tensor], camera_lr: float, num_opts: int):
        self.calibrated_poses, self.uncalibrated_poses = calibrated_poses, uncalibrated_poses
        self.render, self.camera_lr, self.num_opts = render, camera_lr, num_opts
        self.history = []

    def __call__(self, gt_pose: torch.Tensor, rgb: torch.Tensor) -> LearnablePose:
        near = search_nearest_two(gt_pose, self.uncalibrated_poses)
        init = self.calibrated_poses[near[0]].detach().clone().to(torch.float32)          # camera-to-world
        w2c = torch.inverse(init)
        cam = LearnablePose(w2c[:3, :3], w2c[:3, 3]).to(rgb.device)
        opt = torch.optim.Adam(cam.parameters(), lr=self.camera_lr, eps=1e-15)
        self.history = []
        for _ in range(self.num_opts):
            loss = l2_loss(self.render(cam.viewmatrix()), rgb)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            self.history.append(loss.detach())
        return cam
