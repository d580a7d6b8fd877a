"""Host mirror of the reference's Pearson depth losses on the fused HIP kernels (csrc/rdg_depthloss.hip).

Reference: ``GlobalPearsonDepthLoss`` / ``LocalPearsonDepthLoss`` (/root/reference/src/trainer/losses.py:108-182) and
``pearson_depth_loss`` (/root/reference/src/utils/loss_utils.py:100-117).  Same constructors, same
``forward(pred_depth, gt_depth, motion_mask=None)``; the local loss draws its box corners exactly as the reference
does (two ``torch.randint`` calls on the GPU generator, rows then columns) and evaluates all boxes in three kernel
launches instead of a Python loop with a host synchronisation per box.  ``boxes=(rows, cols)`` overrides the draw
(used by the parity tests, whose golden vector was produced with the CPU generator)."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib


class _PearsonBoxes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, rows, cols, bh, bw, eps, weight):
        L = _lib.lib()
        if not pred.is_cuda:
            raise RuntimeError("rodygs_amd Pearson depth loss: tensors must be on the GPU (no CPU fallback exists)")
        p = pred.detach().to(torch.float32).contiguous()
        g = gt.detach().to(torch.float32).contiguous()
        H, W = p.shape[-2], p.shape[-1]
        if p.numel() != H * W or g.numel() != H * W:
            raise RuntimeError("Pearson depth loss expects [1,H,W] (or [H,W]) depth images of equal size")
        m = None if mask is None else mask.detach().to(torch.bool).contiguous()
        if m is not None and m.numel() != H * W:
            raise RuntimeError("motion mask must have the depth image's size")
        n_boxes = 1 if rows is None else int(rows.numel())
        r = None if rows is None else rows.detach().to(torch.int64).contiguous()
        c = None if cols is None else cols.detach().to(torch.int64).contiguous()
        with torch.cuda.device(p.device):
            ws = torch.empty(L.rdg_pearson_ws_bytes(n_boxes), dtype=torch.uint8, device=p.device)
            out = torch.empty(1, dtype=torch.float32, device=p.device)
            _lib.check(L.rdg_pearson_depth_forward(H, W, n_boxes, bh, bw, _lib.ptr(r), _lib.ptr(c), _lib.ptr(p), _lib.ptr(g),
                                                   _lib.ptr(m), float(eps), float(weight), _lib.ptr(ws), _lib.ptr(out),
                                                   _lib.stream_ptr()), "rdg_pearson_depth_forward")
        ctx.save_for_backward(p, g, m, r, c, ws)
        ctx.dims = (H, W, n_boxes, bh, bw)
        ctx.shape = pred.shape
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        L = _lib.lib()
        p, g, m, r, c, ws = ctx.saved_tensors
        H, W, n_boxes, bh, bw = ctx.dims
        gl = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty(H, W, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(L.rdg_pearson_depth_backward(H, W, n_boxes, bh, bw, _lib.ptr(r), _lib.ptr(c), _lib.ptr(p), _lib.ptr(g),
                                                    _lib.ptr(m), _lib.ptr(ws), _lib.ptr(gl), _lib.ptr(d),
                                                    _lib.stream_ptr()), "rdg_pearson_depth_backward")
        return d.view(ctx.shape), None, None, None, None, None, None, None, None


def _mode_mask(mode: Optional[str], motion_mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if motion_mask is None:
        return None
    if mode == "static":
        return ~motion_mask
    if mode == "dynamic":
        return motion_mask
    return None


class GlobalPearsonDepthLoss(nn.Module):
    eps = 1e-6

    def __init__(self, mode=None):
        super().__init__()
        self.mode = mode

    def forward(self, pred_depth, gt_depth, motion_mask=None, **kwargs):
        H, W = pred_depth.shape[-2], pred_depth.shape[-1]
        return _PearsonBoxes.apply(pred_depth, gt_depth, _mode_mask(self.mode, motion_mask), None, None, H, W, self.eps, 1.0)


class LocalPearsonDepthLoss(nn.Module):
    eps = 1e-6

    def __init__(self, box_p: int, p_corr: float, mode=None):
        super().__init__()
        self.box_p, self.p_corr, self.mode = box_p, p_corr, mode

    def forward(self, pred_depth, gt_depth, motion_mask=None, boxes: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                **kwargs):
        H, W = pred_depth.shape[-2], pred_depth.shape[-1]
        n_corr = int(self.p_corr * math.floor(H / self.box_p) * math.floor(W / self.box_p))
        if boxes is None:
            dev = pred_depth.device
            rows = torch.randint(0, H - self.box_p, size=(n_corr,), device=dev)      # losses.py:145 (x_0: rows)
            cols = torch.randint(0, W - self.box_p, size=(n_corr,), device=dev)      # losses.py:146 (y_0: columns)
        else:
            rows, cols = boxes
        if n_corr == 0:
            return torch.zeros((), dtype=pred_depth.dtype, device=pred_depth.device) / 0.0   # reference: 0 / n_corr
        return _PearsonBoxes.apply(pred_depth, gt_depth, _mode_mask(self.mode, motion_mask), rows, cols, self.box_p,
                                   self.box_p, self.eps, 1.0 / n_corr)
