"""``simple_knn._C.distCUDA2`` replacement (call site /root/reference/src/model/rodygs_static.py:130-133)."""
from __future__ import annotations

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest other points.  points [P,3] float32 on the GPU."""
    if not points.is_cuda:
        raise RuntimeError("rodygs_amd.distCUDA2: points must be a CUDA/HIP tensor (no CPU fallback exists)")
    L = _lib.lib()
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("distCUDA2 expects [P,3]")
    P = pts.shape[0]
    out = torch.zeros(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    with torch.cuda.device(pts.device):
        tmp = torch.empty(L.rdg_knn_tmp_bytes(P), dtype=torch.uint8, device=pts.device)
        _lib.check(L.rdg_dist2_knn3(P, _lib.ptr(pts), _lib.ptr(out), _lib.ptr(tmp), _lib.stream_ptr()),
                   "rdg_dist2_knn3")
    return out
