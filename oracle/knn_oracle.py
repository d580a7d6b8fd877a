"""CPU oracle for ``simple_knn._C.distCUDA2`` (TEST INFRASTRUCTURE, never shipped).

PARITY UNPINNED: simple_knn is an un-vendored submodule (reference ``.gitmodules:5-7``); the only reference
evidence is the call site /root/reference/src/model/rodygs_static.py:130-133.  Published behaviour restated:
for every point, the MEAN of the squared Euclidean distances to its 3 nearest OTHER points.
"""
from __future__ import annotations

import numpy as np
import torch


def dist2_knn3(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 -> [P] float32, exact 3-NN via a k-d tree in float64 on the float32 coordinates."""
    from scipy.spatial import cKDTree

    p = points.detach().cpu().numpy().astype(np.float64)
    n = p.shape[0]
    k = min(4, n)
    d, _ = cKDTree(p).query(p, k=k)
    d = np.atleast_2d(d)
    d2 = d[:, 1:] ** 2  # drop self (distance 0)
    out = np.zeros(n, dtype=np.float64)
    if d2.shape[1] > 0:
        out = d2.sum(axis=1) / 3.0
    return torch.from_numpy(out.astype(np.float32))


def knn_points(p1: torch.Tensor, p2: torch.Tensor, K: int):
    """pytorch3d.ops.knn_points restatement: squared dists ascending [N,K] + idx [N,K] (brute force)."""
    d = torch.cdist(p1.double(), p2.double()) ** 2
    dist, idx = torch.topk(d, K, dim=1, largest=False, sorted=True)
    return dist.float(), idx


# ---- pytorch3d.ops.knn_points / knn_gather restated (batched, differentiable; brute force) ---------------------
# PARITY UNPINNED: pytorch3d is an un-vendored dependency (reference .gitmodules:11-13).  Published behaviour:
# knn_points(p1 [N,P1,D], p2 [N,P2,D], K) -> dists [N,P1,K] SQUARED Euclidean, ascending, idx [N,P1,K] int64;
# knn_gather(x [N,M,U], idx [N,L,K]) -> [N,L,K,U].  Call sites: /root/reference/src/trainer/losses.py:235-331.
from collections import namedtuple

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points_batched(p1: torch.Tensor, p2: torch.Tensor, K: int = 1, **_unused):
    diff = p1[:, :, None, :] - p2[:, None, :, :]
    d2 = (diff * diff).sum(-1)
    dists, idx = torch.topk(d2, K, dim=2, largest=False, sorted=True)
    return _KNN(dists, idx, None)


def knn_gather(x: torch.Tensor, idx: torch.Tensor, lengths=None) -> torch.Tensor:
    N, L, K = idx.shape
    U = x.shape[2]
    flat = idx.reshape(N, L * K, 1).expand(N, L * K, U)
    return torch.gather(x, 1, flat).reshape(N, L, K, U)
