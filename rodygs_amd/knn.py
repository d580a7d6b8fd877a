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


# ---- pytorch3d.ops.knn_points / knn_gather ---------------------------------------------------------------------
# Call sites: /root/reference/src/trainer/losses.py:235 (knn_points(target[None], target[None], K=8)) and
# :244-331 (knn_gather).  pytorch3d is an un-vendored dependency (reference .gitmodules:11-13); the published
# behaviour restated here: squared Euclidean distances in ascending order, int64 indices into p2, gradients of
# `dists` flow to both point sets, knn_gather(x, idx)[n, i, k] = x[n, idx[n, i, k]].
from collections import namedtuple

_KNN = namedtuple("KNN", "dists idx knn")


def _knn_forward(q: torch.Tensor, t: torch.Tensor, K: int):
    """dists [Pq,K] (squared, ascending), idx [Pq,K] of the K nearest rows of t for every row of q (float32, contiguous)."""
    L = _lib.lib()
    Pq, Pt = q.shape[0], t.shape[0]
    dists = torch.empty(Pq, K, dtype=torch.float32, device=q.device)
    idx = torch.empty(Pq, K, dtype=torch.int64, device=q.device)
    with torch.cuda.device(q.device):
        tmp = torch.empty(L.rdg_knn_tmp_bytes(Pt), dtype=torch.uint8, device=q.device)
        _lib.check(L.rdg_knn_points_forward(Pq, Pt, K, _lib.ptr(q), _lib.ptr(t), _lib.ptr(dists), _lib.ptr(idx),
                                            _lib.ptr(tmp), _lib.stream_ptr()), "rdg_knn_points_forward")
    return dists, idx


class _KnnPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, K, same):
        q = p1.detach().to(torch.float32).contiguous()
        t = q if same else p2.detach().to(torch.float32).contiguous()
        dists, idx = _knn_forward(q, t, K)
        ctx.save_for_backward(q, t, idx)
        ctx.same = same
        ctx.K = K
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    def backward(ctx, g_dists, _g_idx):
        L = _lib.lib()
        q, t, idx = ctx.saved_tensors
        g = g_dists.to(torch.float32).contiguous()
        d_q = torch.empty_like(q)
        d_t = d_q if ctx.same else torch.empty_like(t)
        with torch.cuda.device(q.device):
            _lib.check(L.rdg_knn_points_backward(q.shape[0], t.shape[0], ctx.K, _lib.ptr(q), _lib.ptr(t), _lib.ptr(idx),
                                                 _lib.ptr(g), _lib.ptr(d_q), _lib.ptr(d_t), _lib.stream_ptr()),
                       "rdg_knn_points_backward")
        return d_q, (None if ctx.same else d_t), None, None


def knn_points(p1: torch.Tensor, p2: torch.Tensor, lengths1=None, lengths2=None, norm: int = 2, K: int = 1,
               version: int = -1, return_nn: bool = False, return_sorted: bool = True):
    """pytorch3d.ops.knn_points: p1 [N,P1,3], p2 [N,P2,3] -> KNN(dists [N,P1,K], idx [N,P1,K], knn or None)."""
    if not p1.is_cuda or not p2.is_cuda:
        raise RuntimeError("rodygs_amd.knn_points: tensors must be on the GPU (no CPU fallback exists)")
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[2] != 3 or p2.shape[2] != 3 or p1.shape[0] != p2.shape[0]:
        raise ValueError("knn_points expects p1 [N,P1,3] and p2 [N,P2,3]")
    if norm != 2:
        raise ValueError("rodygs_amd.knn_points supports norm=2 only")
    if lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("rodygs_amd.knn_points: ragged batches (lengths1/lengths2) are not supported")
    ds, ix = [], []
    for n in range(p1.shape[0]):
        same = (p1 is p2) or (p1.data_ptr() == p2.data_ptr() and p1.shape == p2.shape and p1.stride() == p2.stride())
        if same:
            d, i = _KnnPoints.apply(p1[n], p1[n], K, True)
        else:
            d, i = _KnnPoints.apply(p1[n], p2[n], K, False)
        ds.append(d)
        ix.append(i)
    dists, idx = torch.stack(ds), torch.stack(ix)
    nn = knn_gather(p2, idx) if return_nn else None
    return _KNN(dists, idx, nn)


class _KnnGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        L = _lib.lib()
        xs = x.detach().to(torch.float32).contiguous()
        ii = idx.detach().to(torch.int64).contiguous()
        U = xs.shape[1]
        out = torch.empty(ii.numel(), U, dtype=torch.float32, device=xs.device)
        with torch.cuda.device(xs.device):
            _lib.check(L.rdg_knn_gather_forward(ii.numel(), U, _lib.ptr(xs), _lib.ptr(ii), _lib.ptr(out),
                                                _lib.stream_ptr()), "rdg_knn_gather_forward")
        ctx.save_for_backward(ii)
        ctx.src_shape = xs.shape
        return out.view(*idx.shape, U)

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        (ii,) = ctx.saved_tensors
        n_src, U = ctx.src_shape
        gg = g.to(torch.float32).contiguous()
        d_x = torch.empty(n_src, U, dtype=torch.float32, device=gg.device)
        with torch.cuda.device(gg.device):
            _lib.check(L.rdg_knn_gather_backward(ii.numel(), U, n_src, _lib.ptr(gg), _lib.ptr(ii), _lib.ptr(d_x),
                                                 _lib.stream_ptr()), "rdg_knn_gather_backward")
        return d_x, None


def knn_gather(x: torch.Tensor, idx: torch.Tensor, lengths=None) -> torch.Tensor:
    """pytorch3d.ops.knn_gather: x [N,M,U], idx [N,L,K] -> [N,L,K,U] with out[n,l,k] = x[n, idx[n,l,k]]."""
    if not x.is_cuda:
        raise RuntimeError("rodygs_amd.knn_gather: tensors must be on the GPU (no CPU fallback exists)")
    if x.dim() != 3 or idx.dim() != 3 or x.shape[0] != idx.shape[0]:
        raise ValueError("knn_gather expects x [N,M,U] and idx [N,L,K]")
    if lengths is not None:
        raise NotImplementedError("rodygs_amd.knn_gather: ragged batches are not supported")
    return torch.stack([_KnnGather.apply(x[n], idx[n]) for n in range(x.shape[0])])
