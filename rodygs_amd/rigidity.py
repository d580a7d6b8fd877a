"""Host mirror of the reference's ``RigidityLoss`` (/root/reference/src/trainer/losses.py:185-360) on top of the HIP
K-NN ops (``rodygs_amd.knn.knn_points`` / ``knn_gather``, the pytorch3d replacement of SURVEY.md §8f row 1).

Same constructor, same ``forward(model, pred_translation)``, same random draws in the same order (``random.sample``
for the query subset, then ``torch.randint`` for the time subset), so a seeded call reproduces the reference's
value; ``tests/golden/rigidity_golden.npz`` holds what the imported reference returned for the committed inputs.
The three terms, for a random subset S of the Gaussians and its K nearest neighbours inside S (self included):

  coeff               mean over (i, k) of  w_dist * w_colour * || c_i - c_nn(i,k) ||   (l2 / l1 / cosine),
                      w_dist = exp(-lambda * d2^2) on the SQUARED neighbour distance d2, w_colour likewise on the
                      DC-colour distance
  surface             mean_i || x_i - mean_k x_nn(i,k) ||
  distance_preserving Charbonnier( || x_nn(t) - x_i(t) || , d2 ) over a random quarter of the birth times, with the
                      reference's own flattening of the [t, n, K] distances into rows of t
"""
from __future__ import annotations

import random
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import knn as _knn

_MODES = ("coeff", "surface", "distance_preserving")


def charbonnier_bc(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """CharbonnierLoss(out_norm="bc") of the reference (src/utils/loss_utils.py:206-249): sum sqrt((x-y)^2 + eps^2)
    divided by the first two dimensions of x."""
    return torch.sqrt((x - y).pow(2) + eps ** 2).sum() / (x.shape[0] * x.shape[1])


def _curve_order(p: torch.Tensor) -> torch.Tensor:
    """Permutation that sorts points [n,3] along a 30-bit Morton curve (a handful of elementwise ops + one sort)."""
    # column bounds in two stages: a reduction of millions of rows to three outputs runs in a handful of workgroups
    # (1.2 ms each at 2 M points); 1024 partial rows first, then the last step
    m = p.shape[0] // 1024 * 1024
    if m:
        q3 = p[:m].view(1024, -1, 3)
        lo, hi = q3.amin(1).amin(0), q3.amax(1).amax(0)
        if m < p.shape[0]:
            lo, hi = torch.minimum(lo, p[m:].amin(0)), torch.maximum(hi, p[m:].amax(0))
    else:
        lo, hi = p.amin(0), p.amax(0)
    q = ((p - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).to(torch.int64).clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    code = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    if code.is_cuda:
        return _sort_by_key(code, 30)[1]
    return torch.argsort(code)


def _sort_by_key(keys: torch.Tensor, end_bit: int):
    """Stable sort of non-negative int64 keys below 2^end_bit with the library's LSD radix sort (rdg_sort_pairs: one
    8-bit pass per 8 key bits, where the framework's sort walks all 64 bits of an int64): (sorted keys, permutation)."""
    from . import _lib
    L = _lib.lib()
    n = keys.numel()
    k = keys.detach().to(torch.int64).clone().contiguous()
    v = torch.arange(n, dtype=torch.int32, device=keys.device)
    if n == 0:
        return k, v.to(torch.int64)
    with torch.cuda.device(keys.device):
        nd = torch.full((1,), n, dtype=torch.int32, device=keys.device)
        tmp = torch.empty(L.rdg_sort_tmp_bytes(n), dtype=torch.uint8, device=keys.device)
        _lib.check(L.rdg_sort_pairs(_lib.ptr(k), _lib.ptr(v), n, _lib.ptr(nd), int(end_bit), _lib.ptr(tmp),
                                    _lib.stream_ptr()), "rdg_sort_pairs")
    return k, v.to(torch.int64)


class _NeighbourGraph:
    """The K-NN graph of one rigidity step in the form the fused kernels read it: the sample stored along a space-filling curve
    (a Gaussian's neighbours, and the threads next to it, touch rows that are close in memory and mostly still in the L2),
    neighbour lists in stored ids, and the reverse adjacency (edge ids s*K + k grouped by destination) that lets every
    gradient row be written once by its owner instead of through scattered float atomics.  Built once per step and shared by
    the distance-preserving term, the surface term and the backward of the neighbour search."""

    def __init__(self, pts: torch.Tensor, nn_idx: torch.Tensor):
        n, dev = pts.shape[0], pts.device
        self.n, self.K = n, nn_idx.shape[-1]
        with torch.cuda.device(dev):
            p = pts.detach().to(torch.float32).contiguous()
            self.order = _curve_order(p)                                   # stored position -> row of the sample
            self.rank = torch.empty_like(self.order)                       # row of the sample -> stored position
            self.rank[self.order] = torch.arange(n, device=dev)
            self.nbr = self.rank[nn_idx.detach().to(torch.int64)[self.order]].contiguous()
            self.X = p.index_select(0, self.order).contiguous()
            srt, self.rev_edge = _sort_by_key(self.nbr.reshape(-1), max(1, (n - 1).bit_length()))
            # offsets of every destination in the sorted edge list (searchsorted instead of bincount + cumsum: bincount reads
            # its output size back to the host and stalls the stream of launches)
            self.rev_off = torch.searchsorted(srt, torch.arange(n + 1, device=dev)).contiguous()


class _GraphSlot:
    """Filled with the step's graph once the neighbour search has run; read by that search's backward."""
    graph = None


class _KnnPointsGraph(torch.autograd.Function):
    """knn_points(pts, pts, K) whose backward goes through the step's graph (rdg_graph_points_backward) instead of 27 float
    atomics per query (rdg_knn_points_backward: 2.2 ms at n = 2 M)."""

    @staticmethod
    def forward(ctx, pts, K, slot):
        q = pts.detach().to(torch.float32).contiguous()
        dists, idx = _knn._knn_forward(q, q, K)
        ctx.save_for_backward(q, idx)
        ctx.slot, ctx.K = slot, K
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    def backward(ctx, g_dists, _g_idx):
        from . import _lib
        L = _lib.lib()
        q, idx = ctx.saved_tensors
        G = ctx.slot.graph
        g = g_dists.to(torch.float32).contiguous()
        d_q = torch.empty_like(q)
        with torch.cuda.device(q.device):
            if G is None:
                _lib.check(L.rdg_knn_points_backward(q.shape[0], q.shape[0], ctx.K, _lib.ptr(q), _lib.ptr(q), _lib.ptr(idx),
                                                     _lib.ptr(g), _lib.ptr(d_q), _lib.ptr(d_q), _lib.stream_ptr()),
                           "rdg_knn_points_backward")
            else:
                _lib.check(L.rdg_graph_points_backward(G.n, G.K, _lib.ptr(G.X), _lib.ptr(G.nbr), _lib.ptr(G.rev_off),
                                                       _lib.ptr(G.rev_edge), _lib.ptr(G.order), _lib.ptr(g), _lib.ptr(d_q),
                                                       _lib.stream_ptr()), "rdg_graph_points_backward")
        return d_q, None, None


class _FusedSurface(torch.autograd.Function):
    """sum_i || x_i - mean_k x_nn(i,k) + 1e-6 || and its gradient in two launches (rdg_graph_surface); `pts` only carries the
    gradient, the positions are the graph's."""

    @staticmethod
    def forward(ctx, pts, G):
        from . import _lib
        L = _lib.lib()
        dev = pts.device
        with torch.cuda.device(dev):
            U = torch.empty(G.n, 3, dtype=torch.float32, device=dev)
            d_pts = torch.empty(G.n, 3, dtype=torch.float32, device=dev)
            loss = torch.empty(1, dtype=torch.float64, device=dev)
            _lib.check(L.rdg_graph_surface(G.n, G.K, _lib.ptr(G.X), _lib.ptr(G.nbr), _lib.ptr(G.rev_off), _lib.ptr(G.rev_edge),
                                           _lib.ptr(G.order), _lib.ptr(U), _lib.ptr(loss), _lib.ptr(d_pts), _lib.stream_ptr()),
                       "rdg_graph_surface")
        ctx.save_for_backward(d_pts)
        return loss[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (d_pts,) = ctx.saved_tensors
        return d_pts * g, None


class _FusedDistancePreservingRows(torch.autograd.Function):
    """The distance-preserving sum on the reference's own layout: own [n,nt,3] (translation of every sampled Gaussian at the
    drawn times) and canon [n,3]; positions = canon + own.  rdg_rigidity_pack_rows + rdg_rigidity_dp_rows: a lane group per
    Gaussian, lane = time, every edge end one coalesced row; gradients come back in the caller's layout (no permute, no sum
    over times).  G: the step's graph."""

    @staticmethod
    def forward(ctx, own, canon, d2, G, eps):
        from . import _lib
        L = _lib.lib()
        n, nt, K = own.shape[0], own.shape[1], G.K
        dev = own.device
        dd = d2.detach().to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            o3 = own.detach().to(torch.float32).contiguous()
            c3 = canon.detach().to(torch.float32).contiguous()
            P3 = torch.empty(n, nt, 3, dtype=torch.float32, device=dev)
            _lib.check(L.rdg_rigidity_pack_rows(n, nt, _lib.ptr(o3), _lib.ptr(c3), _lib.ptr(G.order), _lib.ptr(P3),
                                                _lib.stream_ptr()), "rdg_rigidity_pack_rows")
            loss = torch.empty(1, dtype=torch.float64, device=dev)
            IG = torch.empty(n * K * nt, dtype=torch.float32, device=dev)
            RA = torch.empty(n * nt * 2, dtype=torch.float32, device=dev)
            G_own = torch.empty(n, nt, 3, dtype=torch.float32, device=dev)
            G_canon = torch.empty(n, 3, dtype=torch.float32, device=dev)
            d_d2 = torch.empty_like(dd)
            _lib.check(L.rdg_rigidity_dp_rows(n, K, nt, _lib.ptr(P3), _lib.ptr(G.nbr), _lib.ptr(dd), _lib.ptr(G.rev_off),
                                              _lib.ptr(G.rev_edge), _lib.ptr(G.order), _lib.ptr(G.rank), float(eps),
                                              _lib.ptr(loss), _lib.ptr(IG), _lib.ptr(RA), _lib.ptr(d_d2), _lib.ptr(G_own),
                                              _lib.ptr(G_canon), _lib.stream_ptr()), "rdg_rigidity_dp_rows")
        ctx.save_for_backward(G_own, G_canon, d_d2)
        return loss[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        G_own, G_canon, d_d2 = ctx.saved_tensors
        return G_own * g, G_canon * g, d_d2 * g, None, None


class RigidityLoss(nn.Module):
    def __init__(self, scale: float = 2, K: int = 8, sim_metric: str = "l2", dist_weight_lambda: float = 0.1,
                 color_sim: bool = True, dist_preserving_ratio=4, mode: Sequence[str] = ("coeff",),
                 knn_points=None, knn_gather=None, device_sampling: bool = False):
        super().__init__()
        for m in mode:
            if m not in _MODES:
                raise AssertionError(f"Invalid mode: {m}")
        if sim_metric not in ("l2", "l1", "cosine"):
            raise ValueError("Invalid similarity metric")
        self.scale, self.K, self.sim_metric = scale, K, sim_metric
        self.dist_weight_lambda, self.color_sim = dist_weight_lambda, color_sim
        self.dist_preserving_ratio = dist_preserving_ratio
        self.mode = list(mode)
        # the two native ops; replaceable so the CPU suite can run this module against the brute-force oracle
        self._knn_points = knn_points or _knn.knn_points
        self._knn_gather = knn_gather or _knn.knn_gather
        # False (default): the reference's own draw, random.sample on the host -- seed-for-seed reproducible against
        # it, but 0.3 s of Python per call at 1 M Gaussians.  True: the same uniform draw without replacement from
        # torch.randperm on the device.
        self.device_sampling = device_sampling
        # neighbour-search backward, surface and distance_preserving through the fused HIP kernels on the step's graph
        # whenever the native ops are in use (GPU tensors)
        self.fused_dp = knn_points is None and knn_gather is None

    def _neighbours(self, rows: torch.Tensor, nn_idx: torch.Tensor) -> torch.Tensor:
        """rows [n, ...] -> [n, K, ...]: the rows of every query's K neighbours."""
        n = rows.shape[0]
        flat = rows.reshape(1, n, -1)
        return self._knn_gather(flat, nn_idx).reshape(n, self.K, *rows.shape[1:])

    def forward(self, model, pred_translation, **kwargs):
        canon, coeff_all = model._xyz, model._motion_coeff
        moved = canon + pred_translation
        frac = 1 / self.scale if self.scale > 1 else self.scale
        n_all = len(moved)
        if self.device_sampling:
            pick = torch.randperm(n_all, device=moved.device)[:int(n_all * frac)]
        else:
            pick = torch.tensor(random.sample(range(n_all), int(n_all * frac)))      # losses.py:225-229
        # row gathers as index_select: the sample has no repeated rows, and index_select's backward is a plain
        # index_add, where advanced indexing sorts the indices first (0.6 ms per tensor at 2 M rows)
        pick = pick.to(moved.device)
        pts, coeffs = moved.index_select(0, pick), coeff_all.index_select(0, pick)
        colors = model._features_dc.index_select(0, pick) if "coeff" in self.mode else None
        n = pts.shape[0]
        graph = None
        if self.fused_dp and pts.is_cuda:
            # the native path: neighbour search, then the step's graph (curve order + reverse adjacency) that its own backward,
            # the surface term and the distance-preserving term all go through
            slot = _GraphSlot()
            d2_, idx_ = _KnnPointsGraph.apply(pts, self.K, slot)
            slot.graph = graph = _NeighbourGraph(pts, idx_)
            d2, nn_idx = d2_[None], idx_[None]
        else:
            res = self._knn_points(pts[None], pts[None], K=self.K)                    # losses.py:235
            d2, nn_idx = res.dists, res.idx                                           # [1,n,K] squared, [1,n,K]
        total = torch.tensor(0.0, dtype=torch.float32, device=pts.device)

        if "surface" in self.mode:                                                    # losses.py:241-250
            if graph is not None:
                total = total + _FusedSurface.apply(pts, graph) / n
            else:
                centre = self._neighbours(pts, nn_idx).mean(dim=1)
                total = total + F.pairwise_distance(pts, centre, p=2).mean()

        if "coeff" in self.mode:                                                      # losses.py:252-291
            c_nn = self._neighbours(coeffs, nn_idx)                                   # [n,K,1,B]
            col_nn = self._neighbours(colors.reshape(n, 3), nn_idx)                   # [n,K,3]
            col_d = F.pairwise_distance(colors[None], col_nn[None], p=2)[0]           # [n,K]
            w_dist = torch.exp(-self.dist_weight_lambda * d2[0] ** 2)
            w_col = torch.exp(-self.dist_weight_lambda * col_d ** 2)
            mine = coeffs[:, None]                                                    # [n,1,1,B]
            if self.sim_metric == "cosine":
                sim = F.cosine_similarity(mine, c_nn, dim=2)
            else:
                sim = F.pairwise_distance(mine, c_nn, p=2 if self.sim_metric == "l2" else 1)
            sim = sim.squeeze()
            sim = w_col * w_dist * sim if self.color_sim else w_dist * sim
            total = total + sim.mean()

        if "distance_preserving" in self.mode:                                        # losses.py:293-358
            times = model.unique_times
            t_idx = torch.randint(0, len(times) - 1, (len(times) // self.dist_preserving_ratio,))
            nt = len(t_idx)
            basis_xyz = model.get_motion_for_times(timesteps=None, time_indices=t_idx)[..., :3]   # [t,B,3]
            # translation of every sampled Gaussian at each drawn time, [n,t,3].  The reference writes this as a
            # broadcast matmul ([n,1,1,B] @ [t,B,3], losses.py:305-306), which the BLAS sees as n*t batched 1xB
            # products (50 M batches at config-5 size: it faulted there); one [n,B] x [B,3t] product is the same sum
            bmat = basis_xyz.permute(1, 0, 2).reshape(basis_xyz.shape[1], nt * 3)
            c2 = coeffs.reshape(n, -1)
            if c2.is_cuda and n >= 65536:
                # the same product in 1024 row blocks (batched): its gradient w.r.t. the bases is then 1024 partial
                # [B x n/1024]·[n/1024 x 3t] products summed, instead of ONE product with a 2 M-long inner dimension
                # that the BLAS runs in a handful of workgroups (2.7 ms at n = 2 M)
                nb = 1024
                pad = (-n) % nb
                cp = F.pad(c2, (0, 0, 0, pad)) if pad else c2
                own = torch.bmm(cp.view(nb, -1, c2.shape[1]), bmat.unsqueeze(0).expand(nb, -1, -1))
                own = own.view(-1, nt * 3)[:n].reshape(n, nt, 3)
            else:
                own = (c2 @ bmat).reshape(n, nt, 3)
            if graph is not None:
                dp_sum = _FusedDistancePreservingRows.apply(own, canon.index_select(0, pick), d2[0], graph, 1e-6)
                return total + dp_sum / (n * self.K * nt)
            nb = self._knn_gather(own[None].reshape(1, n, -1), nn_idx).reshape(1, n, self.K, own.shape[1], 3)
            nb = nb.squeeze().permute(2, 0, 1, 3)                                     # [t,n,K,3]
            canon_s = canon.index_select(0, pick)
            nb_loc = nb + self._neighbours(canon_s, nn_idx)[None]                     # [t,n,K,3]
            own_loc = own.transpose(0, 1)[None, :] + canon_s[None, None]              # [1,t,n,3]
            gap = torch.norm(nb_loc[None] - own_loc[:, :, :, None], dim=-1)           # [1,t,n,K]
            # the reference compares rows of `nt` consecutive entries of this [t,n,K] block with one squared
            # neighbour distance each (losses.py:352-355); kept as is
            total = total + charbonnier_bc(gap.reshape(-1, nt, 1), d2[None].reshape(-1, 1, 1))
        return total
