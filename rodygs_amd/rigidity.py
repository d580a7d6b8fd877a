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


class _FusedDistancePreserving(torch.autograd.Function):
    """sum over (tau, i, k) of Charbonnier(gap, d2_flat[f // nt]) and its gradients in one HIP kernel
    (rdg_rigidity_dp_forward): no [t,n,K,3] intermediates.  pos_t [nt,n,3] = canonical position + translation."""

    @staticmethod
    def forward(ctx, pos_t, nn_idx, d2, eps):
        from . import _lib
        L = _lib.lib()
        nt, n = pos_t.shape[0], pos_t.shape[1]
        dev = pos_t.device
        ii = nn_idx.detach().to(torch.int64).contiguous()
        dd = d2.detach().to(torch.float32).contiguous()
        K = ii.shape[-1]
        with torch.cuda.device(dev):
            # the sample is a random subset: store it along a space-filling curve so that a Gaussian's neighbours
            # (and the threads next to it) gather from nearby addresses (4.6 -> ~1.5 ms at n = 500 k, t = 25)
            order = _curve_order(pos_t[0].detach())
            rank = torch.empty_like(order)
            rank[order] = torch.arange(n, device=dev)
            ii = rank[ii[order]].contiguous()                       # neighbour lists in stored order
            p3 = pos_t.detach().to(torch.float32).contiguous()
            p4 = torch.empty(nt, n, 4, dtype=torch.float32, device=dev)
            _lib.check(L.rdg_rigidity_pack(n, nt, _lib.ptr(p3), _lib.ptr(order), _lib.ptr(p4), _lib.stream_ptr()),
                       "rdg_rigidity_pack")
            # reverse adjacency of the K-NN graph: edge ids (i*K + k) grouped by their destination
            flat = ii.reshape(-1)
            srt, rev_edge = _sort_by_key(flat, max(1, (n - 1).bit_length()))     # destinations (sorted), edge ids
            # offsets of every destination in the sorted edge list (searchsorted instead of bincount + cumsum:
            # bincount reads its output size back to the host and stalls the stream of launches)
            rev_off = torch.searchsorted(srt, torch.arange(n + 1, device=dev)).contiguous()
            loss = torch.empty(1, dtype=torch.float64, device=dev)
            G3 = torch.empty(nt, n, 3, dtype=torch.float32, device=dev)      # gradient in the caller's own row order
            d_d2 = torch.empty_like(dd)
            _lib.check(L.rdg_rigidity_dp_forward(n, K, nt, _lib.ptr(p4), _lib.ptr(ii), _lib.ptr(dd), _lib.ptr(rev_off),
                                                 _lib.ptr(rev_edge), _lib.ptr(srt), _lib.ptr(order), float(eps), _lib.ptr(loss),
                                                 None, _lib.ptr(d_d2), _lib.ptr(G3), _lib.stream_ptr()),
                       "rdg_rigidity_dp_forward")
        ctx.save_for_backward(G3, d_d2)
        return loss[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        G3, d_d2 = ctx.saved_tensors
        return G3 * g, None, d_d2 * g, None


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
        # distance_preserving through the fused HIP kernel whenever the native ops are in use (GPU tensors)
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
        res = self._knn_points(pts[None], pts[None], K=self.K)                        # losses.py:235
        d2, nn_idx = res.dists, res.idx                                               # [1,n,K] squared, [1,n,K]
        total = torch.tensor(0.0, dtype=torch.float32, device=pts.device)

        if "surface" in self.mode:                                                    # losses.py:241-250
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
            if self.fused_dp and own.is_cuda:
                pos_t = own.permute(1, 0, 2) + canon.index_select(0, pick)[None]        # [t,n,3]: one slab per time
                dp_sum = _FusedDistancePreserving.apply(pos_t, nn_idx[0], d2[0], 1e-6)
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
