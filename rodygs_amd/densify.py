"""Densify-and-prune over the flat parameter buckets.

Reference: ``ThreeDGSTrainer.densify_and_prune`` and helpers (/root/reference/src/trainer/rodygs_static.py:170-319,
dynamic overrides /root/reference/src/trainer/rodygs_dynamic.py:150-197) + the optimizer surgery of
/root/reference/src/trainer/utils.py:15-95.  The reference rebuilds every parameter tensor and both Adam moments
three times per call (clone -> cat, split -> cat + mask, prune -> mask).  Here the three steps are composed into one
list of source rows, and each buffer of the flat buckets (parameters, exp_avg, exp_avg_sq) is rebuilt by ONE HIP
gather (csrc/rdg_densify.hip); the split children are placed by one more small kernel.  Result rows are in the
reference's order: surviving originals, clones, split children (N copies, ``repeat`` order), then the final prune.

Reference quirks kept on purpose:
* ``densification_postfix`` zeroes ``max_radii2D`` before the final prune, so the screen-size criterion never fires
  there; only the world-size criterion (``> 0.1 * extent``) does when ``max_screen_size`` is truthy.
* Clones are appended before the split selection runs; their padded gradient is 0, so they are never split.
* New Gaussians start with zero Adam moments; survivors keep theirs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib
from .dp import FlatParams


@dataclass
class DensifyStats:
    xyz_gradient_accum: torch.Tensor   # [P,1]
    denom: torch.Tensor                # [P,1]
    max_radii2D: torch.Tensor          # [P]

    @staticmethod
    def zeros(P: int, device) -> "DensifyStats":
        return DensifyStats(torch.zeros(P, 1, device=device), torch.zeros(P, 1, device=device),
                            torch.zeros(P, device=device))

    def add(self, viewspace_grad: torch.Tensor, update_filter: torch.Tensor, radii: Optional[torch.Tensor] = None):
        """The reference's own expression -- add_densification_stats (rodygs_static.py:317-319) with the gradient norm of
        rodygs.py:319-341 and the running max of the screen radii (rodygs.py:334-337) -- five boolean-mask indexing ops,
        each a nonzero + host sync.  Kept as the statement the parity tests compare the kernels with; the train step uses
        ``sink()`` (inside the per-Gaussian backward kernel) or ``add_frame`` (one launch)."""
        g = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum[update_filter] += g[update_filter]
        self.denom[update_filter] += 1
        if radii is not None:
            self.max_radii2D[update_filter] = torch.max(self.max_radii2D[update_filter],
                                                        radii[update_filter].to(self.max_radii2D.dtype))

    def add_frame(self, viewspace_grad: torch.Tensor, radii: torch.Tensor, row0: int = 0) -> None:
        """One iteration's update as ONE HIP launch (rdg_densify_stats), for a caller that follows the reference's flow
        (``viewspace_point_tensor.grad`` and ``radii`` in hand after ``loss.backward()``): rows [row0, row0 + P_stats) of
        the concatenated cloud, visible = radii > 0 (the reference's ``visibility_filter``).  No host sync."""
        if not (viewspace_grad.is_cuda and radii.is_cuda and self.denom.is_cuda):
            raise RuntimeError("DensifyStats.add_frame: tensors must be on the GPU (no CPU fallback exists)")
        n = self.denom.numel()
        g = viewspace_grad.detach()
        if g.dtype != torch.float32 or not g.is_contiguous() or g.dim() != 2 or g.shape[1] != 3:
            raise RuntimeError("DensifyStats.add_frame: viewspace gradient must be a contiguous float32 [P,3] tensor")
        if radii.dtype != torch.int32 or not radii.is_contiguous():
            raise RuntimeError("DensifyStats.add_frame: radii must be the rasterizer's contiguous int32 [P] tensor")
        if row0 < 0 or row0 + n > radii.numel() or g.shape[0] != radii.numel():
            raise RuntimeError("DensifyStats.add_frame: the statistics rows do not lie inside the rendered cloud")
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().rdg_densify_stats(n, row0, _lib.ptr(g), _lib.ptr(radii),
                                                    _lib.ptr(self.xyz_gradient_accum), _lib.ptr(self.denom),
                                                    _lib.ptr(self.max_radii2D), _lib.stream_ptr()), "rdg_densify_stats")

    def sink(self, row0: int = 0) -> dict:
        """``grad_sinks["densify"]`` entry for the rasterizer: the per-Gaussian backward kernel applies the update itself
        (no launch, no extra pass: dL/dmean2D and the radius are in its registers)."""
        return {"grad_accum": self.xyz_gradient_accum, "denom": self.denom, "max_radii": self.max_radii2D,
                "row0": int(row0), "rows": int(self.denom.numel())}


def allreduce_stats_(stats: "DensifyStats") -> None:
    """Frame-DP: every rank saw a different camera, so before densifying the statistics are combined
    -- gradient accumulators and visit counts summed, screen radii maxed -- and every rank then takes the same
    clone / split / prune decisions from identical data."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.all_reduce(stats.xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(stats.denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(stats.max_radii2D, op=dist.ReduceOp.MAX)


@dataclass
class DensifyResult:
    fp: FlatParams
    stats: DensifyStats
    per_point: Dict[str, torch.Tensor]
    n_clone: int
    n_split: int
    n_pruned: int
    decisions: Optional[Dict[str, torch.Tensor]] = None   # the clone / split / prune masks that were applied


def _gather_rows(src_flat: torch.Tensor, row_len: int, idx: torch.Tensor, dst_flat: torch.Tensor) -> None:
    _lib.check(_lib.lib().rdg_gather_rows(idx.numel(), row_len, _lib.ptr(idx), _lib.ptr(src_flat), _lib.ptr(dst_flat),
                                          _lib.stream_ptr()), "rdg_gather_rows")


class FlatPool:
    """Two sets of flat buffers used in turn by successive densifications: the rows are gathered from the set the cloud
    lives in into the spare one, which then becomes the cloud's -- no allocation per densification (a new gigabyte-sized
    block every 100 steps stalled the host for 36 ms at 1 M Gaussians: bench.py --loop-profile host).  A set that has become
    too small is replaced by one with ``headroom`` (x the needed size), so that the next few densifications fit."""

    def __init__(self, headroom: float = 1.3):
        self.headroom, self.spare = float(headroom), None

    def take(self, numel: int, device):
        from .dp import FlatStorage
        st, self.spare = self.spare, None
        if st is None or st.capacity < numel or st.buffers[0].device != torch.device(device):
            st = FlatStorage(int(numel * self.headroom) + 4096, device)
        return st

    def give_back(self, fp: FlatParams) -> None:
        """The storage ``fp`` lived in (if it has one of its own) is the next spare; ``fp`` must not be used any more."""
        if fp.storage is not None and (self.spare is None or fp.storage.capacity > self.spare.capacity):
            self.spare = fp.storage


def rebuild_flat_params(fp: FlatParams, src: torch.Tensor, keep_moments: torch.Tensor, pool: Optional[FlatPool] = None) -> FlatParams:
    """New FlatParams whose row i of every segment is row src[i] of ``fp``; Adam moments are carried over where
    ``keep_moments[i]`` and start at zero elsewhere.  Every segment's first dimension is the Gaussian count.
    ``pool``: take the new buffers from it instead of allocating (the caller hands ``fp``'s storage back afterwards)."""
    from .dp import flat_numel
    dev = fp.flat.device
    n_new = int(src.numel())
    spec = {k: ((n_new, *fp.shapes[k][1:]), fp.lr[k]) for k in fp.names}
    out = FlatParams(spec, dev, storage=None if pool is None else pool.take(flat_numel(spec), dev))
    out.step_count = fp.step_count
    src = src.to(torch.int64).contiguous()
    src_m = torch.where(keep_moments, src, torch.full_like(src, -1)).contiguous()
    with torch.cuda.device(dev):
        for k in fp.names:
            row_len = 1
            for s_ in fp.shapes[k][1:]:
                row_len *= int(s_)
            so, sn = fp.offsets[k]
            do, dn = out.offsets[k]
            for a, b, idx in ((fp.flat, out.flat, src), (fp.exp_avg, out.exp_avg, src_m),
                              (fp.exp_avg_sq, out.exp_avg_sq, src_m)):
                _gather_rows(a[so:so + sn], row_len, idx, b[do:do + dn])
    return out


def _densify_general(fp: FlatParams, stats: DensifyStats, per_point: Dict[str, torch.Tensor], max_grad: float,
                      min_opacity: float, extent: float, max_screen_size, percent_dense: float = 0.01, N: int = 2,
                      z: Optional[torch.Tensor] = None, decisions: Optional[Dict[str, torch.Tensor]] = None,
                      spatial_order: bool = False) -> DensifyResult:
    """The composed-source-row form with the three masks taken from ``decisions`` when given (the replay path of
    ``densify_and_prune``; several host read-backs: sizes of boolean selections)."""
    if not fp.flat.is_cuda:
        raise RuntimeError("rodygs_amd.densify_and_prune: buffers must be on the GPU (no CPU fallback exists)")
    if fp.shapes["scaling"][1:] != (3,):
        raise NotImplementedError("isotropic scaling ([P,1]) is not supported")
    dev = fp.flat.device
    P = fp.shapes["xyz"][0]
    with torch.no_grad():
        grads = stats.xyz_gradient_accum / stats.denom
        grads[grads.isnan()] = 0.0
        max_s = torch.exp(fp["scaling"].detach()).max(dim=1).values
        small = max_s <= percent_dense * extent
        clone_mask = (torch.norm(grads, dim=-1) >= max_grad) & small                    # rodygs_static.py:244-251
        split_mask = (grads.squeeze(-1) >= max_grad) & ~small                            # :185-193 (clones: grad 0)
        if decisions is not None:
            clone_mask, split_mask = decisions["clone"].to(dev), decisions["split"].to(dev)
        idx_all = torch.arange(P, device=dev)
        idx_clone = idx_all[clone_mask]
        idx_split = idx_all[split_mask]
        n_clone, n_sel = int(idx_clone.numel()), int(idx_split.numel())
        idx_child = idx_split.repeat(N)                                                  # repeat(N, 1) order
        n_child = n_sel * N
        src = torch.cat([idx_all[~split_mask], idx_clone, idx_child])
        kind = torch.cat([torch.zeros(P - n_sel, dtype=torch.int64, device=dev),
                          torch.ones(n_clone, dtype=torch.int64, device=dev),
                          torch.full((n_child,), 2, dtype=torch.int64, device=dev)])
        child_no = torch.cat([torch.full((P - n_sel + n_clone,), -1, dtype=torch.int64, device=dev),
                              torch.arange(n_child, device=dev)])
        # final prune on the rows as they stand after clone + split (rodygs_static.py:286-298)
        opac = torch.sigmoid(fp["opacity"].detach().reshape(-1))[src]
        scale_now = torch.where(kind == 2, max_s[src] / (0.8 * N), max_s[src])
        prune = opac < min_opacity
        if max_screen_size:
            prune = prune | (scale_now > 0.1 * extent)                                   # max_radii2D was just zeroed
        if decisions is not None:
            prune = decisions["prune"].to(dev)
        used = {"clone": clone_mask.clone(), "split": split_mask.clone(), "prune": prune.clone()}
        keep = ~prune
        src, kind, child_no = src[keep], kind[keep], child_no[keep]
        out = rebuild_flat_params(fp, src, kind == 0)
        # split children: contiguous tail of the new buffers
        n_child_kept = int((kind == 2).sum())
        if n_child_kept:
            if z is None:
                z = torch.randn(n_child, 3, device=dev)
            zz = z.to(device=dev, dtype=torch.float32)[child_no[kind == 2]].contiguous()
            parents = src[kind == 2].contiguous()
            n_new = int(src.numel())
            first = n_new - n_child_kept
            xo, so_ = out["xyz"].detach()[first:], out["scaling"].detach()[first:]
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().rdg_split_children(n_child_kept, N, _lib.ptr(parents), _lib.ptr(fp["xyz"].detach()),
                                                         _lib.ptr(fp["scaling"].detach()),
                                                         _lib.ptr(fp["rotation"].detach()), _lib.ptr(zz), _lib.ptr(xo),
                                                         _lib.ptr(so_), _lib.stream_ptr()), "rdg_split_children")
        new_pp = {k: v[src] for k, v in per_point.items()}
        if spatial_order:
            from .layout import morton_order
            perm = morton_order(out["xyz"].detach())
            out = rebuild_flat_params(out, perm, torch.ones_like(perm, dtype=torch.bool))
            new_pp = {k: v[perm] for k, v in new_pp.items()}
        from .deform import invalidate_birth_order_cache
        invalidate_birth_order_cache()       # the per-point tensors (birth indices) are new objects from here on
        n_new = int(src.numel())
        return DensifyResult(out, DensifyStats.zeros(n_new, dev), new_pp, n_clone, n_sel,
                             int(prune.sum()) + n_sel, used)



class _Phase:
    """Optional per-phase wall times of a densification (``timings`` dict): a device synchronisation at every phase boundary,
    so only for diagnosis (bench.py --loop-profile); without a dict nothing is synchronised."""

    def __init__(self, timings, dev):
        self.t, self.dev, self.last = timings, dev, None
        # timings["_nosync"]: HOST time per phase only (where does the host block?), no device synchronisation
        self.sync = timings is not None and not timings.get("_nosync")
        if timings is not None:
            import time
            self.clock = time.perf_counter
            if self.sync:
                torch.cuda.synchronize(dev)
            self.last = self.clock()

    def mark(self, name: str) -> None:
        if self.t is None:
            return
        if self.sync:
            torch.cuda.synchronize(self.dev)
        now = self.clock()
        self.t[name] = self.t.get(name, 0.0) + (now - self.last) * 1e3
        self.last = now


def mask_rank(mask: torch.Tensor) -> torch.Tensor:
    """int64 [P]: rank[i] = (number of set entries of the bool mask in [0, i]) - 1, the position of row i in the selection
    (what ``torch.cumsum(mask, 0) - 1`` returns).  GPU masks go through the library's scan (rdg_mask_rank), not the framework's."""
    if not mask.is_cuda:
        return torch.cumsum(mask, 0) - 1
    m = mask.detach().contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else (m != 0).view(torch.uint8)
    n = m.numel()
    rank = torch.empty(n, dtype=torch.int64, device=m.device)
    if n == 0:
        return rank
    L = _lib.lib()
    with torch.cuda.device(m.device):
        ws = torch.empty(L.rdg_mask_rank_ws_bytes(n), dtype=torch.uint8, device=m.device)
        _lib.check(L.rdg_mask_rank(n, _lib.ptr(m), _lib.ptr(rank), _lib.ptr(ws), _lib.stream_ptr()), "rdg_mask_rank")
    return rank


def _compact(mask: torch.Tensor, n: int) -> torch.Tensor:
    """Indices of the True entries of ``mask`` in ascending order, their number ``n`` already known on the host: no read-back
    (a boolean-mask index or ``nonzero`` waits for the count it has to size its result with)."""
    P = mask.numel()
    pos = mask_rank(mask)
    out = torch.empty(n + 1, dtype=torch.int64, device=mask.device)
    out.scatter_(0, torch.where(mask, pos, torch.full_like(pos, n)), torch.arange(P, device=mask.device))
    return out[:n]


def densify_and_prune(fp: FlatParams, stats: DensifyStats, per_point: Dict[str, torch.Tensor], max_grad,
                      min_opacity: float, extent: float, max_screen_size, percent_dense: float = 0.01, N: int = 2,
                      z: Optional[torch.Tensor] = None, decisions: Optional[Dict[str, torch.Tensor]] = None,
                      spatial_order: bool = False, want_decisions: bool = True, timings: Optional[dict] = None,
                      pool: Optional[FlatPool] = None) -> DensifyResult:
    """``fp`` holds at least xyz [P,3], scaling [P,3] (log), rotation [P,4] (raw), opacity [P,1] (logit); every other
    segment (SH features, motion coefficients, ...) is carried along row-wise.  ``per_point``: further [P,...]
    tensors that follow the Gaussians (gaussian_to_time, gaussian_to_time_ind).  ``z``: optional standard-normal
    draws [N * n_split, 3] for the split children (default: torch.randn on the device).  ``max_grad``: a float or a 0-dim
    device tensor (a threshold computed on the device needs no read-back).  ``decisions``: replay the masks of an earlier call
    ({"clone": [P], "split": [P], "prune": [rows after clone + split]}, as returned in ``DensifyResult.decisions``) instead of
    thresholding this run's statistics -- for experiments that must hold the set of Gaussians fixed across runs
    (scripts/psnr_delta.py: a borderline Gaussian crossing the gradient threshold in one run and not in the other changes P
    and, from there, the whole trajectory).  ``spatial_order``: the surviving rows along the Z curve of their new positions
    (rodygs_amd/layout.py) -- clones and split children are appended at the end, so without it the memory coherence the kernels
    profit from decays with every densification.

    What the loop pays for (bench.py --loop): the three masks, the final prune (a per-Gaussian property: both copies of a
    split Gaussian share opacity and scale) and the five counts the new buffers are sized with are formed on the device and
    read back ONCE; the source-row list is built from them without further waits (``_compact``); the Z-curve order of the new
    cloud is composed INTO that list (the children's positions are computed first), so every buffer is gathered once, not
    twice.  ``want_decisions=False`` skips building the masks of ``DensifyResult.decisions``.  ``pool``: a ``FlatPool`` the new
    buffers come from (no allocation); the caller gives ``fp``'s storage back to it once nothing refers to ``fp`` any more."""
    if not fp.flat.is_cuda:
        raise RuntimeError("rodygs_amd.densify_and_prune: buffers must be on the GPU (no CPU fallback exists)")
    if fp.shapes["scaling"][1:] != (3,):
        raise NotImplementedError("isotropic scaling ([P,1]) is not supported")
    if decisions is not None:
        return _densify_general(fp, stats, per_point, max_grad, min_opacity, extent, max_screen_size, percent_dense, N, z,
                                decisions, spatial_order)
    dev = fp.flat.device
    P = fp.shapes["xyz"][0]
    ph = _Phase(timings, dev)
    with torch.no_grad(), torch.cuda.device(dev):
        grads = stats.xyz_gradient_accum / stats.denom
        grads[grads.isnan()] = 0.0
        max_s = torch.exp(fp["scaling"].detach()).max(dim=1).values
        small = max_s <= percent_dense * extent
        clone_mask = (torch.norm(grads, dim=-1) >= max_grad) & small                    # rodygs_static.py:244-251
        split_mask = (grads.squeeze(-1) >= max_grad) & ~small                            # :185-193 (clones: grad 0)
        # final prune (rodygs_static.py:286-298) on the rows as they stand after clone + split: originals and clones carry the
        # Gaussian's own opacity and scale, both children of a split its opacity and max scale / (0.8 N)
        low = torch.sigmoid(fp["opacity"].detach().reshape(-1)) < min_opacity
        prune_own, prune_child = low, low
        if max_screen_size:                                                              # max_radii2D was just zeroed
            prune_own = low | (max_s > 0.1 * extent)
            # the children's world size as the reference tests it: get_scaling of the STORED rows, exp(log(s * 1 / (0.8 N))) with
            # the float32 reciprocal rdg_split_children multiplies by (rodygs_static.py:201-203, 292-296) -- max_s / (0.8 N) can
            # sit one ulp on the other side of 0.1 * extent, and the general path (decisions=) tests the stored rows
            inv_shrink = torch.tensor(1.0, dtype=torch.float32, device=dev) / (torch.tensor(0.8, dtype=torch.float32, device=dev) * N)
            prune_child = low | (torch.exp(torch.log(max_s * inv_shrink)) > 0.1 * extent)
        keep0 = ~split_mask & ~prune_own
        keepc = clone_mask & ~prune_own
        keeps = split_mask & ~prune_child
        counts = torch.stack([split_mask.sum(), clone_mask.sum(), keep0.sum(), keepc.sum(), keeps.sum()])
        ph.mark("decisions")
        n_sel, n_clone, k0, kc, ks = (int(v) for v in counts.tolist())                  # the ONE read-back
        ph.mark("readback")
        n_child = n_sel * N
        src0, srcc, srcs = _compact(keep0, k0), _compact(keepc, kc), _compact(keeps, ks)
        src = torch.cat([src0, srcc, srcs.repeat(N)])                                    # repeat(N, 1) order of the children
        n_new, n_child_kept = k0 + kc + ks * N, ks * N
        first = n_new - n_child_kept
        moments = torch.zeros(n_new, dtype=torch.bool, device=dev)
        moments[:k0] = True                                                              # survivors keep their Adam moments
        # split children: positions and scales first (small), so that the Z-curve order can be taken on the NEW cloud
        child_xyz = child_sc = None
        if n_child_kept:
            if z is None:
                z = torch.randn(n_child, 3, device=dev)
            rank = mask_rank(split_mask)                                                 # position in the split selection
            child_no = torch.cat([rank[srcs] + r * n_sel for r in range(N)])
            zz = z.to(device=dev, dtype=torch.float32)[child_no].contiguous()
            parents = src[first:].contiguous()
            child_xyz = torch.empty(n_child_kept, 3, dtype=torch.float32, device=dev)
            child_sc = torch.empty(n_child_kept, 3, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().rdg_split_children(n_child_kept, N, _lib.ptr(parents), _lib.ptr(fp["xyz"].detach()),
                                                     _lib.ptr(fp["scaling"].detach()), _lib.ptr(fp["rotation"].detach()),
                                                     _lib.ptr(zz), _lib.ptr(child_xyz), _lib.ptr(child_sc),
                                                     _lib.stream_ptr()), "rdg_split_children")
        ph.mark("row_list")
        child_dst = None
        if spatial_order:
            from .layout import morton_order
            xyz_new = fp["xyz"].detach()[src]
            if n_child_kept:
                xyz_new[first:] = child_xyz
            perm = morton_order(xyz_new)
            src, moments = src[perm].contiguous(), moments[perm].contiguous()
            if n_child_kept:
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(n_new, device=dev)
                child_dst = inv[first:]
        ph.mark("re_sort")
        out = rebuild_flat_params(fp, src, moments, pool)
        if n_child_kept:
            if child_dst is None:
                out["xyz"].detach()[first:] = child_xyz
                out["scaling"].detach()[first:] = child_sc
            else:
                out["xyz"].detach().index_copy_(0, child_dst, child_xyz)
                out["scaling"].detach().index_copy_(0, child_dst, child_sc)
        new_pp = {k: v[src] for k, v in per_point.items()}
        ph.mark("gather")
        used = None
        if want_decisions:
            # the composite-row prune mask of the general form (rows after clone + split: originals not split, clones, children)
            a0, ac, as_ = _compact(~split_mask, P - n_sel), _compact(clone_mask, n_clone), _compact(split_mask, n_sel)
            used = {"clone": clone_mask.clone(), "split": split_mask.clone(),
                    "prune": torch.cat([prune_own[a0], prune_own[ac], prune_child[as_].repeat(N)])}
        from .deform import invalidate_birth_order_cache
        invalidate_birth_order_cache()       # the per-point tensors (birth indices) are new objects from here on
        res = DensifyResult(out, DensifyStats.zeros(n_new, dev), new_pp, n_clone, n_sel, P + n_clone + n_child - n_new, used)
        ph.mark("finish")
        return res


def reset_opacity_(fp: FlatParams, name: str = "opacity", max_opacity: float = 0.01) -> None:
    """ThreeDGSTrainer.reset_opacity (/root/reference/src/trainer/rodygs_static.py:151-160) on the flat bucket, in
    place: opacity logits become inverse_sigmoid(min(sigmoid(logit), 0.01)) and -- as replace_tensor_to_optimizer
    does (/root/reference/src/trainer/utils.py:15-32) -- both Adam moments of that segment are zeroed while the step
    counter is kept.  One HIP launch (rdg_reset_opacity); the Parameter object, its .grad view and every other
    segment are untouched, so nothing that points into the bucket needs re-binding."""
    if not fp.flat.is_cuda:
        raise RuntimeError("reset_opacity_: the flat bucket must live on the GPU (no CPU fallback exists)")
    o, n = fp.offsets[name]
    with torch.cuda.device(fp.flat.device):
        _lib.check(_lib.lib().rdg_reset_opacity(n, float(max_opacity), fp.flat.data_ptr() + 4 * o,
                                                fp.exp_avg.data_ptr() + 4 * o, fp.exp_avg_sq.data_ptr() + 4 * o,
                                                _lib.stream_ptr()), "rdg_reset_opacity")


# ---- fixed capacity: densify and prune IN PLACE (round 6) ---------------------------------------------------------------------------
# A captured hipGraph holds the address and the launch dimensions of every buffer of the step; densify_and_prune above changes
# both (P changes, the rows move to the other buffer set), so a graph had to be re-captured at every densification -- which costs
# more than the graph saves in the 100 steps between two of them.  The alternative built here: the cloud lives in buffers of a
# FIXED number of rows; rows that hold no Gaussian are "dead" -- parked far behind every camera (culled by the near plane: no
# tile, radius 0, no gradient, no statistics), motion coefficients and Adam moments zero (a zero gradient on zero moments is a
# zero update: they stay where they are).  Densification then is row surgery in place: pruned Gaussians and split parents die,
# clones and split children are written into dead rows.  Nothing a captured graph refers to moves or changes size.
DEAD_XYZ = (0.0, 0.0, -1.0e6)
DEAD_OPACITY_LOGIT = -20.0


def dead_row_template(fp: FlatParams, rows: torch.Tensor) -> None:
    """Park ``rows`` (int64 indices): position behind every camera, opacity logit -20, unit quaternion, log-scale of 1e-3, zero
    features and motion coefficients, zero Adam moments."""
    with torch.no_grad():
        for k in fp.names:
            p = fp[k].detach()
            if k == "xyz":
                p[rows] = torch.tensor(DEAD_XYZ, dtype=torch.float32, device=p.device)
            elif k == "opacity":
                p[rows] = DEAD_OPACITY_LOGIT
            elif k == "scaling":
                p[rows] = -6.9
            elif k == "rotation":
                p[rows] = torch.tensor((1.0, 0.0, 0.0, 0.0), dtype=torch.float32, device=p.device)
            else:
                p[rows] = 0.0
            o, n = fp.offsets[k]
            for buf in (fp.exp_avg, fp.exp_avg_sq):
                buf[o:o + n].view(fp.shapes[k])[rows] = 0.0


def densify_and_prune_inplace(fp: FlatParams, stats: DensifyStats, per_point: Dict[str, torch.Tensor], dead: torch.Tensor,
                              max_grad, min_opacity: float, extent: float, max_screen_size, percent_dense: float = 0.01,
                              N: int = 2, z: Optional[torch.Tensor] = None):
    """``densify_and_prune`` (same decisions: rodygs_static.py:170-319, the fast path's masks above) on a cloud of FIXED capacity,
    in place.  ``dead`` [rows] bool marks the rows that hold no Gaussian and is updated in place; ``per_point`` tensors are
    updated in place; ``stats`` is zeroed.  The set of live Gaussians after the call -- values, moments, per-point data -- is the
    one ``densify_and_prune`` produces (a -m gpu test compares the two as multisets); their ROW ORDER is not: new Gaussians land
    wherever a row is free.  Returns a dict (cloned, split, pruned, live) or None when the free rows do not suffice (nothing has
    been touched then: the caller grows the capacity)."""
    if not fp.flat.is_cuda:
        raise RuntimeError("rodygs_amd.densify_and_prune_inplace: buffers must be on the GPU (no CPU fallback exists)")
    dev = fp.flat.device
    with torch.no_grad(), torch.cuda.device(dev):
        live = ~dead
        grads = stats.xyz_gradient_accum / stats.denom
        grads[grads.isnan()] = 0.0
        max_s = torch.exp(fp["scaling"].detach()).max(dim=1).values
        small = max_s <= percent_dense * extent
        clone_mask = (torch.norm(grads, dim=-1) >= max_grad) & small & live
        split_mask = (grads.squeeze(-1) >= max_grad) & ~small & live
        low = torch.sigmoid(fp["opacity"].detach().reshape(-1)) < min_opacity
        prune_own, prune_child = low, low
        if max_screen_size:
            inv_shrink = torch.tensor(1.0, dtype=torch.float32, device=dev) / (torch.tensor(0.8, dtype=torch.float32, device=dev) * N)
            prune_own = low | (max_s > 0.1 * extent)
            prune_child = low | (torch.exp(torch.log(max_s * inv_shrink)) > 0.1 * extent)
        kill = live & (split_mask | prune_own)             # split parents are replaced by their children; pruned originals go
        keepc = clone_mask & ~prune_own                     # (a pruned original's clone is pruned with it: same opacity, same scale)
        keeps = split_mask & ~prune_child
        free = dead | kill
        counts = torch.stack([keepc.sum(), keeps.sum(), free.sum(), kill.sum(), split_mask.sum(), clone_mask.sum()])
        kc, ks, n_free, n_kill, n_sel, n_clone = (int(v) for v in counts.tolist())            # the ONE read-back
        n_new = kc + ks * N
        if n_new > n_free:
            return None
        srcc, srcs = _compact(keepc, kc), _compact(keeps, ks)
        src = torch.cat([srcc, srcs.repeat(N)])                                                 # clones, then children in repeat order
        dst = _compact(free, n_free)[:n_new]
        child_xyz = child_sc = None
        if ks:
            if z is None:
                z = torch.randn(n_sel * N, 3, device=dev)
            rank = mask_rank(split_mask)
            child_no = torch.cat([rank[srcs] + r * n_sel for r in range(N)])
            zz = z.to(device=dev, dtype=torch.float32)[child_no].contiguous()
            parents = src[kc:].contiguous()
            child_xyz = torch.empty(ks * N, 3, dtype=torch.float32, device=dev)
            child_sc = torch.empty(ks * N, 3, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().rdg_split_children(ks * N, N, _lib.ptr(parents), _lib.ptr(fp["xyz"].detach()),
                                                     _lib.ptr(fp["scaling"].detach()), _lib.ptr(fp["rotation"].detach()),
                                                     _lib.ptr(zz), _lib.ptr(child_xyz), _lib.ptr(child_sc),
                                                     _lib.stream_ptr()), "rdg_split_children")
        # read every source row BEFORE any row is rewritten (a split parent's row may be the very slot its child lands in)
        taken = {k: fp[k].detach().index_select(0, src) for k in fp.names}
        taken_pp = {k: v.index_select(0, src) for k, v in per_point.items()}
        if ks:
            taken["xyz"][kc:] = child_xyz
            taken["scaling"][kc:] = child_sc
        # rows that die: parked behind every camera, zero moments in every segment (the rows new Gaussians land in are dead rows
        # or rows that die right here: their moments are zero either way, which is what a new Gaussian starts with)
        kill_rows = _compact(kill, n_kill)
        fp["xyz"].detach().index_copy_(0, kill_rows, torch.tensor(DEAD_XYZ, dtype=torch.float32, device=dev).expand(n_kill, 3).contiguous())
        fp["opacity"].detach().index_fill_(0, kill_rows, DEAD_OPACITY_LOGIT)
        for k in fp.names:
            o, n = fp.offsets[k]
            rows_ = fp.shapes[k][0]
            if k not in ("xyz", "opacity", "scaling", "rotation"):
                fp[k].detach().view(rows_, -1).index_fill_(0, kill_rows, 0.0)      # features, motion coefficients
            for buf in (fp.exp_avg, fp.exp_avg_sq):
                buf[o:o + n].view(rows_, -1).index_fill_(0, kill_rows, 0.0)
            fp[k].detach().index_copy_(0, dst, taken[k])
        for k, v in per_point.items():
            v.index_copy_(0, dst, taken_pp[k])
        dead |= kill
        dead[dst] = False
        stats.xyz_gradient_accum.zero_(); stats.denom.zero_(); stats.max_radii2D.zero_()
        n_live = int(dead.numel()) - (n_free - n_new)
    # "pruned" as densify_and_prune counts it: the reference's prune mask names the split parents too (rodygs_static.py:205-209)
    return {"cloned": n_clone, "split": n_sel, "pruned": n_kill + (n_clone - kc) + (n_sel - ks) * N, "live": n_live,
            "free_rows": n_free - n_new}
