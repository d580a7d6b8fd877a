"""The reference's ITERATION at the speed of the fused step: ``FusedReferenceIteration``.

``trainstep.ReferenceIteration`` is the semantic mirror of ``RoDyGSTrainer.train`` (/root/reference/src/trainer/rodygs.py:157-179):
a static sub-step, then a dynamic sub-step, each ``train_iteration`` (:198-369) over the CONCATENATED cloud, gradients
accumulating in both clouds' parameters, only the sub-step's own trainer stepping and clearing (:364-369) -- written the way
the reference writes it (separate f_dc / f_rest tensors, autograd accumulating into ``.grad``, ``zero_grad`` after the step)
and tested against that flow in framework ops.  It pays for that shape: the SH features are copied into the rasterizer's
[P,16,3] input and their gradient split back (192 B per Gaussian each way), every gradient makes an extra trip through
autograd's accumulation, the stepped bucket is cleared, two Adam launches.

This class keeps the SEMANTICS and changes the bookkeeping:

* **two gradient buffers, written in turn, never accumulated, never cleared.**  The reference's rule -- a sub-step deposits its
  gradient in BOTH clouds; a cloud steps on (what the other sub-step left) + (its own sub-step's) and then clears -- means that
  at any step a cloud's gradient is the sum of exactly two backward passes: the previous sub-step's and this one's.  So every
  backward of a STATIC sub-step overwrites buffer A (all rows, both clouds, the MLP), every backward of a DYNAMIC sub-step
  overwrites buffer B, and the Adam launch of a sub-step reads A + B for the rows it owns (``RdgAdamSeg.grad2``).  The first
  iteration finds B zero.  One float add per element, the same add AccumulateGrad performs: the trajectories are the same.
* **the SH features of both clouds are ONE [Ps + Pd, 16, 3] tensor** (static rows first, as the reference concatenates): it IS
  the rasterizer's input, and the rasterizer's backward writes dL/dshs straight into the sub-step's gradient buffer; the two
  trainers own row ranges of it (row-structured Adam: DC coefficient at feature_lr, the rest at feature_lr / 20).
* the other getters of both clouds (and the dynamic cloud's deformation) are one autograd node that fills the two row segments
  of the rasterizer's inputs and, backward, writes every parameter gradient into the sub-step's buffer (``_MixedCloud``).
* ONE Adam launch per sub-step: the owner's rows of every group + its small bucket (camera poses for the static trainer, the
  deformation MLP for the dynamic one).

``from_reference(ri)`` builds one from a ``ReferenceIteration`` (same values, moments, network, poses, ground truth), which is
how the -m gpu test holds the two to the same parameters after several iterations.  ``GraphedIteration`` replays the whole
iteration (both sub-steps) as ONE captured hipGraph: at the reference's real cloud sizes (~0.1 M + 0.1 M) the eager iteration is
host-bound."""
from __future__ import annotations

import copy
import math
from typing import Optional

import torch

from . import _lib
from .deform import _birth_order
from .dp import FlatParams
from .losses import fused_photometric_loss
from .model_ops import pose_view_matrix
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, RasterState
from .trainstep import _MLP_SINK_ORDER, bind_module_to_flat

_GROUPS = ("xyz", "scaling", "rotation", "opacity")


class _MixedCloud(torch.autograd.Function):
    """Static ‖ dynamic getters into the two row segments of ONE set of rasterizer inputs (means3D, scales, rotations,
    opacities): rows [0, Ps) = the static cloud's getters (rdg_activate_*), rows [Ps, Ps + Pd) = the dynamic cloud's getters +
    deformation (rdg_dyn_getter_*).  Backward OVERWRITES ``sinks`` (s.xyz ... d.opacity, d.motion_coeff) with the parameter
    gradients and returns only dL/dbases to autograd."""

    @staticmethod
    def forward(ctx, bases, s_par, d_par, coeff, time_ind, scale, sinks):
        L = _lib.lib()
        s_xyz, s_sc, s_ro, s_op = s_par
        d_xyz, d_sc, d_ro, d_op = d_par
        Ps, Pd, dev = s_xyz.shape[0], d_xyz.shape[0], s_xyz.device
        bs = bases.detach().contiguous()
        Tu = bs.shape[0] - 1
        P = Ps + Pd
        f32 = dict(dtype=torch.float32, device=dev)
        m3, sc, ro, op = torch.empty(P, 3, **f32), torch.empty(P, 3, **f32), torch.empty(P, 4, **f32), torch.empty(P, 1, **f32)
        st = _lib.stream_ptr()
        with torch.cuda.device(dev):
            if Ps:
                _lib.check(L.rdg_activate_forward(Ps, 1, _lib.ptr(s_xyz), None, _lib.ptr(s_sc), _lib.ptr(s_ro), None,
                                                  _lib.ptr(s_op), None, None, _lib.ptr(m3), _lib.ptr(sc), _lib.ptr(ro),
                                                  _lib.ptr(op), None, st), "rdg_activate_forward")
            if Pd:
                _lib.check(L.rdg_dyn_getter_forward(Pd, Tu, _lib.ptr(coeff), _lib.ptr(time_ind), _lib.ptr(bs), float(scale),
                                                    _lib.ptr(d_xyz), _lib.ptr(d_sc), _lib.ptr(d_ro), _lib.ptr(d_op),
                                                    _lib.ptr(m3[Ps:]), _lib.ptr(sc[Ps:]), _lib.ptr(ro[Ps:]), _lib.ptr(op[Ps:]),
                                                    st), "rdg_dyn_getter_forward")
        ctx.pars = (s_sc, s_ro, s_op, d_sc, d_ro, d_op, coeff, time_ind, bs)     # parameters: alive for the whole step anyway
        ctx.dims, ctx.scale, ctx.sinks = (Ps, Pd, Tu), float(scale), sinks
        ctx.set_materialize_grads(False)
        return m3, sc, ro, op

    @staticmethod
    def backward(ctx, g_m, g_s, g_r, g_o):
        L = _lib.lib()
        s_sc, s_ro, s_op, d_sc, d_ro, d_op, coeff, ti, bs = ctx.pars
        Ps, Pd, Tu = ctx.dims
        dev = bs.device
        sk = ctx.sinks
        g = [None if t is None else t.contiguous() for t in (g_m, g_s, g_r, g_o)]
        d_bases = torch.empty_like(bs)
        st = _lib.stream_ptr()
        with torch.cuda.device(dev):
            if Ps:
                _lib.check(L.rdg_activate_backward(Ps, 1, _lib.ptr(s_sc), _lib.ptr(s_ro), _lib.ptr(s_op), *[_lib.ptr(t) for t in g],
                                                   None, _lib.ptr(sk["s.xyz"]), _lib.ptr(sk["s.scaling"]),
                                                   _lib.ptr(sk["s.rotation"]), _lib.ptr(sk["s.opacity"]), None, None, st),
                           "rdg_activate_backward")
            if Pd:
                order, inv, seg = _birth_order(ti, Tu)
                sws = torch.empty(L.rdg_deform_sorted_ws_bytes(Pd), dtype=torch.uint8, device=dev)
                gd = [None if t is None else t[Ps:] for t in g]
                _lib.check(L.rdg_dyn_getter_backward(Pd, Tu, _lib.ptr(coeff), _lib.ptr(ti), _lib.ptr(bs), ctx.scale,
                                                     _lib.ptr(d_sc), _lib.ptr(d_ro), _lib.ptr(d_op), *[_lib.ptr(t) for t in gd],
                                                     _lib.ptr(sk["d.xyz"]), _lib.ptr(sk["d.scaling"]), _lib.ptr(sk["d.rotation"]),
                                                     _lib.ptr(sk["d.opacity"]), _lib.ptr(sk["d.motion_coeff"]), _lib.ptr(d_bases),
                                                     _lib.ptr(order), _lib.ptr(inv), _lib.ptr(seg), _lib.ptr(sws), st),
                           "rdg_dyn_getter_backward")
            else:
                d_bases.zero_()
        return d_bases, None, None, None, None, None, None


class FusedReferenceIteration:
    """See the module docstring.  Same constructor as ``trainstep.ReferenceIteration`` (it builds one and converts it)."""

    def __init__(self, static_scene: dict = None, dynamic_scene: dict = None, num_frames: int = 100, sh_degree: int = 3,
                 device="cuda", seed: int = 777, spatial_lr_scale: float = 5.0, orbit_deg: float = 15.0,
                 spatial_order: bool = True, _from=None):
        from .densify import DensifyStats
        from .trainstep import ReferenceIteration
        ri = _from if _from is not None else ReferenceIteration(static_scene, dynamic_scene, num_frames, sh_degree, device, seed,
                                                                spatial_lr_scale, orbit_deg, spatial_order)
        dev = ri.device
        self.device, self.T, self.sh_degree, self.spatial_lr_scale = dev, ri.T, ri.sh_degree, ri.spatial_lr_scale
        self.W, self.H, self.tanfovx, self.tanfovy, self.proj_t, self.bg = ri.W, ri.H, ri.tanfovx, ri.tanfovy, ri.proj_t, ri.bg
        self.Ps, self.Pd = ri.Ps, ri.Pd
        Ps, Pd, P = ri.Ps, ri.Pd, ri.Ps + ri.Pd
        K = 1 + ri.fp_s.shapes["f_rest"][1]
        lr = ri.fp_s.lr
        spec = {}
        for pre, n, f in (("s.", Ps, ri.fp_s), ("d.", Pd, ri.fp_d)):
            for k in _GROUPS:
                spec[pre + k] = ((n,) + tuple(f.shapes[k][1:]), f.lr[k])
        spec["d.motion_coeff"] = ((Pd, 1, 16), ri.fp_d.lr["motion_coeff"])
        spec["features"] = ((P, K, 3), lr["f_dc"])            # static rows first; the rasterizer's shs input itself
        self.fp = fp = FlatParams(spec, dev)
        self.row_lr = (K * 3, 3, lr["f_rest"])
        self.grad = (fp.flat_grad, torch.zeros_like(fp.flat_grad))      # A (static sub-steps), B (dynamic sub-steps)
        self.time_ind = ri.time_ind
        self.emb_rows = ri.emb_rows
        self.net = copy.deepcopy(ri.net)
        for p_ in self.net.parameters():
            p_.grad = None
        self.sp_mlp = bind_module_to_flat(self.net, ri.sp_mlp.lr[ri.sp_mlp.names[0]], dev)
        self.grad_mlp = (self.sp_mlp.flat_grad, torch.zeros_like(self.sp_mlp.flat_grad))
        self.sp_cam = FlatParams({k: (ri.sp_cam.shapes[k], ri.sp_cam.lr[k]) for k in ri.sp_cam.names}, dev)
        self.load_state_from(ri)
        self.m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        self.stats = {"static": DensifyStats.zeros(Ps, dev), "dynamic": DensifyStats.zeros(Pd, dev)}
        self.raster_state = RasterState()
        self.gt = {k: v.clone() for k, v in ri.gt.items()}
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.keep_bases_grad = False
        self._staged = None                 # GraphedIteration: {"static" | "dynamic": (embedding rows, ground truth, step scalars)}
        self._segs = {}
        # per-sub-step sink tables (views of buffer A / B), built once
        self._sinks, self._mlp_sinks, self._feat_sink = [], [], []
        for b in (0, 1):
            g = self.grad[b]
            self._sinks.append({k: fp.segment(g, k).view(fp.shapes[k]) for k in fp.names if k != "features"})
            self._feat_sink.append(fp.segment(g, "features").view(P, K, 3))
            gm = self.grad_mlp[b]
            self._mlp_sinks.append([self.sp_mlp.segment(gm, n).view(self.sp_mlp.shapes[n]) for n in _MLP_SINK_ORDER])
        self._cam_sinks = {"q": self.sp_cam["cam_q"].grad, "t": self.sp_cam["cam_t"].grad}

    @classmethod
    def from_reference(cls, ri) -> "FusedReferenceIteration":
        return cls(_from=ri)

    def load_state_from(self, ri) -> None:
        """Parameters, both Adam moments and the step counters of a ``trainstep.ReferenceIteration`` (both clouds, the
        deformation network, the camera poses) into this object's layout.  The gradient buffers are not touched."""
        fp, Ps, P = self.fp, self.Ps, self.Ps + self.Pd
        K = fp.shapes["features"][1]
        with torch.no_grad():
            for pre, f, rows in (("s.", ri.fp_s, slice(0, Ps)), ("d.", ri.fp_d, slice(Ps, P))):
                names = _GROUPS + (("motion_coeff",) if pre == "d." else ())
                for k in names:
                    for dst, src in ((fp.flat, f.flat), (fp.exp_avg, f.exp_avg), (fp.exp_avg_sq, f.exp_avg_sq)):
                        fp.segment(dst, pre + k).copy_(f.segment(src, k))
                for dst, src in ((fp.flat, f.flat), (fp.exp_avg, f.exp_avg), (fp.exp_avg_sq, f.exp_avg_sq)):
                    feat = fp.segment(dst, "features").view(P, K, 3)
                    feat[rows, :1] = f.segment(src, "f_dc").view(-1, 1, 3)
                    feat[rows, 1:] = f.segment(src, "f_rest").view(-1, K - 1, 3)
            for mine, theirs in ((self.sp_mlp, ri.sp_mlp), (self.sp_cam, ri.sp_cam)):
                for n in mine.names:
                    for a, b in ((mine.flat, theirs.flat), (mine.exp_avg, theirs.exp_avg), (mine.exp_avg_sq, theirs.exp_avg_sq)):
                        mine.segment(a, n).copy_(theirs.segment(b, n))
        self.steps = {"static": ri.fp_s.step_count, "dynamic": ri.fp_d.step_count}

    # ---- the reference's pieces ----------------------------------------------------------------------------------------------
    def settings(self, pose_grads: bool) -> GaussianRasterizationSettings:
        return GaussianRasterizationSettings(self.H, self.W, self.tanfovx, self.tanfovy, self.bg, 1.0, self.proj_t,
                                             self.sh_degree, False, False, pose_grads, pose_grads)

    def make_ground_truth(self, target_scene: dict, frames) -> None:
        dev = self.device
        with torch.no_grad():
            for f in frames:
                vm = pose_view_matrix(self.sp_cam["cam_q"], self.sp_cam["cam_t"], int(f))
                out = GaussianRasterizer(self.settings(False), state=self.raster_state)(
                    means3D=target_scene["means3D"].to(dev), means2D=torch.zeros_like(target_scene["means3D"]).to(dev),
                    shs=target_scene["shs"].to(dev), opacities=target_scene["opacities"].to(dev),
                    scales=target_scene["scales"].to(dev), rotations=target_scene["rotations"].to(dev), viewmatrix=vm)
                self.gt[int(f)] = out[0].clamp(0, 1).clone()

    def params(self, cloud: str) -> dict:
        """The cloud's raw parameters in the reference's names (views): xyz, f_dc, f_rest, scaling, rotation, opacity
        (+ motion_coeff)."""
        pre, rows = ("s.", slice(0, self.Ps)) if cloud == "static" else ("d.", slice(self.Ps, self.Ps + self.Pd))
        out = {k: self.fp[pre + k] for k in _GROUPS}
        feat = self.fp["features"]
        out["f_dc"], out["f_rest"] = feat[rows, :1], feat[rows, 1:]
        if cloud == "dynamic":
            out["motion_coeff"] = self.fp["d.motion_coeff"]
        return out

    def effective_grad(self, cloud: str) -> dict:
        """What the cloud's trainer would step on NOW: buffer A + buffer B over its rows (the reference's ``.grad``)."""
        pre, rows = ("s.", slice(0, self.Ps)) if cloud == "static" else ("d.", slice(self.Ps, self.Ps + self.Pd))
        names = _GROUPS + (("motion_coeff",) if cloud == "dynamic" else ())
        out = {k: self._sinks[0][pre + k] + self._sinks[1][pre + k] for k in names}
        feat = self._feat_sink[0][rows] + self._feat_sink[1][rows]
        out["f_dc"], out["f_rest"] = feat[:, :1], feat[:, 1:]
        return out

    def forward_backward(self, frame: int, which: str) -> torch.Tensor:
        """``train_iteration`` (rodygs.py:198-341) up to the optimiser: render the concatenated cloud, loss, backward -- every
        gradient of BOTH clouds, the MLP and (static sub-step) the poses OVERWRITTEN in this sub-step's buffer -- and the
        densification statistics of the sub-step's own slice."""
        static = which == "static"
        b = 0 if static else 1
        fp = self.fp
        sg = self._staged[which] if self._staged is not None else None
        self.net.grad_sinks = self._mlp_sinks[b]
        allb = self.net.motion_basis(self.emb_rows[frame] if sg is None else sg[0])           # [T + 1, 16, 7]
        if self.keep_bases_grad:            # (tests hold the MLP at its OUTPUT: its parameter gradients are cancellation residue)
            allb.retain_grad()
            self._last_allb = allb
        m3, sc, ro, op = _MixedCloud.apply(
            allb, tuple(fp["s." + k].detach() for k in _GROUPS), tuple(fp["d." + k].detach() for k in _GROUPS),
            fp["d.motion_coeff"].detach().view(self.Pd, 16), self.time_ind, self.spatial_lr_scale, self._sinks[b])
        scal = None if sg is None else sg[2]
        if static:
            vm = pose_view_matrix(self.sp_cam["cam_q"], self.sp_cam["cam_t"], frame, grad_sinks=self._cam_sinks, step_scalars=scal)
        else:
            with torch.no_grad():          # the dynamic sub-step renders with the refined poses, no pose gradient (rodygs.py:170-178)
                vm = pose_view_matrix(self.sp_cam["cam_q"], self.sp_cam["cam_t"], frame, step_scalars=scal)
        self.m2.grad = None
        row0, stats = (0, self.stats["static"]) if static else (self.Ps, self.stats["dynamic"])
        out = GaussianRasterizer(self.settings(static), state=self.raster_state)(
            means3D=m3, means2D=self.m2, shs=fp["features"], opacities=op, scales=sc, rotations=ro, viewmatrix=vm,
            grad_sinks={"shs": self._feat_sink[b], "densify": stats.sink(row0)})
        loss = fused_photometric_loss(out[0], self.gt[frame] if sg is None else sg[1], 0.2)
        loss.backward(self._one)
        return loss.detach()

    def _adam_segments(self, which: str):
        hit = self._segs.get(which)
        if hit is not None:
            return hit
        fp = self.fp
        static = which == "static"
        pre, row0, rows = ("s.", 0, self.Ps) if static else ("d.", self.Ps, self.Pd)
        ent = []       # (bucket, offset, n, lr_head, lr_tail, row_len, head_len, grad A, grad B or None)
        for k in _GROUPS + (() if static else ("motion_coeff",)):
            o, n = fp.offsets[pre + k]
            ent.append((fp, o, n, fp.lr[pre + k], fp.lr[pre + k], 1, 1, self.grad[0], self.grad[1]))
        o, _ = fp.offsets["features"]
        rl = self.row_lr[0]
        ent.append((fp, o + row0 * rl, rows * rl, fp.lr["features"], self.row_lr[2], rl, self.row_lr[1], self.grad[0], self.grad[1]))
        if static:          # the camera optimiser steps with the static trainer (rodygs.py:364-369); its gradient is this sub-step's
            for k in self.sp_cam.names:
                o, n = self.sp_cam.offsets[k]
                ent.append((self.sp_cam, o, n, self.sp_cam.lr[k], self.sp_cam.lr[k], 1, 1, self.sp_cam.flat_grad, None))
        else:               # the deformation network: one segment (one learning rate; padding carries zero gradients)
            sp = self.sp_mlp
            ent.append((sp, 0, sp.numel, sp.lr[sp.names[0]], sp.lr[sp.names[0]], 1, 1, self.grad_mlp[0], self.grad_mlp[1]))
        ent = [e for e in ent if e[2] > 0]
        segs = (_lib.RdgAdamSeg * len(ent))()
        for i, (f, o, n, lh, lt, rl_, hl, ga, gb) in enumerate(ent):
            segs[i].n = n
            segs[i].param = f.flat.data_ptr() + 4 * o
            segs[i].grad = ga.data_ptr() + 4 * o
            segs[i].grad2 = None if gb is None else gb.data_ptr() + 4 * o
            segs[i].exp_avg = f.exp_avg.data_ptr() + 4 * o
            segs[i].exp_avg_sq = f.exp_avg_sq.data_ptr() + 4 * o
            segs[i].lr_head, segs[i].lr_tail, segs[i].row_len, segs[i].head_len = lh, lt, rl_, hl
        self._segs[which] = (segs, len(ent))
        return self._segs[which]

    def step(self, which: str) -> None:
        """``current_gs.optimizer.step(); zero_grad()`` (+ the camera optimiser for the static trainer): ONE launch over the
        trainer's rows with gradient = buffer A + buffer B; nothing to clear (the next backward of either kind overwrites)."""
        L = _lib.lib()
        segs, n = self._adam_segments(which)
        self.steps[which] += 1
        sg = self._staged[which] if self._staged is not None else None
        with torch.cuda.device(self.device):
            if sg is not None:
                _lib.check(L.rdg_adam_step_multi_dev(n, segs, 0.9, 0.999, 1e-15, _lib.ptr(sg[2]), _lib.stream_ptr()),
                           "rdg_adam_step_multi_dev")
            else:
                _lib.check(L.rdg_adam_step_multi(n, segs, 0.9, 0.999, 1e-15, self.steps[which], _lib.stream_ptr()),
                           "rdg_adam_step_multi")

    def sub_step(self, frame: int, which: str) -> torch.Tensor:
        loss = self.forward_backward(frame, which)
        self.step(which)
        return loss

    @staticmethod
    def frames_of(it: int, perm):
        return perm[(2 * it) % len(perm)], perm[(2 * it + 1) % len(perm)]

    def iteration(self, it: int, perm) -> tuple:
        """Static sub-step, then dynamic sub-step (each draws its own frame, as the two data loaders of the reference do)."""
        fs, fd = self.frames_of(it, perm)
        return self.sub_step(fs, "static"), self.sub_step(fd, "dynamic")


class GraphedIteration:
    """The whole iteration (static sub-step + dynamic sub-step) as ONE captured hipGraph, replayed with one launch per
    iteration.  What changes from iteration to iteration lives in device memory at fixed addresses: per sub-step the frame's
    time-embedding rows, its ground truth and a 128-byte ``RdgStepScalars`` (Adam's bias corrections, the frame index for the
    pose kernels), refreshed by three small copies each before the replay.  Same kernels, same arithmetic as the eager
    iteration.  Learning rates are the ones of capture time (by value in the Adam segments).  The instance count of every
    replayed forward is folded into a sticky device maximum; ``check()`` raises if any frame outgrew the captured capacity."""

    def __init__(self, ri: FusedReferenceIteration, perm, first_iteration: int = 0, warmup: int = 2):
        self.ri, self.perm = ri, list(perm)
        dev = ri.device
        W = _lib.STEP_SCALARS_FLOATS
        f0 = self.perm[0]
        self.inputs = {w: (torch.empty_like(ri.emb_rows[f0]), torch.empty_like(ri.gt[f0]),
                           torch.zeros(W, dtype=torch.float32, device=dev)) for w in ("static", "dynamic")}
        # pinned staging ring (one row per sub-step and iteration): the H2D copies are asynchronous, so a row may only be
        # rewritten once the copy that read it is known to be done -- an event per half ring, as trainstep.GraphedStep does
        self.RING = 128
        self._ring = torch.zeros(self.RING, 2, W, dtype=torch.float32).pin_memory()
        self._ring_i32 = self._ring.view(torch.int32)
        self._slot, self._fence = 0, []
        st = ri.raster_state
        st.nren_max = torch.zeros(1, dtype=torch.int32, device=dev)
        st.nren_max_key = (ri.Ps + ri.Pd, ri.H, ri.W)
        it = first_iteration
        for _ in range(max(1, warmup)):              # eager, on the staged inputs: hints, caches, lazy initialisation
            self._stage(it)
            ri._staged = self.inputs
            try:
                ri.iteration(it, self.perm)
            finally:
                ri._staged = None
            it += 1
        torch.cuda.synchronize(dev)
        key = st.nren_max_key
        with st.lock:
            st.capacity_hint[key] = max(int(st.capacity_hint.get(key, 0)), int(st.d_high.get(key, 0)))
        steps = dict(ri.steps)
        self.graph = torch.cuda.CUDAGraph()
        self._stage(it)
        torch.cuda.synchronize(dev)
        st.graph_capture = True
        ri._staged = self.inputs
        try:
            with torch.cuda.graph(self.graph):
                self.losses = ri.iteration(it, self.perm)
        finally:
            st.graph_capture = False
            ri._staged = None
            st.keep_alive.clear()
        ri.steps = steps                              # capture advanced the host counters only
        self.next_iteration = it
        self._cap = st.last_nren[2]
        st.nren_max.zero_()

    def _stage(self, it: int) -> None:
        ri = self.ri
        i = self._slot
        self._slot = (i + 1) % self.RING
        if i % (self.RING // 2) == 0:
            if self._fence:
                self._fence.pop(0).synchronize()
            ev = torch.cuda.Event()
            ev.record()
            self._fence.append(ev)
        for j, (which, frame) in enumerate(zip(("static", "dynamic"), ri.frames_of(it, self.perm))):
            emb, gt, scal = self.inputs[which]
            k = ri.steps[which] + 1
            row = self._ring[i, j]
            row[0] = 1.0 / (1.0 - 0.9 ** k)            # float(1 / bc1), float(sqrt(bc2)) from doubles, as rdg_adam_step_multi
            row[1] = math.sqrt(1.0 - 0.999 ** k)
            self._ring_i32[i, j, 2] = int(frame)
            self._ring_i32[i, j, 3] = 0                # learning rates: by value, from the captured segments
            scal.copy_(row, non_blocking=True)
            emb.copy_(ri.emb_rows[frame])
            gt.copy_(ri.gt[frame])

    def step(self) -> tuple:
        self._stage(self.next_iteration)
        self.graph.replay()
        self.next_iteration += 1
        self.ri.steps["static"] += 1
        self.ri.steps["dynamic"] += 1
        return self.losses

    def check(self) -> int:
        from . import rasterizer
        st = self.ri.raster_state
        n = int(st.nren_max.item())
        st.nren_max.zero_()
        key = st.nren_max_key
        with st.lock:
            st.capacity_hint[key] = max(n, int(st.capacity_hint.get(key, 0) * rasterizer.HINT_DECAY))
            st.note_instances(key, n)
        if n > self._cap:
            raise rasterizer.RasterizerCapacityOverflow(
                f"a replayed frame needed {n} instances, the graph was captured with {self._cap}: re-build the GraphedIteration")
        return n

    def close(self) -> None:
        torch.cuda.current_stream(self.ri.device).synchronize()
        self.ri.raster_state.nren_max = None
        self.ri.raster_state.nren_max_key = None
