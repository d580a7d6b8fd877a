"""Fused per-Gaussian glue between raw parameters and the rasterizer (csrc/rdg_model.hip).

* ``activate_gaussians`` = the reference's model getters + the deformation add + the feature concat
  (/root/reference/src/model/rodygs_static.py:82-105, /root/reference/src/trainer/rodygs.py:68-113) in two HIP
  launches forward and two backward, instead of ~25 elementwise framework kernels and a cat.
* ``gs_properties`` = ``get_GS_properties`` for a static + a dynamic cloud: both activated straight into the two row
  segments of one set of rasterizer inputs (no ``torch.cat``).
* ``pose_view_matrix`` = ``FixedCameraTorch.world_view_transform(...).transpose(0, 1)``
  (/root/reference/src/data/utils.py:161-170; the transpose is the "glm storage" of renderer.py:97-99) for one
  frame of the learnable pose tables, one launch each way instead of ~130 scalar-tensor kernels.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from contextlib import nullcontext as _nullcontext

from . import _lib


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, dxyz, scaling, rotation, drot, opacity, f_dc, f_rest, sinks):
        L = _lib.lib()
        if not xyz.is_cuda:
            raise RuntimeError("rodygs_amd.activate_gaussians: tensors must be on the GPU (no CPU fallback exists)")
        dev = xyz.device
        P = xyz.shape[0]
        K = 1 if f_dc is None else 1 + f_rest.shape[1]
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()   # noqa: E731
        xyz_, dxyz_, sc_, ro_, drot_, op_, dc_, fr_ = map(c, (xyz, dxyz, scaling, rotation, drot, opacity, f_dc, f_rest))
        f32 = dict(dtype=torch.float32, device=dev)
        means3D = torch.empty(P, 3, **f32)
        scales = torch.empty(P, 3, **f32)
        rots = torch.empty(P, 4, **f32)
        opac = torch.empty(P, 1, **f32)
        shs = torch.empty(P, K, 3, **f32) if f_dc is not None else None
        with torch.cuda.device(dev):
            _lib.check(L.rdg_activate_forward(P, K, _lib.ptr(xyz_), _lib.ptr(dxyz_), _lib.ptr(sc_), _lib.ptr(ro_),
                                              _lib.ptr(drot_), _lib.ptr(op_), _lib.ptr(dc_), _lib.ptr(fr_),
                                              _lib.ptr(means3D), _lib.ptr(scales), _lib.ptr(rots), _lib.ptr(opac),
                                              _lib.ptr(shs), _lib.stream_ptr()), "rdg_activate_forward")
        ctx.save_for_backward(sc_, ro_, op_)
        ctx.sinks = sinks
        ctx.split = f_dc is not None
        ctx.shapes = (xyz.shape, scaling.shape, rotation.shape, opacity.shape) + (
            (f_dc.shape, f_rest.shape) if f_dc is not None else ())
        ctx.has_d = (dxyz is not None, drot is not None)
        ctx.K = K
        if shs is None:
            shs = torch.empty(0, **f32)
            ctx.mark_non_differentiable(shs)
        return means3D, scales, rots, opac, shs

    @staticmethod
    def backward(ctx, g_m, g_s, g_r, g_o, g_sh):
        L = _lib.lib()
        sc_, ro_, op_ = ctx.saved_tensors
        dev = sc_.device
        P = sc_.shape[0]
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()   # noqa: E731
        g_m, g_s, g_r, g_o, g_sh = map(c, (g_m, g_s, g_r, g_o, g_sh))
        sinks = ctx.sinks
        names = ("xyz", "scaling", "rotation", "opacity") + (("f_dc", "f_rest") if ctx.split else ())
        if sinks is None:
            outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.shapes]
        else:
            outs = [sinks[n] for n in names]
        if not ctx.split:
            outs = outs + [None, None]
            g_sh = None
        with torch.cuda.device(dev):
            _lib.check(L.rdg_activate_backward(P, ctx.K, _lib.ptr(sc_), _lib.ptr(ro_), _lib.ptr(op_), _lib.ptr(g_m),
                                               _lib.ptr(g_s), _lib.ptr(g_r), _lib.ptr(g_o), _lib.ptr(g_sh),
                                               *[_lib.ptr(o) for o in outs], _lib.stream_ptr()),
                       "rdg_activate_backward")
        d_dxyz = g_m if ctx.has_d[0] else None
        d_drot = g_r if ctx.has_d[1] else None
        if sinks is None:
            return outs[0], d_dxyz, outs[1], outs[2], d_drot, outs[3], outs[4], outs[5], None
        return None, d_dxyz, None, None, d_drot, None, None, None, None


def activate_gaussians(xyz, dxyz, scaling, rotation, drot, opacity, f_dc, f_rest,
                       grad_sinks: Optional[Dict[str, torch.Tensor]] = None):
    """(means3D, scales, rotations, opacities, shs) ready for the rasterizer.

    With ``grad_sinks`` (name -> tensor for xyz, scaling, rotation, opacity, f_dc, f_rest) the backward pass
    OVERWRITES those tensors with the parameter gradients instead of returning them through autograd -- used to
    write straight into the flat gradient bucket of ``rodygs_amd.dp.FlatParams`` (no AccumulateGrad copies)."""
    return _Activate.apply(xyz, dxyz, scaling, rotation, drot, opacity, f_dc, f_rest, grad_sinks)


class _GSProperties(torch.autograd.Function):
    """Static and dynamic clouds activated straight into the two segments of ONE set of rasterizer inputs."""

    @staticmethod
    def forward(ctx, s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_frest,
                d_xyz, d_scaling, d_rotation, d_opacity, d_fdc, d_frest, dxyz, drot):
        L = _lib.lib()
        if not s_xyz.is_cuda:
            raise RuntimeError("rodygs_amd.gs_properties: tensors must be on the GPU (no CPU fallback exists)")
        dev = s_xyz.device
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()   # noqa: E731
        S = list(map(c, (s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_frest)))
        D = list(map(c, (d_xyz, d_scaling, d_rotation, d_opacity, d_fdc, d_frest)))
        dx, dr = c(dxyz), c(drot)
        Ps, Pd = S[0].shape[0], D[0].shape[0]
        K = 1 + S[5].shape[1]
        if 1 + D[5].shape[1] != K:
            raise RuntimeError("static and dynamic Gaussians must store the same number of SH coefficients")
        P = Ps + Pd
        f32 = dict(dtype=torch.float32, device=dev)
        means3D, scales = torch.empty(P, 3, **f32), torch.empty(P, 3, **f32)
        rots, opac, shs = torch.empty(P, 4, **f32), torch.empty(P, 1, **f32), torch.empty(P, K, 3, **f32)
        outs = (means3D, scales, rots, opac, shs)
        with torch.cuda.device(dev):
            for n, first, src, a, b in ((Ps, 0, S, None, None), (Pd, Ps, D, dx, dr)):
                if n == 0:
                    continue
                seg = [t[first:] for t in outs]      # a row offset into the shared buffers: nothing is copied
                _lib.check(L.rdg_activate_forward(n, K, _lib.ptr(src[0]), _lib.ptr(a), _lib.ptr(src[1]), _lib.ptr(src[2]),
                                                  _lib.ptr(b), _lib.ptr(src[3]), _lib.ptr(src[4]), _lib.ptr(src[5]),
                                                  *[_lib.ptr(t) for t in seg], _lib.stream_ptr()),
                           "rdg_activate_forward")
        ctx.save_for_backward(S[1], S[2], S[3], D[1], D[2], D[3])
        ctx.dims = (Ps, Pd, K)
        ctx.shapes = [t.shape for t in (s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_frest)] + \
                     [t.shape for t in (d_xyz, d_scaling, d_rotation, d_opacity, d_fdc, d_frest)]
        ctx.has_d = (dxyz is not None, drot is not None)
        ctx.set_materialize_grads(False)
        return means3D, scales, rots, opac, shs

    @staticmethod
    def backward(ctx, g_m, g_s, g_r, g_o, g_sh):
        L = _lib.lib()
        s_sc, s_ro, s_op, d_sc, d_ro, d_op = ctx.saved_tensors
        Ps, Pd, K = ctx.dims
        dev = s_sc.device
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()   # noqa: E731
        g_m, g_s, g_r, g_o, g_sh = map(c, (g_m, g_s, g_r, g_o, g_sh))
        grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        with torch.cuda.device(dev):
            for n, first, par, out in ((Ps, 0, (s_sc, s_ro, s_op), grads[0:6]), (Pd, Ps, (d_sc, d_ro, d_op), grads[6:12])):
                if n == 0:
                    continue
                gs = [None if t is None else t[first:] for t in (g_m, g_s, g_r, g_o, g_sh)]
                _lib.check(L.rdg_activate_backward(n, K, _lib.ptr(par[0]), _lib.ptr(par[1]), _lib.ptr(par[2]),
                                                   *[_lib.ptr(t) for t in gs], *[_lib.ptr(t) for t in out],
                                                   _lib.stream_ptr()), "rdg_activate_backward")
        d_dxyz = g_m[Ps:] if (ctx.has_d[0] and g_m is not None) else None      # identities: views, no kernel
        d_drot = g_r[Ps:] if (ctx.has_d[1] and g_r is not None) else None
        return (*grads, d_dxyz, d_drot)


def gs_properties(static: Dict[str, torch.Tensor], dynamic: Dict[str, torch.Tensor],
                  dyn_translation: Optional[torch.Tensor] = None, dyn_rotation: Optional[torch.Tensor] = None):
    """``RoDyGSTrainer.get_GS_properties`` (/root/reference/src/trainer/rodygs.py:68-113) without its five
    ``torch.cat`` copies (236 B per Gaussian per call): the static cloud's getters and the
    dynamic cloud's getters + deformation add write directly into the two row segments (static first, as the
    reference concatenates) of one set of rasterizer inputs; the backward hands each cloud its slice of the
    gradients by pointer offset.  ``static`` / ``dynamic``: dicts with the raw parameters ``xyz, scaling, rotation,
    opacity, f_dc, f_rest``.  Returns ``(xyz, opacity, scaling, rotation, features)`` in the reference's order."""
    k = ("xyz", "scaling", "rotation", "opacity", "f_dc", "f_rest")
    m, s_, r, o, sh = _GSProperties.apply(*[static[n] for n in k], *[dynamic[n] for n in k], dyn_translation, dyn_rotation)
    return m, o, s_, r, sh


class _PoseView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam_q, cam_t, frame, grad_sinks=None, step_scalars=None):
        L = _lib.lib()
        ctx.grad_sinks = grad_sinks
        ctx.step_scalars = step_scalars
        if not cam_q.is_cuda:
            raise RuntimeError("rodygs_amd.pose_view_matrix: tensors must be on the GPU (no CPU fallback exists)")
        q = cam_q.detach().to(torch.float32).contiguous()
        t = cam_t.detach().to(torch.float32).contiguous()
        T = q.shape[0]
        view = torch.empty(4, 4, dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            if step_scalars is not None:      # graph replay: the frame index is read from device memory
                _lib.check(L.rdg_pose_view_forward_dev(T, _lib.ptr(step_scalars), _lib.ptr(q), _lib.ptr(t),
                                                       _lib.ptr(view), _lib.stream_ptr()), "rdg_pose_view_forward_dev")
            else:
                _lib.check(L.rdg_pose_view_forward(T, int(frame), _lib.ptr(q), _lib.ptr(t), _lib.ptr(view),
                                                   _lib.stream_ptr()), "rdg_pose_view_forward")
        ctx.save_for_backward(q, t)
        ctx.frame = int(frame)
        return view

    @staticmethod
    def backward(ctx, g_view):
        L = _lib.lib()
        q, t = ctx.saved_tensors
        T = q.shape[0]
        g = g_view.to(torch.float32).contiguous()
        sinks = ctx.grad_sinks
        if sinks is not None:
            d_q, d_t = sinks["q"], sinks["t"]
            for d, ref in ((d_q, q), (d_t, t)):
                if d.shape != ref.shape or not d.is_contiguous() or d.dtype != torch.float32:
                    raise RuntimeError("pose_view_matrix grad_sinks must be contiguous float32 tensors shaped like "
                                       "cam_q / cam_t")
        else:
            d_q = torch.empty_like(q)
            d_t = torch.empty_like(t)
        # graph capture (trainstep.GraphedStep): the kernel joins the pose-gradient chain on the rasterizer state's second stream
        # (sinks["aux"] = the RasterState; everything it touches is persistent: the state's dL/dviewmatrix buffer, the sinks)
        st_ = None if sinks is None else sinks.get("aux")
        aux = None
        if st_ is not None:
            if st_.graph_capture and st_.aux_stream is not None and g.data_ptr() == st_.pose_grad.data_ptr():
                aux = st_.aux_stream
            elif st_.pose_fork_eager is not None and g.data_ptr() == st_.pose_fork_eager[1].data_ptr():
                aux = st_.pose_fork_eager[0]          # the eager step's fork: same chain, same stream, no graph
        with torch.cuda.device(q.device), (torch.cuda.stream(aux) if aux is not None else _nullcontext()):
            if ctx.step_scalars is not None:
                _lib.check(L.rdg_pose_view_backward_dev(T, _lib.ptr(ctx.step_scalars), _lib.ptr(q), _lib.ptr(t),
                                                        _lib.ptr(g), _lib.ptr(d_q), _lib.ptr(d_t), _lib.stream_ptr()),
                           "rdg_pose_view_backward_dev")
            else:
                _lib.check(L.rdg_pose_view_backward(T, ctx.frame, _lib.ptr(q), _lib.ptr(t), _lib.ptr(g), _lib.ptr(d_q),
                                                    _lib.ptr(d_t), _lib.stream_ptr()), "rdg_pose_view_backward")
        if sinks is not None:
            return None, None, None, None, None
        return d_q, d_t, None, None, None


def pose_view_matrix(cam_q: torch.Tensor, cam_t: torch.Tensor, frame: int, grad_sinks=None,
                     step_scalars=None) -> torch.Tensor:
    """W2C^T (glm storage, what the rasterizer takes as ``viewmatrix``) of frame ``frame`` from the learnable
    camera-to-world quaternions cam_q[T,4] (r,i,j,k) and translations cam_t[T,3].  ``step_scalars``: a device
    ``RdgStepScalars`` (128-byte tensor) whose ``frame`` field replaces the argument -- the form a captured hipGraph can
    replay for a different frame (rodygs_amd.trainstep.GraphedStep)."""
    return _PoseView.apply(cam_q, cam_t, frame, grad_sinks, step_scalars)
