"""Time-deformation path of RoDyGS (reference: /root/reference/src/model/rodygs_dynamic.py).

Host-side mirror of the reference interface, same names and argument meaning:
  * ``TimestepEmbedder``      (rodygs_dynamic.py:190-220)
  * ``MLPMotionBasis``        (:223-240)
  * ``MLPBasisNetwork``       (:243-327)  -- state_dict-compatible with the reference class
  * ``gaussian_deformation``  = the per-Gaussian part of ``DynRoDyGS.get_gaussian_deformation`` (:122-138)
  * ``DeformationField``      = the bookkeeping of ``DynRoDyGS`` around it (birth-time table, cache)

The tiny MLP (68 656 parameters, batch <= T+1 rows) stays in torch.  The reference runs its 16 heads as a
16-iteration Python loop over separate nn.Linear modules (~40 launches forward, ~550 tiny kernels per train step
with backward and per-parameter optimiser work); here the head weights ARE four stacked parameters
([16,32,64], [16,32], [16,7,32], [16,7]) driven by two batched matmuls, and state_dict()/load_state_dict()
translate to and from the reference's per-head key names, so checkpoints interchange.
The per-Gaussian contraction -- the only part that scales with P -- is the HIP kernel
``rdg_deform_forward/backward`` (csrc/rdg_deform.hip): it never materialises the reference's
``table[gaussian_to_time_ind]`` ([P,16,7], 448 MB at P = 1 M).
"""
from __future__ import annotations

import re
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class TimestepEmbedder(nn.Module):
    """[t, sin(t f_i), cos(t f_i)]_i with f = pi * linspace(1, 2^(m-1), m) (or 2^linspace when log_sampling)."""

    include_input: bool = True

    def __init__(self, emb_multires, input_dims, log_sampling):
        super().__init__()
        self.num_freqs = emb_multires
        self.max_freq_log2 = emb_multires - 1
        self.input_dims = input_dims
        self.log_sampling = log_sampling
        if log_sampling:
            freq = 2.0 ** torch.linspace(0.0, self.num_freqs - 1, self.num_freqs)
        else:
            freq = torch.linspace(1.0, 2.0 ** (self.num_freqs - 1), self.num_freqs)
        # same float32 values as the reference computes on every call (freq_bands * np.pi)
        self.register_buffer("freq_bands", freq * np.pi, persistent=False)

    def forward(self, timestep):
        t = torch.as_tensor(timestep, dtype=torch.float32, device=self.freq_bands.device)
        # full-range sinf/cosf (arguments reach pi*2^25): torch's device sin/cos do proper range reduction
        arg = t.unsqueeze(-1) * self.freq_bands            # [..., m]
        sc = torch.stack([torch.sin(arg), torch.cos(arg)], dim=-1).flatten(-2)  # sin f0, cos f0, sin f1, ...
        out = torch.cat([t.unsqueeze(-1), sc], dim=-1)     # [..., 2m+1]
        # the reference stacks along dim 0: a [1]-shaped t gives [2m+1, 1]
        if t.dim() >= 1:
            return out.movedim(-1, 0)
        return out


class MLPMotionBasis(nn.Module):
    """One reference head (rodygs_dynamic.py:223-240).  Kept for API parity; MLPBasisNetwork stores its heads
    stacked and does not instantiate this class."""

    def __init__(self, input_dim, output_dim, activation):
        super().__init__()
        self.basis = nn.Sequential(nn.Linear(input_dim, input_dim // 2), activation,
                                   nn.Linear(input_dim // 2, output_dim))
        for module in self.basis.modules():
            if isinstance(module, nn.Linear):
                nn.init.normal_(module.weight, mean=0, std=1e-2)
                nn.init.constant_(module.bias, 0)

    def forward(self, x):
        return self.basis(x)


FUSED_MLP = True   # GPU tensors: run MLPBasisNetwork through the HIP MFMA kernels (csrc/rdg_mlp.hip)


class _FusedMLP(torch.autograd.Function):
    """timenet + 16 heads on v_mfma_f32_16x16x4_f32: rdg_mlp_forward / rdg_mlp_backward."""

    @staticmethod
    def forward(ctx, x, W0, b0, W1, b1, W2, b2, hw1, hb1, hw2, hb2, grad_sinks=None):
        L = _lib.lib()
        ctx.grad_sinks = grad_sinks
        ps = [t.detach().to(torch.float32).contiguous() for t in (x, W0, b0, W1, b1, W2, b2, hw1, hb1, hw2, hb2)]
        x_, W0_, _, W1_, _, W2_, _, hw1_, _, hw2_, _ = ps
        NR, D0 = x_.shape
        H, NB, OUT = W0_.shape[0], hw1_.shape[0], hw2_.shape[1]
        if W1_.shape != (H, H) or W2_.shape != (H // 2, H) or hw1_.shape[1:] != (H // 4, H // 2) or \
                hw2_.shape[2] != H // 4:
            raise RuntimeError("rodygs_amd fused MLP: unexpected layer shapes")
        dev = x_.device
        with torch.cuda.device(dev):
            ws = torch.empty(L.rdg_mlp_ws_bytes(NR, H, NB), dtype=torch.uint8, device=dev)
            out = torch.empty(NR, NB, OUT, dtype=torch.float32, device=dev)
            _lib.check(L.rdg_mlp_forward(NR, D0, H, NB, OUT, *[_lib.ptr(t) for t in ps], _lib.ptr(ws), _lib.ptr(out),
                                         _lib.stream_ptr()), "rdg_mlp_forward")
        ctx.save_for_backward(x_, W1_, W2_, hw1_, hw2_, ws)
        ctx.dims = (NR, D0, H, NB, OUT)
        ctx.shapes = [t.shape for t in ps[1:]]
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        x_, W1_, W2_, hw1_, hw2_, ws = ctx.saved_tensors
        NR, D0, H, NB, OUT = ctx.dims
        g = g_out.to(torch.float32).contiguous()
        sinks = ctx.grad_sinks
        if sinks is not None:
            # the kernels overwrite every element of the ten parameter gradients: write them straight into the
            # caller's buffers (a flat gradient bucket) and hand nothing to AccumulateGrad
            if len(sinks) != len(ctx.shapes) or any(tuple(t.shape) != tuple(sh) or not t.is_contiguous() or
                                                     t.dtype != torch.float32 for t, sh in zip(sinks, ctx.shapes)):
                raise RuntimeError("MLPBasisNetwork.grad_sinks must be 10 contiguous float32 tensors shaped like "
                                   "(W0, b0, W1, b1, W2, b2, head_w1, head_b1, head_w2, head_b2)")
            grads = list(sinks)
        else:
            grads = [torch.empty(s, dtype=torch.float32, device=x_.device) for s in ctx.shapes]
        with torch.cuda.device(x_.device):
            _lib.check(L.rdg_mlp_backward(NR, D0, H, NB, OUT, _lib.ptr(x_), _lib.ptr(W1_), _lib.ptr(W2_), _lib.ptr(hw1_),
                                          _lib.ptr(hw2_), _lib.ptr(ws), _lib.ptr(g), *[_lib.ptr(t) for t in grads],
                                          _lib.stream_ptr()), "rdg_mlp_backward")
        if sinks is not None:
            return (None,) * 12
        return (None, *grads, None)


_HEAD_KEY = re.compile(r"^(.*)basis_xyz\.(\d+)\.basis\.(0|2)\.(weight|bias)$")
_STACKED = {("0", "weight"): "head_w1", ("0", "bias"): "head_b1", ("2", "weight"): "head_w2", ("2", "bias"): "head_b2"}


class MLPBasisNetwork(nn.Module):
    time_input_dim: int = 1
    trans_dim: int = 3
    rot_dim: int = 4

    def __init__(self, netwidth, num_basis, t_emb_multires, t_log_sampling, activation="gelu"):
        super().__init__()
        self.netwidth = netwidth
        self.num_basis = num_basis
        self.t_embed_dim = t_emb_multires * self.time_input_dim * 2 + self.time_input_dim
        self.t_embedder = TimestepEmbedder(t_emb_multires, self.time_input_dim, t_log_sampling)
        self.activation = nn.GELU() if activation.lower() != "relu" else nn.ReLU(inplace=False)
        self.timenet = nn.Sequential(
            nn.Linear(self.t_embed_dim, netwidth), self.activation,
            nn.Linear(netwidth, netwidth), self.activation,
            nn.Linear(netwidth, netwidth // 2), self.activation,
        )
        for module in self.timenet.modules():
            if isinstance(module, nn.Linear):
                nn.init.normal_(module.weight, mean=0, std=1e-2)
                nn.init.constant_(module.bias, 0)
        hin, hmid, hout = netwidth // 2, netwidth // 4, self.trans_dim + self.rot_dim
        # the 16 heads, stacked: reference init N(0, 1e-2) weights, zero biases (rodygs_dynamic.py:234-237)
        self.head_w1 = nn.Parameter(torch.randn(num_basis, hmid, hin) * 1e-2)
        self.head_b1 = nn.Parameter(torch.zeros(num_basis, hmid))
        self.head_w2 = nn.Parameter(torch.randn(num_basis, hout, hmid) * 1e-2)
        self.head_b2 = nn.Parameter(torch.zeros(num_basis, hout))
        self.grad_sinks = None   # optional: 10 tensors the fused backward overwrites (see _FusedMLP.backward)
        self._register_state_dict_hook(MLPBasisNetwork._to_reference_keys)
        self._register_load_state_dict_pre_hook(self._from_reference_keys)

    # -- checkpoint interchange with the reference's per-head modules ---------------------------------------------
    @staticmethod
    def _to_reference_keys(module, state_dict, prefix, local_metadata):
        for (layer, kind), name in _STACKED.items():
            t = state_dict.pop(prefix + name)
            for b in range(t.shape[0]):
                state_dict[f"{prefix}basis_xyz.{b}.basis.{layer}.{kind}"] = t[b]
        return state_dict

    def _from_reference_keys(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                             error_msgs):
        found = {}
        for k in list(state_dict.keys()):
            m = _HEAD_KEY.match(k)
            if m and m.group(1) == prefix:
                found.setdefault(_STACKED[(m.group(3), m.group(4))], {})[int(m.group(2))] = state_dict.pop(k)
        for name, parts in found.items():
            if sorted(parts) != list(range(self.num_basis)):
                error_msgs.append(f"{prefix}{name}: expected {self.num_basis} heads, found {sorted(parts)}")
                continue
            state_dict[prefix + name] = torch.stack([parts[b] for b in range(self.num_basis)])

    # -- forward ---------------------------------------------------------------------------------------------------
    def _heads(self, h: torch.Tensor) -> torch.Tensor:
        """h [N, netwidth/2] -> [N, num_basis, 7] with two batched matmuls."""
        hb = h.unsqueeze(0).expand(self.num_basis, -1, -1)
        u = self.activation(torch.baddbmm(self.head_b1.unsqueeze(1), hb, self.head_w1.transpose(1, 2)))
        o = torch.baddbmm(self.head_b2.unsqueeze(1), u, self.head_w2.transpose(1, 2))   # [B, N, 7]
        return o.transpose(0, 1)

    def motion_basis(self, t_emb: torch.Tensor) -> torch.Tensor:
        """t_emb [..., 53] -> motion basis [..., num_basis, 7].

        GPU tensors with the default GELU run the whole network as 5 MFMA launches forward / 5 backward
        (csrc/rdg_mlp.hip); the torch expression below is the same arithmetic and serves CPU tensors (checkpoint
        tooling, golden tests) and the ReLU variant."""
        lead = t_emb.shape[:-1]
        x = t_emb.reshape(-1, t_emb.shape[-1])
        if x.is_cuda and FUSED_MLP and isinstance(self.activation, nn.GELU):
            tn = self.timenet
            out = _FusedMLP.apply(x, tn[0].weight, tn[0].bias, tn[2].weight, tn[2].bias, tn[4].weight, tn[4].bias,
                                  self.head_w1, self.head_b1, self.head_w2, self.head_b2, self.grad_sinks)
            return out.reshape(*lead, self.num_basis, self.trans_dim + self.rot_dim)
        h = self.timenet(x)
        return self._heads(h).reshape(*lead, self.num_basis, self.trans_dim + self.rot_dim)

    def batch_embedding(self, timesteps):
        dev = self.t_embedder.freq_bands.device
        return torch.stack([self.t_embedder(t) for t in timesteps]).to(dev).squeeze()

    def batch_inference(self, t_embs):
        """[T,53] -> [T,num_basis,7]  (rodygs_dynamic.py:296-306)."""
        return self.motion_basis(t_embs)

    def forward(self, coeff, timestep):
        """Reference semantics (:308-327): (translation [P,3], rotation [P,4]) = squeeze(coeff) @ basis(t)."""
        t_emb = self.t_embedder(timestep)
        basis = self.motion_basis(t_emb.reshape(1, -1)).squeeze(0)      # [B,7]
        c = coeff.reshape(coeff.shape[0], -1)
        ind = torch.zeros(c.shape[0], dtype=torch.int64, device=c.device)
        return gaussian_deformation(c, ind, basis, None, 1.0)


_ORDER_CACHE = {}


def _birth_order(time_ind: torch.Tensor, n_births: int = 0):
    """(order, inverse, seg_start): int32 permutation sorting the Gaussians by birth index, its inverse, and (for
    n_births > 0) where each birth index starts in the sorted sequence (int32 [n_births + 1]); cached until
    time_ind changes (it only does at densification), so the sort is not part of the step.  The key is the tensor's
    address / version / length, and the entry keeps the tensor alive: a live tensor's address cannot be handed to another
    one, so a key never matches a different birth-index array (the dB reduction REQUIRES a truly sorted sequence: its
    kernel checks that and poisons the gradient with NaN otherwise).  Code that replaces a birth-index tensor
    (densification does) calls ``invalidate_birth_order_cache()`` to drop the old entries."""
    key = (time_ind.data_ptr(), time_ind._version, time_ind.shape[0], str(time_ind.device), int(n_births))
    o = _ORDER_CACHE.get(key)
    if o is None:
        if len(_ORDER_CACHE) > 8:
            _ORDER_CACHE.clear()
        order = _stable_order(time_ind, n_births)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)
        seg = None
        if n_births > 0:
            seg = torch.searchsorted(time_ind[order].contiguous(),
                                     torch.arange(n_births + 1, device=order.device, dtype=time_ind.dtype))
            seg = seg.to(torch.int32).contiguous()
        o = (order.to(torch.int32).contiguous(), inv.to(torch.int32).contiguous(), seg, time_ind)
        _ORDER_CACHE[key] = o
    return o[:3]


def _stable_order(time_ind: torch.Tensor, n_births: int) -> torch.Tensor:
    """Stable ascending order of the birth indices (int64 [P]): on the GPU the library's LSD radix sort over the bits the
    indices have (rdg_sort_pairs; the product path does not go through the framework's sort), the framework's on the CPU."""
    if time_ind.is_cuda and 0 < time_ind.numel() < 2 ** 31:
        from .rigidity import _sort_by_key
        hi = int(n_births) if n_births > 0 else int(time_ind.max()) + 1
        return _sort_by_key(time_ind, max(1, int(hi).bit_length()))[1]
    return torch.argsort(time_ind, stable=True)


def invalidate_birth_order_cache() -> None:
    _ORDER_CACHE.clear()


def refresh_birth_order_inplace(time_ind: torch.Tensor, n_births: int) -> None:
    """``time_ind`` was modified IN PLACE (a fixed-capacity densification): recompute the sorted order and write it into the SAME
    (order, inverse, seg_start) tensors the cache holds for it -- a captured graph reads them by address --, and re-key the entry
    to the tensor's new version.  Creates the entry if there is none."""
    old_key = next((k for k in _ORDER_CACHE if len(k) == 5 and k[0] == time_ind.data_ptr() and k[2] == time_ind.shape[0]
                    and k[4] == int(n_births)), None)
    if old_key is None:
        _birth_order(time_ind, n_births)
        return
    order_t, inv_t, seg_t, _ = _ORDER_CACHE.pop(old_key)
    order = _stable_order(time_ind, n_births)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=order.device)
    order_t.copy_(order.to(torch.int32))
    inv_t.copy_(inv.to(torch.int32))
    if seg_t is not None:
        seg = torch.searchsorted(time_ind[order].contiguous(), torch.arange(n_births + 1, device=order.device, dtype=time_ind.dtype))
        seg_t.copy_(seg.to(torch.int32))
    key = (time_ind.data_ptr(), time_ind._version, time_ind.shape[0], str(time_ind.device), int(n_births))
    _ORDER_CACHE[key] = (order_t, inv_t, seg_t, time_ind)


def _identity_order(P: int, dev) -> torch.Tensor:
    key = ("id", P, str(dev))
    o = _ORDER_CACHE.get(key)
    if o is None:
        o = torch.arange(P, dtype=torch.int32, device=dev)
        _ORDER_CACHE[key] = o
    return o


class _DeformFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coeff, time_ind, basis_t, table, scale, grad_sinks=None, packed=None):
        L = _lib.lib()
        if not coeff.is_cuda:
            raise RuntimeError("rodygs_amd.gaussian_deformation: tensors must be on the GPU (no CPU fallback exists)")
        c = coeff.detach().to(torch.float32).contiguous()
        P, B = c.shape
        ctx.grad_sinks = grad_sinks
        ctx.packed = packed is not None
        if packed is not None:
            # bases [Tu+1,B,7]: rows 0..Tu-1 = the birth-time table, row Tu = B(t); ONE gradient tensor comes back
            pk = packed.detach().to(torch.float32).contiguous()
            basis_t, table = pk[-1], pk[:-1]
        bt = basis_t.detach().to(torch.float32).contiguous()
        tb = None if table is None else table.detach().to(torch.float32).contiguous()
        ti = time_ind.detach().to(torch.int64).contiguous()
        Tu = 0 if tb is None else tb.shape[0]
        dxyz = torch.empty(P, 3, dtype=torch.float32, device=c.device)
        drot = torch.empty(P, 4, dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            _lib.check(L.rdg_deform_forward(P, B, Tu, _lib.ptr(c), _lib.ptr(ti), _lib.ptr(bt), _lib.ptr(tb),
                                            float(scale), _lib.ptr(dxyz), _lib.ptr(drot), _lib.stream_ptr()),
                       "rdg_deform_forward")
        ctx.save_for_backward(c, ti, bt, tb)
        ctx.scale = float(scale)
        return dxyz, drot

    @staticmethod
    def backward(ctx, g_xyz, g_rot):
        L = _lib.lib()
        c, ti, bt, tb = ctx.saved_tensors
        P, B = c.shape
        Tu = 0 if tb is None else tb.shape[0]
        dev = c.device
        g_xyz = (torch.zeros(P, 3, device=dev) if g_xyz is None else g_xyz).to(torch.float32).contiguous()
        g_rot = (torch.zeros(P, 4, device=dev) if g_rot is None else g_rot).to(torch.float32).contiguous()
        sink_c = None if ctx.grad_sinks is None else ctx.grad_sinks.get("coeff")
        if sink_c is not None:
            if sink_c.numel() != c.numel() or not sink_c.is_contiguous() or sink_c.dtype != torch.float32:
                raise RuntimeError("grad_sinks['coeff'] must be a contiguous float32 tensor with coeff's element count")
            d_c = sink_c.view(P, B)
        else:
            d_c = torch.empty_like(c)
        if ctx.packed:
            d_pk = torch.empty(Tu + 1, B, bt.shape[-1], dtype=torch.float32, device=dev)
            d_bt, d_tb = d_pk[-1], d_pk[:-1]
        else:
            d_pk = None
            d_bt = torch.empty_like(bt)
            d_tb = None if tb is None else torch.empty_like(tb)
        with torch.cuda.device(dev):
            if tb is not None:
                order, inv, seg = _birth_order(ti, Tu)
                sws = torch.empty(L.rdg_deform_sorted_ws_bytes(P), dtype=torch.uint8, device=dev)
            else:
                order, inv, seg, sws = _identity_order(P, dev), None, None, None
            _lib.check(L.rdg_deform_backward(P, B, Tu, _lib.ptr(c), _lib.ptr(ti), _lib.ptr(bt), _lib.ptr(tb),
                                             ctx.scale, _lib.ptr(g_xyz), _lib.ptr(g_rot), _lib.ptr(d_c),
                                             _lib.ptr(d_bt), _lib.ptr(d_tb), _lib.ptr(order), _lib.ptr(inv),
                                             _lib.ptr(seg), _lib.ptr(sws), _lib.stream_ptr()),
                       "rdg_deform_backward")
        if sink_c is not None:
            d_c = None
        if ctx.packed:
            return d_c, None, None, None, None, None, d_pk
        return d_c, None, d_bt, d_tb, None, None, None


def gaussian_deformation(coeff: torch.Tensor, time_ind: torch.Tensor, basis_t: torch.Tensor,
                         table: Optional[torch.Tensor], spatial_lr_scale: float, grad_sinks=None):
    """(scaled_translation [P,3], rotation_delta [P,4]) = coeff . (B(t) - B_table[time_ind]).

    coeff [P,B] (or the reference's [P,1,B]); basis_t [B,7]; table [Tu,B,7] or None (inverse_motion=False);
    translation is multiplied by ``spatial_lr_scale`` exactly as rodygs_dynamic.py:136.
    ``grad_sinks={"coeff": t}``: the backward overwrites ``t`` with dL/dcoeff instead of returning it."""
    c = coeff.reshape(coeff.shape[0], -1)
    return _DeformFn.apply(c, time_ind, basis_t, table, spatial_lr_scale, grad_sinks, None)


def gaussian_deformation_packed(coeff: torch.Tensor, time_ind: torch.Tensor, bases: torch.Tensor,
                                spatial_lr_scale: float, grad_sinks=None):
    """Same op with the MLP output used as it comes: ``bases`` [Tu+1,B,7] = the birth-time table followed by B(t)
    (one MLP pass over the Tu+1 embedding rows).  Its gradient comes back as ONE tensor, so autograd has no
    slice-backward (zeros + copy + add) to run."""
    c = coeff.reshape(coeff.shape[0], -1)
    return _DeformFn.apply(c, time_ind, None, None, spatial_lr_scale, grad_sinks, bases)


class _DynamicGetter(torch.autograd.Function):
    """rdg_dyn_getter_forward / backward: deformation + activations fused (see ``dynamic_gaussians``)."""

    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, coeff, time_ind, bases, scale, grad_sinks):
        L = _lib.lib()
        if not xyz.is_cuda:
            raise RuntimeError("rodygs_amd.dynamic_gaussians: tensors must be on the GPU (no CPU fallback exists)")
        f = lambda t: t.detach().to(torch.float32).contiguous()   # noqa: E731
        x, sc, ro, op, c, bs = f(xyz), f(scaling), f(rotation), f(opacity), f(coeff), f(bases)
        ti = time_ind.detach().to(torch.int64).contiguous()
        P, Tu, dev = x.shape[0], bs.shape[0] - 1, x.device
        f32 = dict(dtype=torch.float32, device=dev)
        means3D, scales = torch.empty(P, 3, **f32), torch.empty(P, 3, **f32)
        rots, opac = torch.empty(P, 4, **f32), torch.empty(P, 1, **f32)
        with torch.cuda.device(dev):
            _lib.check(L.rdg_dyn_getter_forward(P, Tu, _lib.ptr(c), _lib.ptr(ti), _lib.ptr(bs), float(scale), _lib.ptr(x),
                                                _lib.ptr(sc), _lib.ptr(ro), _lib.ptr(op), _lib.ptr(means3D),
                                                _lib.ptr(scales), _lib.ptr(rots), _lib.ptr(opac), _lib.stream_ptr()),
                       "rdg_dyn_getter_forward")
        ctx.save_for_backward(sc, ro, op, c, ti, bs)
        ctx.scale, ctx.grad_sinks = float(scale), grad_sinks
        ctx.set_materialize_grads(False)
        return means3D, scales, rots, opac

    @staticmethod
    def backward(ctx, g_m, g_s, g_r, g_o):
        L = _lib.lib()
        sc, ro, op, c, ti, bs = ctx.saved_tensors
        P, Tu, dev = sc.shape[0], bs.shape[0] - 1, sc.device
        g = lambda t: None if t is None else t.to(torch.float32).contiguous()   # noqa: E731
        g_m, g_s, g_r, g_o = g(g_m), g(g_s), g(g_r), g(g_o)
        sinks = ctx.grad_sinks or {}
        f32 = dict(dtype=torch.float32, device=dev)
        shapes = {"xyz": (P, 3), "scaling": (P, 3), "rotation": (P, 4), "opacity": (P, 1), "coeff": (P, 16)}
        out = {}
        for k, shp in shapes.items():
            t = sinks.get(k)
            if t is not None and (t.numel() != shp[0] * shp[1] or not t.is_contiguous() or t.dtype != torch.float32):
                raise RuntimeError(f"dynamic_gaussians grad_sinks[{k!r}] must be a contiguous float32 tensor of "
                                   f"{shp[0] * shp[1]} elements")
            out[k] = t if t is not None else torch.empty(*shp, **f32)
        d_bases = torch.empty_like(bs)
        with torch.cuda.device(dev):
            order, inv, seg = _birth_order(ti, Tu)
            sws = torch.empty(L.rdg_deform_sorted_ws_bytes(P), dtype=torch.uint8, device=dev)
            _lib.check(L.rdg_dyn_getter_backward(P, Tu, _lib.ptr(c), _lib.ptr(ti), _lib.ptr(bs), ctx.scale, _lib.ptr(sc),
                                                 _lib.ptr(ro), _lib.ptr(op), _lib.ptr(g_m), _lib.ptr(g_s), _lib.ptr(g_r),
                                                 _lib.ptr(g_o), _lib.ptr(out["xyz"]), _lib.ptr(out["scaling"]),
                                                 _lib.ptr(out["rotation"]), _lib.ptr(out["opacity"]),
                                                 _lib.ptr(out["coeff"]), _lib.ptr(d_bases), _lib.ptr(order), _lib.ptr(inv),
                                                 _lib.ptr(seg), _lib.ptr(sws), _lib.stream_ptr()),
                       "rdg_dyn_getter_backward")
        ret = [None if k in sinks and sinks[k] is not None else out[k] for k in ("xyz", "scaling", "rotation", "opacity")]
        d_c = None if sinks.get("coeff") is not None else out["coeff"]
        after = sinks.get("after_rows")
        if after is not None:
            # every per-Gaussian gradient of the step is in its sink now; what is left of backward is the MLP's (d_bases).
            # The owner may start on the rows here (trainstep: their Adam launch on a second stream, next to the MLP backward)
            after()
        return ret[0], ret[1], ret[2], ret[3], d_c, None, d_bases, None, None


def dynamic_getter_supported(num_basis: int, num_birth_times: int) -> bool:
    return bool(_lib.lib().rdg_dyn_getter_supported(int(num_basis), int(num_birth_times)))


def dynamic_gaussians(xyz, scaling, rotation, opacity, coeff, time_ind, bases, spatial_lr_scale, grad_sinks=None):
    """Deformed + activated dynamic Gaussians in one kernel each way:
    ``means3D = xyz + spatial_lr_scale * (c . dB)[:, :3]``, ``scales = exp(scaling)``,
    ``rots = normalize(rotation) + (c . dB)[:, 3:]``, ``opac = sigmoid(opacity)``, ``dB = bases[-1] - bases[birth]``
    -- what rodygs.py:68-113 builds from get_gaussian_deformation (rodygs_dynamic.py:122-138) and the getters
    (rodygs_static.py:82-105).  ``bases`` [Tu+1,16,7] as for ``gaussian_deformation_packed``; ``coeff`` [P,16] or
    [P,1,16].  ``grad_sinks``: optional {"xyz","scaling","rotation","opacity","coeff"} tensors the backward
    overwrites instead of returning the gradients."""
    c = coeff.reshape(coeff.shape[0], -1)
    return _DynamicGetter.apply(xyz, scaling, rotation, opacity, c, time_ind, bases, spatial_lr_scale, grad_sinks)


class DeformationField(nn.Module):
    """The deformation bookkeeping of ``DynRoDyGS`` (rodygs_dynamic.py:56-147) around the HIP op.

    gaussian_to_time: float birth time of every Gaussian; unique birth times are keyed by
    ``int(trunc(float32(t) * 1000))`` exactly as ``timetokey`` (:44)."""

    def __init__(self, num_gaussians: int, gaussian_to_time: torch.Tensor, netwidth=128, num_basis=16,
                 t_emb_multires=26, t_log_sampling=False, inverse_motion=True, spatial_lr_scale=1.0,
                 activation="gelu", device="cuda"):
        super().__init__()
        self.inverse_motion = inverse_motion
        self.spatial_lr_scale = float(spatial_lr_scale)
        self._deform_network = MLPBasisNetwork(netwidth, num_basis, t_emb_multires, t_log_sampling,
                                               activation=activation).to(device)
        self._motion_coeff = nn.Parameter(torch.zeros(num_gaussians, 1, num_basis, device=device))
        self.gaussian_to_time = gaussian_to_time.to(torch.float32).to(device)
        self.temporal_motion_table = None
        self.sync_gaussian_to_time_ind()
        self._time_batch_embeddings = self._deform_network.batch_embedding(self.real_times)
        if self._time_batch_embeddings.dim() == 1:
            self._time_batch_embeddings = self._time_batch_embeddings.unsqueeze(0)

    @staticmethod
    def timetokey(time) -> int:
        return int(torch.trunc(torch.tensor(time, dtype=torch.float32) * 1000).item())

    def sync_gaussian_to_time_ind(self):
        self.real_times, _ = torch.sort(torch.unique(self.gaussian_to_time))
        keys = torch.trunc(self.gaussian_to_time * 1000).to(torch.int64)
        uniq = torch.unique(keys)  # sorted
        self.unique_times = uniq.tolist()
        self.gaussian_to_time_ind = torch.searchsorted(uniq, keys)

    def get_total_motion_table(self):
        if self.temporal_motion_table is None:
            self.temporal_motion_table = self._deform_network.batch_inference(self._time_batch_embeddings)
        return self.temporal_motion_table

    def clean_motion_table(self):
        self.temporal_motion_table = None

    def motion_at(self, time):
        """(basis_t [B,7], table [Tu,B,7] or None) from ONE pass of the MLP over the Tu+1 time rows."""
        net = self._deform_network
        t_emb = net.t_embedder(torch.as_tensor(time, dtype=torch.float32).reshape(())).reshape(1, -1)
        if not self.inverse_motion:
            return net.motion_basis(t_emb).squeeze(0), None
        if self.temporal_motion_table is not None:
            return net.motion_basis(t_emb).squeeze(0), self.temporal_motion_table
        allb = net.motion_basis(torch.cat([self._time_batch_embeddings, t_emb], dim=0))
        self.temporal_motion_table = allb[:-1]
        return allb[-1], self.temporal_motion_table

    def get_gaussian_deformation(self, time):
        basis_t, table = self.motion_at(time)
        return gaussian_deformation(self._motion_coeff, self.gaussian_to_time_ind, basis_t, table,
                                    self.spatial_lr_scale)
