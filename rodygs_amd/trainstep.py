"""Minimal RoDyGS dynamic train step around the hot path ("build re-states a minimal train
step for configs 3-5; full trainer out of scope").  It is the caller bench.py times, not a trainer replacement.

One step = what /root/reference/src/trainer/rodygs.py:198-369 does for a dynamic sub-step on one camera:
  time-deformation (MLP basis in torch + HIP per-Gaussian contraction, rodygs_dynamic.py:122-138)
  -> activations (exp / sigmoid / normalize, rodygs_static.py:82-105)
  -> rasterize (HIP, renderer.py:87-101) -> 0.8 L1 + 0.2 D-SSIM (train_kubric_mrig.yaml:134-146)
  -> backward (HIP) -> [frame-DP: one all-reduce of the flat gradient bucket] -> Adam (fused HIP, eps 1e-15).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from .deform import (MLPBasisNetwork, dynamic_gaussians, dynamic_getter_supported, gaussian_deformation,
                     gaussian_deformation_packed)
from .dp import BucketedAllReduce, FlatParams, allreduce_sum_, frame_for
from .losses import fused_photometric_loss
from .model_ops import activate_gaussians, pose_view_matrix
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, RasterState


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """Same convention as /root/reference/src/utils/graphic_utils.py:76-102 (two_s = 2/|q|^2)."""
    r, i, j, k = q[0], q[1], q[2], q[3]
    two_s = 2.0 / (q * q).sum()
    return torch.stack([
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)]).reshape(3, 3)


def world_view_transform(q_c2w: torch.Tensor, t_c2w: torch.Tensor) -> torch.Tensor:
    """FixedCameraTorch.world_view_transform (/root/reference/src/data/utils.py:161-170)."""
    R_w2c = quaternion_to_matrix(q_c2w).transpose(0, 1)
    T_w2c = -(R_w2c @ t_c2w)
    top = torch.cat([R_w2c, T_w2c.unsqueeze(1)], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=q_c2w.device, dtype=q_c2w.dtype)
    return torch.cat([top, bottom], dim=0)


def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
             max_steps: int = 1000000) -> float:
    """The xyz learning-rate schedule of the reference (get_expon_lr_func,
    /root/reference/src/utils/general_utils.py:40-73; used by update_learning_rate,
    /root/reference/src/trainer/rodygs_static.py:143-149): log-linear from lr_init to lr_final over max_steps, with the
    optional reverse-cosine warm-up.  Feed the value to ``fused_adam_(..., lr_override={"xyz": lr})``."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def fused_adam_(fp: FlatParams, lr_scale: float = 1.0, betas=(0.9, 0.999), eps=1e-15, row_lr=None, extra=(),
                names=None, advance: bool = True, lr_override=None, step_scalars=None) -> None:
    """ONE fused HIP launch for all parameter groups of the flat buffers (rdg_adam_step_multi).
    lr_override: {segment name: lr} for THIS step only (the reference re-sets the xyz group's lr every iteration,
    rodygs_static.py:143-149); segments not named keep ``fp.lr``.
    row_lr: {name: (row_len, head_len, lr_tail)} for segments whose rows mix two learning rates.
    extra: further FlatParams stepped by the same launch (e.g. the MLP + camera-pose bucket); neighbouring
    segments of one buffer with the same learning rate are merged (their alignment padding has zero gradients).
    names: restrict ``fp`` to these segments (one piece of an overlapped gradient exchange); ``advance`` = this call
    opens a new optimiser step (the step counter moves once per step, however many pieces it is applied in).
    step_scalars: a device ``RdgStepScalars`` tensor -- the bias corrections are read from it instead of being computed
    from the step counter (the form a captured hipGraph can replay, see GraphedStep)."""
    L = _lib.lib()
    if advance:
        fp.step_count += 1
    for f in extra:
        f.step_count = fp.step_count
    # the segment table only depends on the layout and the rates: built once per distinct call shape (it is ~60 us of
    # Python per step otherwise -- a tenth of the whole step at the sizes where the step is host-bound)
    key = (tuple(names) if names is not None else None, lr_scale, tuple(sorted(lr_override.items())) if lr_override else None,
           tuple(sorted((k, tuple(v)) for k, v in row_lr.items())) if row_lr else None, step_scalars is not None,
           tuple((id(f), f.flat.data_ptr(), f.flat_grad.data_ptr(), f.exp_avg.data_ptr(), f.exp_avg_sq.data_ptr(),
                  tuple(f.lr.values())) for f in (fp, *extra)))
    cache = fp.__dict__.setdefault("_adam_segs", {})
    hit = cache.get(key)
    if hit is None:
        entries = []   # [flat-params, first offset, element count, lr, row_len, head_len, lr_tail, names]
        for f in (fp, *extra):
            for k in f.names:
                if f is fp and names is not None and k not in names:
                    entries.append(None)      # a gap: the neighbours must not be merged across it
                    continue
                o, n = f.offsets[k]
                lr_k = float(lr_override[k]) if (f is fp and lr_override and k in lr_override) else f.lr[k]
                row_len, head_len, lr_tail = ((row_lr or {}).get(k, (1, 1, lr_k)) if f is fp else (1, 1, lr_k))
                last = entries[-1] if entries else None
                # graph replay: the groups of ``fp`` keep a segment each -- their rates are refreshed per step from device
                # memory (RdgStepScalars.seg_lr_*), and a schedule may move one of two groups that share a rate today
                mergeable = step_scalars is None or f is not fp
                if (mergeable and last is not None and last[0] is f and last[4] == 1 and row_len == 1
                        and last[3] == lr_k):
                    last[2] = o + n - last[1]
                    last[7].append(k)
                else:
                    entries.append([f, o, n, lr_k, row_len, head_len, lr_tail, [k]])
        entries = [e for e in entries if e is not None]
        if len(cache) > 64:
            cache.clear()
        segs = None
        if entries:
            segs = (_lib.RdgAdamSeg * len(entries))()
            for i, (f, o, n, lr, row_len, head_len, lr_tail, _) in enumerate(entries):
                b = o * 4
                segs[i].n = n
                segs[i].param = f.flat.data_ptr() + b
                segs[i].grad = f.flat_grad.data_ptr() + b
                segs[i].exp_avg = f.exp_avg.data_ptr() + b
                segs[i].exp_avg_sq = f.exp_avg_sq.data_ptr() + b
                segs[i].lr_head = lr * lr_scale
                segs[i].lr_tail = lr_tail * lr_scale
                segs[i].row_len = row_len
                segs[i].head_len = head_len
        # (bucket, names of the groups merged into the segment, name whose row_lr gives the tail rate or None)
        layout = [(e[0], tuple(e[7]), e[7][0] if e[4] > 1 else None) for e in entries]
        hit = cache[key] = (segs, len(entries), layout)
    segs, n_ent, layout = hit
    if step_scalars is not None:
        fp.__dict__["_adam_graph_layout"] = (layout, lr_scale)      # GraphedStep fills the device rate table from it
    if n_ent == 0:
        return
    if step_scalars is not None:
        _lib.check(L.rdg_adam_step_multi_dev(n_ent, segs, betas[0], betas[1], eps, _lib.ptr(step_scalars),
                                             _lib.stream_ptr()), "rdg_adam_step_multi_dev")
        return
    _lib.check(L.rdg_adam_step_multi(n_ent, segs, betas[0], betas[1], eps, fp.step_count, _lib.stream_ptr()),
               "rdg_adam_step_multi")


_LOW_PRIORITY_STREAMS: dict = {}


def _low_priority_stream(device):
    """A stream of the LOWEST priority the device offers (hipStreamCreateWithPriority through the HIP runtime torch has loaded,
    wrapped as an ExternalStream; ONE per device and process, shared by every scene and never destroyed).  The framework's own
    pool only hands out streams of the default priority or higher, and the work put here -- an HBM-bound launch that fills every
    CU -- must yield to the chain of small dependent launches on the caller's stream, not starve it."""
    import ctypes
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _LOW_PRIORITY_STREAMS.get(idx)
    if st is not None:
        return st
    st = None
    try:
        # the HIP runtime THIS process already runs on (torch's; a stream made by another copy of the runtime would be a foreign
        # handle): the mapped file, by path -- the loader hands back the loaded instance for it
        path = "libamdhip64.so"
        try:
            with open("/proc/self/maps") as maps:
                for line in maps:
                    if "libamdhip64.so" in line:
                        path = line.split(None, 5)[-1].strip()
                        break
        except OSError:
            pass
        hip = ctypes.CDLL(path)
        least, greatest = ctypes.c_int(0), ctypes.c_int(0)
        with torch.cuda.device(idx):
            if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0:
                h = ctypes.c_void_p()
                prio = int(os.environ.get("RDG_SIDE_STREAM_PRIORITY", str(least.value)))
                if hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, prio) == 0 and h.value:   # 1 = hipStreamNonBlocking
                    st = torch.cuda.ExternalStream(h.value, device=idx)
    except OSError:
        st = None
    if st is None:
        st = torch.cuda.Stream(device=idx)      # the runtime would not give one: a pool stream of the default priority
    _LOW_PRIORITY_STREAMS[idx] = st
    return st


_FUSE_SH_ADAM = os.environ.get("RDG_FUSE_SH_ADAM", "1") != "0"
_EARLY_ROWS_ADAM = os.environ.get("RDG_EARLY_ROWS_ADAM", "1") != "0"   # the rows' Adam launch next to the MLP backward (eager step)
_EAGER_POSE_FORK = os.environ.get("RDG_EAGER_POSE_FORK", "1") != "0"   # the pose-gradient chain as a branch of the eager step
_PLAIN_FULL_FAST = os.environ.get("RDG_PLAIN_FULL_FAST", "1") != "0"   # 0: every full-loss step on the general path

_MLP_SINK_ORDER = ("timenet.0.weight", "timenet.0.bias", "timenet.2.weight", "timenet.2.bias", "timenet.4.weight",
                   "timenet.4.bias", "head_w1", "head_b1", "head_w2", "head_b2")


def bind_module_to_flat(net: torch.nn.Module, lr: float, device, extra_spec=None) -> FlatParams:
    """Move every parameter of ``net`` into one FlatParams bucket (values kept): the module's Parameters become
    views of the flat buffer, so ONE all-reduce covers their gradients and ONE Adam segment steps them."""
    spec = {name: (tuple(p.shape), lr) for name, p in net.named_parameters()}
    spec.update(extra_spec or {})
    sp = FlatParams(spec, device)
    with torch.no_grad():
        for name, p in list(net.named_parameters()):
            sp[name].copy_(p)
            mod_name, _, leaf = name.rpartition(".")
            mod = net.get_submodule(mod_name) if mod_name else net
            newp = torch.nn.Parameter(sp[name].detach())      # shares the flat storage
            newp.grad = sp[name].grad                         # ... and the flat gradient bucket
            mod._parameters[leaf] = newp
    return sp


class DynamicScene:
    """1 cloud of dynamic Gaussians + deformation MLP + per-frame learnable camera poses, synthetic data of
    SURVEY.md §8d (seeded), replicated on every rank."""

    def __init__(self, scene: dict, num_frames: int = 100, sh_degree: int = 3, device="cuda", seed: int = 777,
                 spatial_lr_scale: float = 5.0, orbit_deg: float = 15.0, full_losses: bool = False,
                 spatial_order: bool = False):
        g = torch.Generator().manual_seed(seed + 1)
        dev = torch.device(device)
        if spatial_order:
            # rows along the Z curve of the canonical positions (rodygs_amd/layout.py): same cloud, coherent memory
            from .layout import morton_order
            perm = morton_order(scene["means3D"])
            scene = dict(scene)
            for k in ("means3D", "shs", "scales", "rotations", "opacities"):
                scene[k] = scene[k][perm].contiguous()
        self.spatial_order = bool(spatial_order)
        self.device = dev
        self.W, self.H = scene["W"], scene["H"]
        self.tanfovx, self.tanfovy = scene["tanfovx"], scene["tanfovy"]
        self.sh_degree = sh_degree
        self.T = num_frames
        self.spatial_lr_scale = spatial_lr_scale
        P = scene["means3D"].shape[0]
        self.P = P
        K = scene["shs"].shape[1]
        spec = {
            "xyz": ((P, 3), 0.00016 * spatial_lr_scale),
            # SH features as ONE [P,K,3] tensor (what the rasterizer consumes): no cat forward, no split backward.
            # Row-structured Adam keeps the reference's two groups: f_dc at feature_lr, f_rest at feature_lr/20.
            "features": ((P, K, 3), 0.0025),
            "scaling": ((P, 3), 0.001),
            "rotation": ((P, 4), 0.001),
            "opacity": ((P, 1), 0.05),
            "motion_coeff": ((P, 1, 16), 0.00016),
        }
        fp = FlatParams(spec, dev)
        with torch.no_grad():
            fp["xyz"].copy_(scene["means3D"])
            fp["features"].copy_(scene["shs"])
            fp["scaling"].copy_(torch.log(scene["scales"]))
            fp["rotation"].copy_(scene["rotations"])
            op = scene["opacities"].clamp(1e-4, 1 - 1e-4)
            fp["opacity"].copy_(torch.log(op / (1 - op)))
            fp["motion_coeff"].copy_(0.1 * torch.randn(P, 1, 16, generator=g))
        self.fp = fp
        self.row_lr = {"features": (K * 3, 3, 0.0025 / 20.0)}
        self.time_ind = torch.randint(0, num_frames, (P,), generator=g).to(dev)
        # the MLP initialises from the GLOBAL random stream: pin it to the scene seed (without disturbing the caller's
        # stream) so that every rank of a frame-DP job builds the same network
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed + 2)
            self.net = MLPBasisNetwork(128, 16, 26, False)
        self.net = self.net.to(dev)
        self.times = torch.arange(num_frames, dtype=torch.float32) / num_frames
        self.time_batch_embeddings = self.net.batch_embedding(self.times.to(dev))
        self.frame_embeddings = self.time_batch_embeddings  # frame i is rendered at time i/T (same rows)
        # cameras: orbit of +-orbit_deg about the scene centroid (z = 11 on the optical axis)
        cz = 11.0
        qs, ts = [], []
        for i in range(num_frames):
            a = math.radians(orbit_deg) * math.sin(2 * math.pi * i / num_frames)
            # camera-to-world: rotate about y by a around the centroid
            q = torch.tensor([math.cos(a / 2), 0.0, math.sin(a / 2), 0.0])
            c = torch.tensor([-cz * math.sin(a), 0.0, cz - cz * math.cos(a)])
            qs.append(q)
            ts.append(c)
        # the MLP and the camera poses live in a second, small flat bucket: their gradients are written by the HIP
        # backward kernels (no AccumulateGrad), summed by one small all-reduce, and stepped by the SAME Adam launch
        # as the Gaussians (reference groups: deform network 0.0016, pose rotation 1e-5, pose translation 1e-6)
        sp = bind_module_to_flat(self.net, 0.0016, dev, {"cam_q": ((num_frames, 4), 1e-5),
                                                          "cam_t": ((num_frames, 3), 1e-6)})
        with torch.no_grad():
            sp["cam_q"].copy_(torch.stack(qs))
            sp["cam_t"].copy_(torch.stack(ts))
        self.sp = sp
        self.cam_q, self.cam_t = sp["cam_q"], sp["cam_t"]
        self.net.grad_sinks = [sp[k].grad for k in _MLP_SINK_ORDER]
        self.pose_sinks = {"q": sp["cam_q"].grad, "t": sp["cam_t"].grad}
        self.proj_t = scene["projmatrix"].to(dev).contiguous()   # already P^T (glm storage)
        self.bg = torch.zeros(3, device=dev)
        # frame f: the T birth-time embedding rows followed by the frame's own row -> one MLP pass, no cat per step
        emb = self.time_batch_embeddings
        self.emb_rows = torch.cat([emb.unsqueeze(0).expand(num_frames, -1, -1), self.frame_embeddings.unsqueeze(1)],
                                  dim=1).contiguous()                       # [T, T+1, 53]
        self.m2 = torch.zeros(P, 3, device=dev, requires_grad=True)        # means2D: values unused, gradient sink
        self.sync = BucketedAllReduce(self.fp, [self.sp.flat_grad])
        self.gt = {}
        self.gt_depth = {}
        self.stats = None        # DensifyStats once track_densification() is called
        self.raster_state = RasterState()    # this scene's frame-to-frame rasterizer memory (capacity / binning hints)
        # full_losses: the whole loss set of the reference's dynamic sub-step (config 5,
        # configs/train/train_kubric_mrig.yaml:186-232): photometric + motion L1 / sparsity / basis regularisers +
        # global and local Pearson depth + rigidity every 5th step.  Several losses then feed the same parameters, so
        # only the SH features keep an overwriting gradient sink; the other segments are zeroed and accumulated.
        self.full_losses = full_losses
        # optimizer-in-backward for the SH features (see render()); train_step switches it on for the single-GPU
        # step only -- a gradient exchange needs dL/dshs itself
        self.fuse_sh_adam = False
        self._graph_inputs = None    # GraphedStep: (time-embedding rows, ground truth, RdgStepScalars, depth ground truth) at fixed addresses
        self._grad_one = None
        self._rows_adam_hook = None  # set for one backward by train_step: _step_rows_early (the rows' Adam next to the MLP backward)
        self._rows_stepped = False
        self._side_stream = None
        self._pose_fork = None       # (stream, dL/dviewmatrix buffer) of the eager step's pose-chain branch
        self._plain_full = False     # full_losses on a step without rigidity: the photometric step's fused kernels
        if full_losses:
            from .depth_losses import GlobalPearsonDepthLoss, LocalPearsonDepthLoss
            from .motion_losses import MotionBasisRegularizaiton, MotionL1Loss, MotionSparsityLoss
            from .rigidity import RigidityLoss
            self.loss_terms = {"motion_l1": (0.01, MotionL1Loss()), "motion_sparsity": (0.002, MotionSparsityLoss()),
                               "motion_basis_reg": (0.1, MotionBasisRegularizaiton(transl_degree=0))}
            self.depth_terms = [(0.05, GlobalPearsonDepthLoss("all")), (0.15, LocalPearsonDepthLoss(128, 0.5, "all"))]
            self.rigidity = (0.5, 5, RigidityLoss(mode=["distance_preserving", "surface"], K=8,
                                                  device_sampling=True))

    # ---- pieces of the step ------------------------------------------------------------------------------------
    def settings(self) -> GaussianRasterizationSettings:
        return GaussianRasterizationSettings(self.H, self.W, self.tanfovx, self.tanfovy, self.bg, 1.0, self.proj_t,
                                             self.sh_degree, False, False, True, True)

    def gaussians_at(self, frame: int):
        """DynRoDyGS.get_gaussian_deformation + activations for the frame's time."""
        fp, net = self.fp, self.net
        # ONE pass of the MLP over the T birth-time rows + the frame's own time (row T)
        gi = self._graph_inputs
        allb = net.motion_basis(self.emb_rows[frame] if gi is None else gi[0])   # [T+1,16,7]: table rows, then B(t)
        self._last_allb = allb
        if (not self.full_losses or self._plain_full) and dynamic_getter_supported(allb.shape[1], allb.shape[0] - 1):
            # deformation + activations in ONE kernel each way; all five parameter gradients go straight to the bucket
            sinks = {k: fp[k].grad for k in ("xyz", "scaling", "rotation", "opacity")}
            sinks["coeff"] = fp["motion_coeff"].grad
            if self._rows_adam_hook is not None:
                sinks["after_rows"] = self._rows_adam_hook
            xyz, scaling, rot, opacity = dynamic_gaussians(fp["xyz"], fp["scaling"], fp["rotation"], fp["opacity"],
                                                           fp["motion_coeff"], self.time_ind, allb,
                                                           self.spatial_lr_scale, grad_sinks=sinks)
            return xyz, opacity, scaling, rot, fp["features"]
        coeff_sink = None if self.full_losses else {"coeff": fp["motion_coeff"].grad}
        dxyz, drot = gaussian_deformation_packed(fp["motion_coeff"], self.time_ind, allb, self.spatial_lr_scale,
                                                 grad_sinks=coeff_sink)
        # activations + deformation add + feature concat: 2 HIP launches; the parameter gradients are written by
        # the backward kernel straight into the flat gradient bucket (no AccumulateGrad copies)
        sinks = None if self.full_losses else {k: fp[k].grad for k in ("xyz", "scaling", "rotation", "opacity")}
        xyz, scaling, rot, opacity, _ = activate_gaussians(fp["xyz"], dxyz, fp["scaling"], fp["rotation"], drot,
                                                           fp["opacity"], None, None, grad_sinks=sinks)
        self._last = (dxyz, allb)
        return xyz, opacity, scaling, rot, fp["features"]

    def render(self, frame: int):
        xyz, opacity, scaling, rot, feats = self.gaussians_at(frame)
        gi = self._graph_inputs
        vm = pose_view_matrix(self.cam_q, self.cam_t, frame, grad_sinks=self.pose_sinks,
                              step_scalars=None if gi is None else gi[2])
        m2 = self.m2
        m2.grad = None
        sinks = {"shs": self.fp["features"].grad,
                 # frame-DP: the SH gradient goes on the wire while the rest of backward is still running
                 "on_shs_ready": lambda: self.sync.ready("features")}
        if self.fuse_sh_adam and torch.is_grad_enabled():
            # single GPU, photometric step: nothing else needs dL/dshs, so the per-Gaussian backward kernel applies the
            # Adam update of the SH features itself (64 % of all optimiser bytes never make the round trip)
            o, n = self.fp.offsets["features"]
            row, head, lr_tail = self.row_lr["features"]
            sinks = {"shs_adam": {"param": self.fp["features"], "exp_avg": self.fp.exp_avg[o:o + n],
                                  "exp_avg_sq": self.fp.exp_avg_sq[o:o + n], "head_len": head,
                                  "lr_head": self.fp.lr["features"], "lr_tail": lr_tail, "betas": (0.9, 0.999),
                                  "eps": 1e-15, "step": lambda: self.fp.step_count + 1,
                                  "step_scalars": None if gi is None else gi[2]}}
        if self.stats is not None and torch.is_grad_enabled():
            # densification statistics of the iteration (rodygs.py:316-341): updated by the per-Gaussian backward kernel
            sinks["densify"] = self.stats.sink()
        out = GaussianRasterizer(self.settings(), state=self.raster_state)(
            means3D=xyz, means2D=m2, shs=feats, opacities=opacity, scales=scaling, rotations=rot, viewmatrix=vm,
            grad_sinks=sinks)
        self._last_radii = out[4]
        return out, m2

    # ---- densification in the loop (rodygs.py:319-362) ----------------------------------------------------------
    def track_densification(self, headroom: float = 1.3) -> None:
        """Start keeping the densification statistics.  The cloud is moved into flat buffers with ``headroom`` (x its size) and
        a spare set of the same size is allocated NOW: every densification from here on gathers the rows from one set into the
        other and allocates nothing while the cloud stays below the headroom (``densify.FlatPool``)."""
        from .densify import DensifyStats, FlatPool, rebuild_flat_params
        from .dp import FlatStorage
        self.stats = DensifyStats.zeros(self.P, self.device)
        if self.fp.flat.is_cuda and self.fp.storage is None and headroom > 1.0:
            self._fp_pool = FlatPool(headroom)
            with torch.no_grad():
                every = torch.arange(self.P, device=self.device)
                self.fp = rebuild_flat_params(self.fp, every, torch.ones(self.P, dtype=torch.bool, device=self.device),
                                              self._fp_pool)
            self._fp_pool.spare = FlatStorage(self.fp.storage.capacity, self.device)
            self.sync = BucketedAllReduce(self.fp, [self.sp.flat_grad])

    # ---- fixed capacity: a cloud whose buffers never move or change size (graph replay across densifications) -------------
    def fix_capacity(self, headroom: float = 1.15) -> None:
        """Move the cloud into buffers of ``ceil(P * headroom)`` rows (rounded up to 256) and keep them for good: the rows beyond
        the live Gaussians are DEAD (``densify.dead_row_template``: parked behind every camera, zero motion coefficients and
        moments -- culled by the near plane, no gradient, no statistics, a zero Adam update), and ``densify_inplace`` turns dead
        rows into clones / split children and pruned Gaussians into dead rows.  ``self.P`` becomes the CAPACITY -- what every launch
        dimension and every tensor shape of the step is made of --, ``self.P_live`` counts the Gaussians.  Addresses, shapes and
        launch dimensions then survive a densification, so a ``GraphedStep`` captured once keeps replaying across it.  The price:
        the per-Gaussian kernels also walk the dead rows (``headroom`` - 1 of their time while the cloud is young)."""
        from .densify import DensifyStats, dead_row_template
        from .deform import invalidate_birth_order_cache
        if self._graph_inputs is not None:
            raise RuntimeError("fix_capacity(): close the open GraphedStep first")
        dev = self.device
        old = self.fp
        # called again on a cloud that already has dead rows (its capacity ran out): the live rows move up, in order
        live = None if getattr(self, "dead", None) is None else (~self.dead).nonzero().squeeze(1)
        P = self.P if live is None else int(live.numel())
        cap = ((int(math.ceil(P * float(headroom))) + 255) // 256) * 256
        spec = {k: ((cap,) + tuple(old.shapes[k][1:]), old.lr[k]) for k in old.names}
        fp = FlatParams(spec, dev)
        with torch.no_grad():
            for k in old.names:
                oo, on = old.offsets[k]
                no, _ = fp.offsets[k]
                rl = on // old.shapes[k][0]
                for dst, src in ((fp.flat, old.flat), (fp.exp_avg, old.exp_avg), (fp.exp_avg_sq, old.exp_avg_sq)):
                    rows = src[oo:oo + on].view(old.shapes[k][0], rl)
                    dst[no:no + P * rl].view(P, rl).copy_(rows if live is None else rows.index_select(0, live))
        fp.step_count = old.step_count
        self.dead = torch.zeros(cap, dtype=torch.bool, device=dev)
        self.dead[P:] = True
        dead_row_template(fp, torch.arange(P, cap, device=dev))
        ti = torch.zeros(cap, dtype=self.time_ind.dtype, device=dev)
        ti[:P] = self.time_ind if live is None else self.time_ind.index_select(0, live)
        old_key = (self.P, self.H, self.W)
        self.fp, self.time_ind, self.P, self.P_live = fp, ti, cap, P
        # the rasterizer's frame-to-frame memory is keyed by the row count: carried over (same Gaussians, a few more dead rows)
        st = self.raster_state
        with st.lock:
            for tbl in (st.capacity_hint, st.d_high, st.bin_hint, st.split_hint):
                if old_key in tbl and old_key != (cap, self.H, self.W):
                    tbl[(cap, self.H, self.W)] = tbl.pop(old_key)
        invalidate_birth_order_cache()
        self.m2 = torch.zeros(cap, 3, device=dev, requires_grad=True)
        self.sync = BucketedAllReduce(self.fp, [self.sp.flat_grad])
        self.stats = DensifyStats.zeros(cap, dev)
        self._fp_pool = None

    def live_gradient_quantile(self, q: float) -> torch.Tensor:
        """The ``q`` quantile of the mean screen-space gradient over the LIVE Gaussians (a 0-dim device tensor: no read-back)."""
        g_mean = (self.stats.xyz_gradient_accum / self.stats.denom.clamp_min(1)).reshape(-1)
        if getattr(self, "dead", None) is None:
            return torch.quantile(g_mean, q)
        # dead rows carry 0 and the live count is known on the host: the q quantile of the live values is the
        # (dead + q * live) / rows quantile of all of them up to the position of the zeros, which sort first
        rows, live = self.P, self.P_live
        return torch.quantile(g_mean, min(1.0, ((rows - live) + q * live) / rows))

    def densify_inplace(self, max_grad=0.0002, min_opacity: float = 0.005, extent: Optional[float] = None, max_screen_size=None,
                        percent_dense: float = 0.01, z: Optional[torch.Tensor] = None):
        """``densify()`` on a cloud of fixed capacity (``fix_capacity``): same decisions, row surgery in place; may be called
        while a ``GraphedStep`` of this scene is open -- nothing the graph refers to moves.  Returns the info dict, or None when
        the dead rows do not suffice (the caller closes its graph, calls ``fix_capacity`` again and re-captures)."""
        from .deform import refresh_birth_order_inplace
        from .densify import allreduce_stats_, densify_and_prune_inplace
        if getattr(self, "dead", None) is None:
            raise RuntimeError("call fix_capacity() first")
        allreduce_stats_(self.stats)
        info = densify_and_prune_inplace(self.fp, self.stats, {"time_ind": self.time_ind}, self.dead, max_grad, min_opacity,
                                         extent if extent is not None else self.spatial_lr_scale, max_screen_size, percent_dense,
                                         z=z)
        if info is None:
            return None
        self.P_live = info["live"]
        refresh_birth_order_inplace(self.time_ind, self.T)      # the sorted order the dB reduction reads: same tensors, new content
        info["P"] = self.P_live
        return info

    def densify(self, max_grad=0.0002, min_opacity: float = 0.005, extent: Optional[float] = None,
                max_screen_size=None, percent_dense: float = 0.01, z: Optional[torch.Tensor] = None,
                decisions=None, want_decisions: bool = True, timings: Optional[dict] = None) -> dict:
        """densify_and_prune over the flat bucket, then re-point everything that referred to the old buffers
        (gradient sinks live in the new bucket, birth indices follow the Gaussians, exchange object rebuilt).
        The rasterizer's frame-to-frame memory is carried across the row surgery: the capacity of the binning workspace
        scaled by P' / P (+ 25 %: clones and split children are drawn from the Gaussians with the largest screen-space gradient,
        the instance count grows faster than the cloud -- a 2 500-step loop outgrew + 10 % after 18 densifications), the binning /
        split-compositing hints as they stand -- the first forward of the new cloud then needs no read-back of its instance
        count (a capacity that turns out too small is found by the usual overflow check).  ``timings``: per-phase wall times (ms) of this call, synchronising at every phase boundary (diagnosis)."""
        from .densify import FlatPool, _Phase, allreduce_stats_, densify_and_prune
        if self.stats is None:
            raise RuntimeError("call track_densification() first")
        if self._graph_inputs is not None:
            # the buffers this call hands back to the pool become the gather target of the NEXT densification: a captured graph
            # that still replays into them would train on -- and overwrite -- live parameters of a later cloud, silently
            raise RuntimeError("DynamicScene.densify(): a GraphedStep of this scene is still open; close() it first (its "
                               "graph holds the addresses of the buffers that are about to be recycled)")
        allreduce_stats_(self.stats)
        P_old = self.P
        if getattr(self, "_fp_pool", None) is None:
            self._fp_pool = FlatPool()
        old_fp = self.fp
        res = densify_and_prune(self.fp, self.stats, {"time_ind": self.time_ind}, max_grad, min_opacity,
                                extent if extent is not None else self.spatial_lr_scale, max_screen_size, percent_dense,
                                z=z, decisions=decisions, spatial_order=self.spatial_order, want_decisions=want_decisions,
                                timings=timings, pool=self._fp_pool)
        ph = _Phase(timings, self.device)
        self.fp, self.stats, self.time_ind = res.fp, res.stats, res.per_point["time_ind"].contiguous()
        self._fp_pool.give_back(old_fp)          # (stream-ordered: the gathers that read it are queued before any re-use)
        # the old bucket's storage is the pool's again: whoever still holds `old_fp` must not find live-looking views in it
        old_fp.flat = old_fp.flat_grad = old_fp.exp_avg = old_fp.exp_avg_sq = None
        old_fp.params = {}
        del old_fp
        self.P = self.fp.shapes["xyz"][0]
        self.m2 = torch.zeros(self.P, 3, device=self.device, requires_grad=True)
        self.sync = BucketedAllReduce(self.fp, [self.sp.flat_grad])
        st = self.raster_state
        old, new = (P_old, self.H, self.W), (self.P, self.H, self.W)
        with st.lock:
            if old in st.capacity_hint and new != old:
                st.capacity_hint[new] = max(int(st.capacity_hint.get(new, 0)),
                                            int(st.capacity_hint[old] * (self.P / max(P_old, 1)) * 1.25) + 4096)
                st.d_high[new] = max(int(st.d_high.get(new, 0)), st.capacity_hint[new],
                                     int(st.d_high.get(old, 0) * (self.P / max(P_old, 1)) * 1.25))
                for table in (st.bin_hint, st.split_hint):
                    if old in table:
                        table[new] = table[old]
                # the old cloud's entries are dead weight from here on
                for table in (st.capacity_hint, st.d_high, st.bin_hint, st.split_hint):
                    table.pop(old, None)
        ph.mark("re_hint")
        return {"P": self.P, "cloned": res.n_clone, "split": res.n_split, "pruned": res.n_pruned,
                "decisions": res.decisions}

    def make_ground_truth(self, target_scene: dict, frames):
        """GT images = HIP render of a different-seed static cloud from each frame's camera."""
        dev = self.device
        with torch.no_grad():
            for f in frames:
                vm = pose_view_matrix(self.cam_q, self.cam_t, int(f))
                z = torch.zeros_like(target_scene["means3D"])
                out = GaussianRasterizer(self.settings(), state=self.raster_state)(
                    means3D=target_scene["means3D"].to(dev), means2D=z.to(dev), shs=target_scene["shs"].to(dev),
                    opacities=target_scene["opacities"].to(dev), scales=target_scene["scales"].to(dev),
                    rotations=target_scene["rotations"].to(dev), viewmatrix=vm)
                self.gt[int(f)] = out[0].clamp(0, 1).clone()
                self.gt_depth[int(f)] = out[1].clone()

    def _full_loss(self, step: int, frame: int) -> torch.Tensor:
        fp = self.fp
        o0 = fp.offsets["features"][0]
        o1, n1 = fp.offsets["features"]
        fp.flat_grad[:o0].zero_()                                   # accumulated segments (features keep their sink)
        fp.flat_grad[o1 + n1:].zero_()
        out, _ = self.render(frame)
        dxyz, allb = self._last
        scene = self

        class _Model:                                               # what the reference's loss modules read
            _xyz, _motion_coeff = fp["xyz"], fp["motion_coeff"]
            _features_dc = fp["features"].detach()[:, :1]
            unique_times = list(range(scene.T))

            @staticmethod
            def get_total_motion_table():
                return allb[:-1]

            @staticmethod
            def get_motion_for_times(timesteps, time_indices=None):
                return allb[:-1][time_indices.to(allb.device)]

        loss = fused_photometric_loss(out[0], self.gt[frame], 0.2)
        # motion L1 + sparsity: one HIP pass each way, gradient added straight into the flat bucket; the basis
        # regulariser (on the small motion table) stays the host mirror
        from .motion_losses import fused_motion_l1_sparsity
        loss = loss + fused_motion_l1_sparsity(fp["motion_coeff"], self.loss_terms["motion_l1"][0],
                                               self.loss_terms["motion_sparsity"][0], grad_sink=fp["motion_coeff"].grad)
        w, mod = self.loss_terms["motion_basis_reg"]
        loss = loss + w * mod(_Model)
        for w, mod in self.depth_terms:
            loss = loss + w * mod(out[1], self.gt_depth[frame])
        w, freq, mod = self.rigidity
        if step % freq == 0:
            loss = loss + w * mod(_Model, dxyz)
        return loss

    def _full_loss_plain(self, frame: int, fuse: bool):
        """The config-5 loss set on a step WITHOUT rigidity (4 of 5): nothing then needs the deformation as a tensor,
        so the step runs on the photometric step's fused kernels (deformation + activations in one kernel each way,
        gradients overwritten in the flat bucket, SH Adam in backward) with the depth terms on the rendered depth and
        the basis regulariser on the motion table joining the same backward pass.  Returns (main loss, callable that
        adds the per-Gaussian motion regularisers AFTER the main backward: their gradient is accumulated on top of
        the kernels' overwriting writes, so it must come second)."""
        from .motion_losses import fused_motion_l1_sparsity
        fp = self.fp
        self._plain_full, self.fuse_sh_adam = True, fuse
        try:
            out, _ = self.render(frame)
        finally:
            self._plain_full, self.fuse_sh_adam = False, False
        allb = self._last_allb

        class _Model:
            @staticmethod
            def get_total_motion_table():
                return allb[:-1]

        gi = self._graph_inputs
        loss = fused_photometric_loss(out[0], self.gt[frame] if gi is None else gi[1], 0.2)
        w, mod = self.loss_terms["motion_basis_reg"]
        loss = loss + w * mod(_Model)
        gt_depth = self.gt_depth[frame] if gi is None else gi[3]
        for w, mod in self.depth_terms:
            loss = loss + w * mod(out[1], gt_depth)

        def after():
            reg = fused_motion_l1_sparsity(fp["motion_coeff"], self.loss_terms["motion_l1"][0],
                                           self.loss_terms["motion_sparsity"][0], grad_sink=fp["motion_coeff"].grad)
            reg.backward()
            return reg.detach()
        return loss, after

    def _step_rows_early(self) -> None:
        """Called from inside backward (deform._DynamicGetter.backward, ``after_rows``) once every per-Gaussian gradient is in the
        bucket: the Adam launch of the per-Gaussian rows (0.13 ms at 1 M, HBM-bound) goes to a second stream and runs NEXT TO the
        rest of backward -- the MLP's products, a chain of small latency-bound launches (0.045 ms) that touches neither the rows
        nor their gradients.  train_step() steps the MLP + pose bucket afterwards and joins the streams.  Same launches, same
        arithmetic, same bits as the one-launch form (RDG_EARLY_ROWS_ADAM=0)."""
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = _low_priority_stream(self.device)
        side = self._side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fused_adam_(self.fp, names=[k for k in self.fp.names if k != "features"])
        self._rows_stepped = True

    def train_step(self, step: int, rank: int = 0, world: int = 1, perm=None) -> torch.Tensor:
        perm = perm if perm is not None else list(self.gt.keys())
        frame = frame_for(step, rank, world, perm)
        after = None
        self._rows_adam_hook, self._rows_stepped = None, False
        self.raster_state.pose_fork_eager = None
        if self.full_losses and step % self.rigidity[1] != 0 and _PLAIN_FULL_FAST:
            fuse = world == 1 and _FUSE_SH_ADAM
            loss, after = self._full_loss_plain(frame, fuse)
        elif self.full_losses:
            loss = self._full_loss(step, frame)
        else:
            # every segment of both flat gradient buckets is OVERWRITTEN by a backward kernel: nothing to zero
            self.fuse_sh_adam = fuse = world == 1 and _FUSE_SH_ADAM
            eager = self._graph_inputs is None and not self.raster_state.graph_capture
            if fuse and _EARLY_ROWS_ADAM and eager:
                self._rows_adam_hook = self._step_rows_early
            if fuse and _EAGER_POSE_FORK and eager and self.pose_sinks is not None:
                # the pose-gradient chain (two reductions + the pose backward, three small dependent launches between the
                # per-Gaussian backward and the getter's) as a branch on a second stream; joined before the pose Adam
                if self._pose_fork is None:
                    self._pose_fork = (torch.cuda.Stream(device=self.device),
                                       torch.empty(4, 4, dtype=torch.float32, device=self.device))
                self.raster_state.pose_fork_eager = self._pose_fork
                self.pose_sinks["aux"] = self.raster_state
            try:
                out, _ = self.render(frame)
            finally:
                self.fuse_sh_adam = False
            gi = self._graph_inputs
            loss = fused_photometric_loss(out[0], self.gt[frame] if gi is None else gi[1], 0.2)
        # the seed gradient is a resident 1.0 (autograd would otherwise launch a fill kernel for ones_like(loss))
        if self._grad_one is None or self._grad_one.device != loss.device:
            self._grad_one = torch.ones((), dtype=torch.float32, device=loss.device)
        loss.backward(self._grad_one)
        self._rows_adam_hook = None
        st_ = self.raster_state
        if st_.pose_fork_eager is not None:
            torch.cuda.current_stream(self.device).wait_stream(st_.pose_fork_eager[0])
            st_.pose_fork_eager = None
        if st_.graph_capture and st_.aux_stream is not None:
            # the pose-gradient chain ran as a branch of the graph (RasterState.aux_stream): join before the optimiser reads it
            torch.cuda.current_stream(self.device).wait_stream(st_.aux_stream)
        if after is not None:
            loss = loss.detach() + after()
        # (densification statistics, when tracked: updated inside backward by the per-Gaussian kernel, see render())
        if world > 1:
            # overlapped exchange: pieces arrive in issue order; Adam steps each piece while the next one is in flight
            self.sync.finish()
            first = True
            for names in self.sync.drain():
                if names is None:                       # the small MLP + pose bucket
                    fused_adam_(self.fp, names=(), extra=(self.sp,), advance=first)
                else:
                    fused_adam_(self.fp, row_lr=self.row_lr, names=names, advance=first)
                first = False
        elif self._rows_stepped:
            # the per-Gaussian rows are being stepped on the second stream (_step_rows_early): the small MLP + pose bucket here,
            # then the two streams meet -- the next forward reads both
            self._rows_stepped = False
            fused_adam_(self.fp, names=(), extra=(self.sp,), advance=False)
            torch.cuda.current_stream(self.device).wait_stream(self._side_stream)
        elif (not self.full_losses or after is not None) and fuse:
            # the SH features were stepped inside backward; everything else in the usual single launch
            fused_adam_(self.fp, names=[k for k in self.fp.names if k != "features"], extra=(self.sp,),
                        step_scalars=None if self._graph_inputs is None else self._graph_inputs[2])
        else:
            fused_adam_(self.fp, row_lr=self.row_lr, extra=(self.sp,),
                        step_scalars=None if self._graph_inputs is None else self._graph_inputs[2])
        return loss.detach()


class GraphedStep:
    """hipGraph replay of ``DynamicScene.train_step`` (single GPU; the photometric step, with or without densification
    statistics, and the config-5 loss set on its steps without rigidity): ONE graph launch per step instead of ~50 kernel
    launches, ~30 allocations and ten autograd nodes driven from Python.  At the size of the reference's real clouds
    (<= 120 k points per cloud, /root/reference/configs/train/train_kubric_mrig.yaml:42,102) the eager step is host-bound;
    the graph runs at the speed of its kernels.

    What changes from step to step lives in device memory at fixed addresses, refreshed before each replay: the frame's
    time-embedding rows and ground truth (device copies), and a 128-byte ``RdgStepScalars`` -- Adam's two bias corrections
    (computed on the host exactly as the eager path does), the frame index for the camera-pose kernels and the LEARNING
    RATES of every parameter group (``fp.lr`` / ``sp.lr`` as they stand when the step is staged, or ``step(lr_override=)``:
    the reference re-sets the xyz rate every iteration, /root/reference/src/trainer/rodygs_static.py:143-149) -- one small
    H2D copy out of a ring of pinned slots.  Same kernels, same arithmetic, same bits as the eager step.
    The instance count of every replayed frame is folded into a STICKY device maximum (RdgRasterSettings
    .num_rendered_max); ``check()`` reads it (one sync) and raises ``RasterizerCapacityOverflow`` if ANY frame since the
    last check outgrew the capacity the graph was captured with -- such a frame was rendered empty and its Adam step ran
    on zero gradients (re-build the GraphedStep then; also after a densification, which changes every buffer)."""

    RING = 256

    # pinned staging rings are expensive to allocate (a page-locked allocation: ~1 ms) and carry nothing across graphs: a
    # trainer that re-captures after every densification gets the previous one back (keyed by device)
    _RING_CACHE: dict = {}

    def __init__(self, ds: "DynamicScene", perm, warmup: int = 2, first_step: int = 0, timings: Optional[dict] = None,
                 pool=None):
        """``pool``: a graph memory-pool handle (``GraphedStep.pool()`` of an earlier, closed graph of this scene) the capture
        allocates from instead of a fresh private pool -- a re-capture after a densification then re-uses the blocks of the
        graph it replaces instead of asking the driver for a new gigabyte.  ``timings``: phase wall times (ms) of this
        constructor (bench.py --loop --graph)."""
        import time as _time
        t_last = [_time.perf_counter()]

        def mark(name):
            if timings is not None:
                torch.cuda.synchronize(ds.device)
                now = _time.perf_counter()
                timings[name] = timings.get(name, 0.0) + (now - t_last[0]) * 1e3
                t_last[0] = now
        if ds.full_losses and not _PLAIN_FULL_FAST:
            raise NotImplementedError("GraphedStep needs the fused plain full-loss step (RDG_PLAIN_FULL_FAST=1)")
        self.ds, self.perm = ds, list(perm)
        dev = ds.device
        W = _lib.STEP_SCALARS_FLOATS
        self.scal = torch.zeros(W, dtype=torch.float32, device=dev)                    # RdgStepScalars
        ring = GraphedStep._RING_CACHE.pop(str(dev), None)
        self.ring = ring if ring is not None else torch.zeros(self.RING, W, dtype=torch.float32).pin_memory()
        self.ring_i32 = self.ring.view(torch.int32)
        self.emb_in = torch.empty_like(ds.emb_rows[0])
        self.gt_in = torch.empty_like(ds.gt[self.perm[0]])
        self.gt_depth_in = torch.empty_like(ds.gt_depth[self.perm[0]]) if ds.full_losses else None
        self._slot, self._fence = 0, []
        ds._graph_inputs = (self.emb_in, self.gt_in, self.scal, self.gt_depth_in)
        st = ds.raster_state
        st.nren_max = torch.zeros(1, dtype=torch.int32, device=dev)       # sticky maximum of D over the replays
        # forwards of another (P, H, W) through the same state (an evaluation render, the eager rigidity step of another
        # size) must not fold their instance count into this graph's record: the rasterizer binds it for this key only
        st.nren_max_key = (ds.P, ds.H, ds.W)
        ds.fp.__dict__.pop("_adam_graph_layout", None)       # a layout left by an earlier GraphedStep on this bucket is stale
        step = first_step
        if ds.full_losses and step % ds.rigidity[1] == 0:
            step += 1                            # a rigidity step is not part of the graph (see step())
        for _ in range(max(1, warmup)):          # eager, on the same staged inputs: hints, caches, lazy initialisation
            if ds.full_losses and step % ds.rigidity[1] == 0:
                step += 1
            self._stage(step)
            ds.train_step(step, 0, 1, self.perm)
            step += 1
        if ds.full_losses and step % ds.rigidity[1] == 0:
            step += 1
        self.next_step = step
        # RDG_GRAPH_BRANCHES=1 (opt-in, A/B switch): the pose-gradient chain (two reduction launches + the pose backward) as an
        # independent branch of the captured step next to the deformation / MLP backward -- forked inside the library onto a second
        # stream (RdgRasterSettings.aux_stream), joined before the optimiser.  Built in round 5 and measured as a LOSS: 0.626 ms
        # against 0.597 ms per replayed step at 100 k points (profiles/r05_experiments.txt 6) -- the fork / join nodes of the
        # hipGraph cost more than the 14 us the branch hides.  Off by default; not for the deterministic mode's two-pass backward.
        if os.environ.get("RDG_GRAPH_BRANCHES", "0") == "1" and not st.mode("deterministic"):
            _lib.check(_lib.lib().rdg_pose_fork_prepare(), "rdg_pose_fork_prepare")
            st.aux_stream = torch.cuda.Stream(device=dev)
            st.pose_grad = torch.empty(4, 4, dtype=torch.float32, device=dev)
            ds.pose_sinks["aux"] = st
        torch.cuda.synchronize(dev)
        mark("setup_and_eager_warmup")
        counts = (ds.fp.step_count, ds.sp.step_count)
        # the captured capacity is fixed for every replay: size it with the recent MAXIMUM of the instance count (and what a
        # densification carried over), not with the one frame the warm-up step happened to render -- frames of a run differ
        # by 25 % and more (a 2 500-step loop with a re-capture every 100 steps outgrew a capacity taken from the last frame)
        with st.lock:
            key_ = (ds.P, ds.H, ds.W)
            st.capacity_hint[key_] = max(int(st.capacity_hint.get(key_, 0)), int(st.d_high.get(key_, 0)))
            if getattr(ds, "dead", None) is not None:
                # fixed capacity: this graph is meant to outlive densifications -- the instance count can grow with the cloud
                # until every dead row is a Gaussian (and a little faster: new Gaussians sit where the gradient is large)
                st.capacity_hint[key_] = int(st.capacity_hint[key_] * 1.1 * ds.P / max(int(ds.P_live), 1))
        self.graph = torch.cuda.CUDAGraph()
        st.graph_capture = True
        try:
            self._stage(step)
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(self.graph, pool=pool):
                self.loss = ds.train_step(step, 0, 1, self.perm)       # recorded, not executed
        finally:
            st.graph_capture = False
            st.keep_alive.clear()
        mark("capture_and_instantiate")
        ds.fp.step_count, ds.sp.step_count = counts                    # capture advanced the host counters only
        self._nren, self._key, self._cap = st.last_nren
        st.nren_max.zero_()

    def _stage(self, step: int, lr_override=None) -> int:
        ds = self.ds
        frame = frame_for(step, 0, 1, self.perm)
        k = ds.fp.step_count + 1
        i = self._slot
        self._slot = (i + 1) % self.RING
        if i % (self.RING // 2) == 0:
            # the half of the ring about to be rewritten was copied from more than RING / 2 steps ago: wait for the
            # event recorded half a ring ago (after those copies in stream order), then leave one for the next half
            if self._fence:
                self._fence.pop(0).synchronize()
            ev = torch.cuda.Event()
            ev.record()
            self._fence.append(ev)
        row = self.ring[i]
        # float(1 / bc1) and float(sqrt(bc2)) from doubles: what rdg_adam_step_multi computes on the host
        row[0] = 1.0 / (1.0 - 0.9 ** k)
        row[1] = math.sqrt(1.0 - 0.999 ** k)
        self.ring_i32[i, 2] = frame
        # learning rates: the segment table of the (captured) Adam launch, in its order; 0 until the first eager step of
        # this object has recorded the layout (the by-value rates of that very step are then the current ones anyway)
        rec = ds.fp.__dict__.get("_adam_graph_layout")
        self.ring_i32[i, 3] = 0
        if rec is not None:
            layout, lr_scale = rec
            for j, (f, names, row_name) in enumerate(layout):
                rates = {float(lr_override[n]) if (f is ds.fp and lr_override and n in lr_override) else f.lr[n]
                         for n in names}
                if len(rates) != 1:
                    raise RuntimeError(f"GraphedStep: the parameter groups {names} were merged into one Adam segment at "
                                       "capture and now have different learning rates; re-build the GraphedStep")
                lr = rates.pop()
                row[4 + j] = lr * lr_scale
                row[16 + j] = (ds.row_lr[row_name][2] if row_name is not None else lr) * lr_scale
            lr_f = (float(lr_override["features"]) if lr_override and "features" in lr_override
                    else ds.fp.lr["features"])
            row[28] = lr_f
            row[29] = ds.row_lr["features"][2]
            self.ring_i32[i, 3] = 1
        self.scal.copy_(row, non_blocking=True)
        self.emb_in.copy_(ds.emb_rows[frame])
        self.gt_in.copy_(ds.gt[frame])
        if self.gt_depth_in is not None:
            self.gt_depth_in.copy_(ds.gt_depth[frame])
        return frame

    def step(self, lr_override=None) -> torch.Tensor:
        """Stage the step's inputs, replay the graph; returns the (device) loss of the step.  ``lr_override``: {group
        name: lr} for this step only, as ``fused_adam_``; groups not named take ``fp.lr`` as it stands now.  With the
        config-5 loss set, every ``freq``-th step carries the rigidity loss (host-side sampling, K-NN): that step runs
        eagerly on the same staged inputs, the four steps between them replay the graph."""
        ds = self.ds
        if ds.full_losses and self.next_step % ds.rigidity[1] == 0:
            if lr_override:
                raise NotImplementedError("lr_override on an eager rigidity step")
            gi, ds._graph_inputs = ds._graph_inputs, None
            try:
                loss = ds.train_step(self.next_step, 0, 1, self.perm)
            finally:
                ds._graph_inputs = gi
            self.next_step += 1
            return loss
        self._stage(self.next_step, lr_override)
        self.graph.replay()
        self.next_step += 1
        ds.fp.step_count += 1
        ds.sp.step_count = ds.fp.step_count
        return self.loss

    def check(self) -> int:
        """Largest instance count of ANY frame replayed since the last check (synchronises); raises if it exceeded the
        captured capacity -- that frame was rendered empty."""
        from . import rasterizer
        st = self.ds.raster_state
        n = int(st.nren_max.item())
        st.nren_max.zero_()
        if n >= rasterizer._INSTANCE_LIMIT:
            raise RuntimeError(rasterizer._too_many(self._key))
        with st.lock:
            # the replayed maximum feeds BOTH records the next capture sizes itself with: the hint (a slowly decaying maximum,
            # the rule of every other path) and d_high -- the new GraphedStep's eager warm-up forward takes its capacity from
            # the hint, and without d_high having seen the overflowing count the re-captured graph could come out too small
            # again whenever no densification sits between check() and the re-capture
            st.capacity_hint[self._key] = max(n, int(st.capacity_hint.get(self._key, 0) * rasterizer.HINT_DECAY))
            st.note_instances(self._key, n)
        if n > self._cap:
            raise rasterizer.RasterizerCapacityOverflow(
                f"a replayed frame needed {n} instances, the graph was captured with {self._cap}: that frame was "
                f"rendered empty and its optimiser step ran on zero gradients; re-build the GraphedStep (the capacity "
                f"hint is now {n})")
        return n

    def pool(self):
        """The memory pool this graph's capture allocated from (hand it to the next GraphedStep of the scene)."""
        return self.graph.pool()

    def close(self) -> None:
        # wait for the staging copies still reading the ring, then leave it to the next graph
        torch.cuda.current_stream(self.ds.device).synchronize()
        GraphedStep._RING_CACHE[str(self.ds.device)] = self.ring
        self.ds._graph_inputs = None
        self.ds.pose_sinks.pop("aux", None)
        self.ds.raster_state.aux_stream = self.ds.raster_state.pose_grad = None
        self.ds.raster_state.nren_max = None
        self.ds.raster_state.nren_max_key = None
        self.ds.fp.__dict__.pop("_adam_graph_layout", None)


class ReferenceIteration:
    """The iteration a RoDyGS user runs (``RoDyGSTrainer.train``, /root/reference/src/trainer/rodygs.py:157-179): a STATIC
    sub-step, then a DYNAMIC sub-step, each of them ``train_iteration`` (:198-369) on the CONCATENATED cloud --

        MLP basis -> deformation of the dynamic Gaussians -> getters of both clouds written into the two row segments of one
        set of rasterizer inputs (``gs_properties``: the reference's five ``torch.cat`` copies are not made) -> rasterize
        (static ‖ dynamic) -> 0.8 L1 + 0.2 D-SSIM -> backward through BOTH clouds -> densification statistics of the
        sub-step's own slice (rows [0, Ps) / [Ps, Ps + Pd)) -> Adam step and zero_grad of the sub-step's OWN trainer only.

    **Stale-gradient semantics, kept** (SURVEY.md section 3.1): a sub-step's backward deposits gradients in the other cloud's
    parameters too, and only the cloud that steps clears its gradients afterwards -- so the dynamic step applies
    (gradient of the static sub-step's frame) + (gradient of its own frame), and the next static step the same with the roles
    swapped.  Here every parameter's ``.grad`` is a view of its flat bucket's gradient buffer and autograd ACCUMULATES into it
    (no overwriting sinks on this path); ``fp.zero_grad()`` after a trainer's Adam launch is its ``zero_grad(set_to_none)``.
    The static trainer owns the camera poses (``is_optimizable_cam``: pose gradients through the rasterizer's view matrix in the
    static sub-step only; the dynamic sub-step renders with the refined poses, gates off -- rodygs.py:170-178, 268-269).
    bench.py --iteration reference times it; it is not a trainer (no data module, schedules or densification calls)."""

    def __init__(self, static_scene: dict, dynamic_scene: dict, num_frames: int = 100, sh_degree: int = 3, device="cuda",
                 seed: int = 777, spatial_lr_scale: float = 5.0, orbit_deg: float = 15.0, spatial_order: bool = True):
        from .densify import DensifyStats
        from .layout import morton_order
        dev = torch.device(device)
        g = torch.Generator().manual_seed(seed + 1)
        self.device, self.T, self.sh_degree, self.spatial_lr_scale = dev, num_frames, sh_degree, spatial_lr_scale
        self.W, self.H = dynamic_scene["W"], dynamic_scene["H"]
        self.tanfovx, self.tanfovy = dynamic_scene["tanfovx"], dynamic_scene["tanfovy"]
        self.proj_t = dynamic_scene["projmatrix"].to(dev).contiguous()
        self.bg = torch.zeros(3, device=dev)

        def bucket(sc, dynamic):
            if spatial_order:
                perm = morton_order(sc["means3D"])
                sc = {k: (v[perm].contiguous() if k in ("means3D", "shs", "scales", "rotations", "opacities") else v)
                      for k, v in sc.items()}
            P, K = sc["means3D"].shape[0], sc["shs"].shape[1]
            spec = {"xyz": ((P, 3), 0.00016 * spatial_lr_scale), "f_dc": ((P, 1, 3), 0.0025),
                    "f_rest": ((P, K - 1, 3), 0.0025 / 20.0), "scaling": ((P, 3), 0.001), "rotation": ((P, 4), 0.001),
                    "opacity": ((P, 1), 0.05)}
            if dynamic:
                spec["motion_coeff"] = ((P, 1, 16), 0.00016)
            fp = FlatParams(spec, dev)
            with torch.no_grad():
                fp["xyz"].copy_(sc["means3D"])
                fp["f_dc"].copy_(sc["shs"][:, :1])
                fp["f_rest"].copy_(sc["shs"][:, 1:])
                fp["scaling"].copy_(torch.log(sc["scales"]))
                fp["rotation"].copy_(sc["rotations"])
                op = sc["opacities"].clamp(1e-4, 1 - 1e-4)
                fp["opacity"].copy_(torch.log(op / (1 - op)))
                if dynamic:
                    fp["motion_coeff"].copy_(0.1 * torch.randn(P, 1, 16, generator=g))
            return fp

        self.fp_s, self.fp_d = bucket(static_scene, False), bucket(dynamic_scene, True)
        self.Ps, self.Pd = self.fp_s.shapes["xyz"][0], self.fp_d.shapes["xyz"][0]
        self.time_ind = torch.randint(0, num_frames, (self.Pd,), generator=g).to(dev)
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed + 2)
            self.net = MLPBasisNetwork(128, 16, 26, False)
        self.net = self.net.to(dev)
        emb = self.net.batch_embedding((torch.arange(num_frames, dtype=torch.float32) / num_frames).to(dev))
        self.emb_rows = torch.cat([emb.unsqueeze(0).expand(num_frames, -1, -1), emb.unsqueeze(1)], dim=1).contiguous()
        self.sp_mlp = bind_module_to_flat(self.net, 0.0016, dev)          # the dynamic trainer's deform_network group
        self.net.grad_sinks = None                                         # gradients ACCUMULATE here (stale-gradient semantics)
        cz = 11.0
        qs, ts = [], []
        for i in range(num_frames):
            a = math.radians(orbit_deg) * math.sin(2 * math.pi * i / num_frames)
            qs.append(torch.tensor([math.cos(a / 2), 0.0, math.sin(a / 2), 0.0]))
            ts.append(torch.tensor([-cz * math.sin(a), 0.0, cz - cz * math.cos(a)]))
        self.sp_cam = FlatParams({"cam_q": ((num_frames, 4), 1e-5), "cam_t": ((num_frames, 3), 1e-6)}, dev)
        with torch.no_grad():
            self.sp_cam["cam_q"].copy_(torch.stack(qs))
            self.sp_cam["cam_t"].copy_(torch.stack(ts))
        self.m2 = torch.zeros(self.Ps + self.Pd, 3, device=dev, requires_grad=True)
        self.stats = {"static": DensifyStats.zeros(self.Ps, dev), "dynamic": DensifyStats.zeros(self.Pd, dev)}
        self.raster_state = RasterState()
        self.gt = {}
        self._one = torch.ones((), dtype=torch.float32, device=dev)

    def settings(self, pose_grads: bool) -> GaussianRasterizationSettings:
        return GaussianRasterizationSettings(self.H, self.W, self.tanfovx, self.tanfovy, self.bg, 1.0, self.proj_t,
                                             self.sh_degree, False, False, pose_grads, pose_grads)

    def properties(self, frame: int):
        """get_GS_properties (rodygs.py:68-113) of the frame: (xyz, opacity, scaling, rotation, features) of static ‖ dynamic."""
        from .model_ops import gs_properties
        allb = self.net.motion_basis(self.emb_rows[frame])                 # [T + 1, 16, 7]: birth-time table, then B(t)
        self._last_allb = allb                                             # (tests look at its gradient)
        dxyz, drot = gaussian_deformation_packed(self.fp_d["motion_coeff"], self.time_ind, allb, self.spatial_lr_scale)
        names = ("xyz", "scaling", "rotation", "opacity", "f_dc", "f_rest")
        return gs_properties({k: self.fp_s[k] for k in names}, {k: self.fp_d[k] for k in names}, dxyz, drot)

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

    def forward_backward(self, frame: int, which: str) -> torch.Tensor:
        """Everything of ``train_iteration`` (rodygs.py:198-341) up to the optimiser: render the concatenated cloud, loss,
        backward (gradients ACCUMULATE in both clouds' buckets), statistics of the sub-step's own slice."""
        static = which == "static"
        xyz, opacity, scaling, rot, feats = self.properties(frame)
        vm = pose_view_matrix(self.sp_cam["cam_q"], self.sp_cam["cam_t"], frame)
        if not static:
            vm = vm.detach()
        self.m2.grad = None
        row0, stats = (0, self.stats["static"]) if static else (self.Ps, self.stats["dynamic"])
        out = GaussianRasterizer(self.settings(static), state=self.raster_state)(
            means3D=xyz, means2D=self.m2, shs=feats, opacities=opacity, scales=scaling, rotations=rot, viewmatrix=vm,
            grad_sinks={"densify": stats.sink(row0)})
        loss = fused_photometric_loss(out[0], self.gt[frame], 0.2)
        loss.backward(self._one)
        return loss.detach()

    def step(self, which: str) -> None:
        """``current_gs.optimizer.step(); zero_grad()`` (+ the camera optimiser for the static trainer, rodygs.py:364-369): the
        other trainer's gradients stay where they are."""
        if which == "static":
            fused_adam_(self.fp_s, extra=(self.sp_cam,))
            self.fp_s.zero_grad()
            self.sp_cam.zero_grad()
        else:
            fused_adam_(self.fp_d, extra=(self.sp_mlp,))
            self.fp_d.zero_grad()
            self.sp_mlp.zero_grad()

    def sub_step(self, frame: int, which: str) -> torch.Tensor:
        """One ``train_iteration`` (rodygs.py:198-369) with ``learn_static`` (which == "static") or ``learn_dynamic``."""
        loss = self.forward_backward(frame, which)
        self.step(which)
        return loss

    def iteration(self, it: int, perm) -> tuple:
        """Static sub-step, then dynamic sub-step (each draws its own frame, as the two data loaders of the reference do)."""
        fs = perm[(2 * it) % len(perm)]
        fd = perm[(2 * it + 1) % len(perm)]
        return self.sub_step(fs, "static"), self.sub_step(fd, "dynamic")
