"""Drop-in replacement for ``diff_gauss_pose`` (slothfulxtx/diff-gaussian-rasterization@pose), MI355X-native.

Mirrors the surface the reference uses:
  * ``GaussianRasterizationSettings`` -- built at /root/reference/src/trainer/renderer.py:50-63,
    src/model/rodygs_static.py:221-236, src/evaluator/eval.py:118-133 (same 12 keyword fields, same order);
  * ``GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
    cov3Ds_precomp, viewmatrix)`` -> ``(color, depth, normal, alpha, radii, extra)`` -- called at
    renderer.py:65,87-101;
  * ``rasterize_gaussians`` / ``_RasterizeGaussians`` autograd.Function with gradients for means3D, means2D
    (``.grad[:, :2]`` read at src/trainer/rodygs.py:322-324), shs/colors, opacities, scales, rotations,
    cov3Ds_precomp and **viewmatrix** (camera pose).

All arithmetic runs in hand-written HIP kernels behind the C-ABI of ``include/rodygs_hip.h``; PyTorch only owns
device memory and the stream.  There is no CPU or eager fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib

# module-level knobs (not part of the reference surface)
RENDER_NORMAL = True          # composite the (unused-by-RoDyGS) normal channels
PREZERO_GRAD_ROWS = os.environ.get("RDG_PREZERO_GRAD_ROWS", "1") != "0"   # the forward clears the backward's gradient rows
NREN_HOST_MIRROR = os.environ.get("RDG_NREN_MIRROR", "1") != "0"           # deferred check: the device writes D to pinned memory
# Bucket binning (count / scan / scatter + per-tile sort) is the fast path on ordinary frames, but its cost grows with
# the tile lists' lengths (same-address atomics in the count pass; the per-tile sort leaves its one-wave register form
# above 1024 instances and needs merges above that); the radix path (Gaussians sorted by depth once, their instances
# partitioned by tile in two 8-bit passes) does not care how instances are spread.  Same result bit for bit, so the choice
# is made per frame from the PREVIOUS frame's largest tile list and instance count (they arrive together: no extra
# read-back), with hysteresis.  Measured at 1080p (profiles/r04_binning_modes.txt): 3.6 M instances (440 per tile) bucket
# 0.14 ms / radix 0.26; 9.4 M (1 150 per tile) 0.36 / 0.35; 17 M (2 100 per tile) 0.71 / 0.45.
BIN_RADIX_ABOVE = 32768
BIN_BUCKET_BELOW = 16384
BIN_RADIX_MEAN_LIST_ABOVE = 1000        # instances per tile, averaged over the tile grid
BIN_BUCKET_MEAN_LIST_BELOW = 800
# Compositing of long tile lists: one workgroup walks a tile's list front to back, which is serial in the list length (a
# 200 k-instance tile: 15 ms).  Lists above 4096 instances can be cut into segments composited by a workgroup each
# (csrc/rdg_render.hip "split path"; the backward follows the forward's choice) -- three extra launches in the forward and
# one in the backward, so, like the binning algorithm, the choice is made per frame from the previous frame's largest list.
SPLIT_ABOVE = 4096            # = RDG_SPLIT_MIN of the library: shorter lists are never split
SPLIT_BELOW = 3072
# RDG_BIN_MODE=radix / bucket in the environment: every forward takes that binning (A/B switch; read once, here)
_FORCE_RADIX = os.environ.get("RDG_BIN_MODE", "") == "radix"
_FORCE_BUCKET = os.environ.get("RDG_BIN_MODE", "") == "bucket"

# Deterministic backward (RDG_DETERMINISTIC=1 or set at run time): the compositing backward stores
# per-(wave, list position) partial rows and reduces them per Gaussian in a fixed order instead of accumulating with float
# atomics (rdg_composite_backward_det) -- two runs give the same bits; ~2x the compositing backward (1.3x the train step) and 256 B per instance.
DETERMINISTIC = os.environ.get("RDG_DETERMINISTIC", "0") == "1"

# Tight tile rectangles (RdgRasterSettings.cull; RDG_CULL=0 in the environment or RasterState(cull=False) = the reference's
# 3-sigma squares): the per-Gaussian stage hands the binning stage only the tiles that hold a pixel centre inside the box of
# the splat's alpha >= 1/255 ellipse -- 27-32 % fewer (tile, Gaussian) instances on the benchmark frames, same image, same
# gradients, same radii; every tile list a subsequence of the reference's.  The stage entry points the bit-exact key-stream
# tests drive (rdg_preprocess_forward + rdg_bin_forward) take the switch from their settings struct like everything else.
CULL = os.environ.get("RDG_CULL", "1") != "0"

# hipGraph capture (rodygs_amd.trainstep.GraphedStep sets RasterState.graph_capture around capture; the module flag is the
# same switch for every state): the forward touches nothing on the host -- capacity and binning mode come from the hints
# of the warm-up steps, the instance count stays on the device (last_num_rendered()) and is checked by the owner of the
# graph between replays.
GRAPH_CAPTURE = False

HINT_DECAY = 0.98                 # deferred mode: how fast the capacity hint follows lighter frames down (per checked frame)
DEFERRED_OVERFLOW_CHECK = False   # opt-in (see poll_overflow); the default reads D once per forward, like upstream


class RasterizerCapacityOverflow(RuntimeError):
    """Raised (deferred mode only) when an earlier forward produced more (tile, Gaussian) instances than its binning
    workspace could hold.  That frame was rendered as an EMPTY scene (background, zero gradients); the capacity hint
    has been raised, so re-running the frame succeeds."""


class RasterState:
    """What the rasterizer remembers from one frame to the next, owned by the CALLER ("reentrant per
    workspace; no global mutable state"): the instance count of the last frame of every (P, H, W) -- it sizes the next
    frame's binning workspace without a host wait --, the two speed hints derived from the last frame's largest tile list
    (binning algorithm, split compositing), the forwards whose instance count has not been checked yet (deferred mode)
    and the workspaces of the most recent forward.  None of it changes a result; all of it decides speed, so two scenes of
    equal (P, H, W) -- a static and a dynamic model, a training and an evaluation renderer -- should not share one: give
    each its own ``RasterState`` (``GaussianRasterizer(settings, state=...)``; ``DynamicScene``, ``PoseOptimizer`` and the
    sharded scenes do).  A caller that builds a rasterizer per call without one, as the reference does
    (/root/reference/src/trainer/renderer.py:65), gets ``DEFAULT_STATE``.  Safe to use from several threads: the host
    bookkeeping of a forward runs under the state's lock (the launches themselves do not)."""

    def __init__(self, deterministic: Optional[bool] = None, force_radix: Optional[bool] = None,
                 force_bucket: Optional[bool] = None, deferred_overflow_check: Optional[bool] = None,
                 render_normal: Optional[bool] = None, cull: Optional[bool] = None):
        """The mode switches ride on the state (two trainers in one process can differ in them): ``deterministic``
        (compositing backward without float atomics), ``force_radix`` / ``force_bucket`` (binning algorithm on every frame),
        ``deferred_overflow_check`` (no host wait for the instance count), ``render_normal`` (composite the normal channels),
        ``cull`` (tight tile rectangles: see CULL).
        None = follow the module attribute of that name (DETERMINISTIC, _FORCE_RADIX, _FORCE_BUCKET, DEFERRED_OVERFLOW_CHECK,
        RENDER_NORMAL, CULL) as it stands when a forward runs -- the process-wide default, initialised from the environment."""
        self.deterministic, self.force_radix, self.force_bucket = deterministic, force_radix, force_bucket
        self.deferred_overflow_check, self.render_normal = deferred_overflow_check, render_normal
        self.cull = cull
        self.capacity_hint = {}       # (P, H, W) -> last num_rendered
        self.d_high = {}              # (P, H, W) -> slowly decaying maximum of num_rendered over the checked frames (HINT_DECAY)
        self.bin_hint = {}            # (P, H, W) -> 1 while the largest tile list of the last frame calls for the radix path
        self.split_hint = {}          # (P, H, W) -> 1 while it calls for the split compositing path
        self.pending = []             # (stream, pinned int32[2], key, capacity, device pair) of forwards not yet checked
        self.pinned_free = []
        self.last_nren = None         # (nren int32[2] device tensor, key, capacity) of the most recent forward
        self.last_image = None        # (image workspace, H, W) of the most recent forward
        self.graph_capture = False
        self.nren_max = None          # optional device int32[1]: sticky maximum of D (RdgRasterSettings.num_rendered_max)
        self.nren_max_key = None      # ... folded into by forwards of THIS (P, H, W) only (None: by every forward)
        self.det_ws = None            # deterministic mode: the per-instance row workspace, kept across steps
        # graph capture only (trainstep.GraphedStep): a second stream the pose-gradient chain of backward is forked onto
        # (RdgRasterSettings.aux_stream: a graph branch next to the deformation / MLP backward), the persistent [4,4] buffer
        # dL/dviewmatrix is written to (it is read on that stream after the backward's own tensors are gone) and the gradient
        # workspaces kept alive until the owner has joined the streams
        self.aux_stream = None
        self.pose_grad = None
        self.pose_fork_eager = None   # (stream, [4,4] buffer): the same fork for ONE eager step, set and cleared by its owner
        self.keep_alive = []
        self.lock = threading.RLock()

    def mode(self, name: str) -> bool:
        """Effective value of a mode switch for this state (its own setting, else the module default)."""
        v = getattr(self, name)
        if v is not None:
            return bool(v)
        return bool({"deterministic": DETERMINISTIC, "force_radix": _FORCE_RADIX, "force_bucket": _FORCE_BUCKET,
                     "deferred_overflow_check": DEFERRED_OVERFLOW_CHECK, "render_normal": RENDER_NORMAL,
                     "cull": CULL}[name])

    def note_instances(self, key, n: int) -> None:
        """Fold a frame's instance count into the slowly decaying maximum a graph owner sizes its fixed capacity with (consecutive
        frames of a training run can be far apart on the camera path: the LAST frame says little about the next hundred)."""
        self.d_high[key] = max(int(n), int(self.d_high.get(key, 0) * HINT_DECAY))

    def pinned_slot(self) -> torch.Tensor:
        return self.pinned_free.pop() if self.pinned_free else torch.empty(2, dtype=torch.int32).pin_memory()

    def note_largest_tile(self, key, largest: int, n: int = 0) -> None:
        """What a frame's largest tile list and instance count say about the next frame of that (P, H, W)."""
        tiles = ((key[2] + 15) // 16) * ((key[1] + 15) // 16)
        mean = n / max(tiles, 1)
        if (largest > BIN_RADIX_ABOVE or mean > BIN_RADIX_MEAN_LIST_ABOVE) and not self.mode("force_bucket"):
            self.bin_hint[key] = 1
        elif largest < BIN_BUCKET_BELOW and mean < BIN_BUCKET_MEAN_LIST_BELOW:
            self.bin_hint.pop(key, None)
        if largest > SPLIT_ABOVE:
            self.split_hint[key] = 1
        elif largest < SPLIT_BELOW:
            self.split_hint.pop(key, None)

    def poll_overflow(self, block: bool = False) -> None:
        """Deferred mode: check the instance counts of forwards whose count has landed (all of them if ``block``)."""
        with self.lock:
            while self.pending:
                stream, host, key, cap, nren = self.pending[0]
                if int(host[0]) < 0:                  # the binning stage of that forward has not written its mirror yet
                    if not block:
                        return
                    stream.synchronize()
                    if int(host[0]) < 0:              # (a path that does not mirror: read the device pair)
                        host[0], host[1] = (int(v) for v in nren.tolist())
                n = int(host[0])
                self.pending.pop(0)
                self.pinned_free.append(host)
                if n >= _INSTANCE_LIMIT:
                    raise RuntimeError(_too_many(key))
                # a slowly decaying maximum (2 % per checked frame): consecutive frames of a training run can be far apart on
                # the camera path, and a hint that followed every light frame down (10 % per frame until round 5) was outgrown by
                # the next heavy one -- 52 frames rendered empty in a 2 500-step loop once the frames differed by more than 25 %
                self.capacity_hint[key] = max(n, int(self.capacity_hint.get(key, 0) * HINT_DECAY))
                self.note_instances(key, n)
                self.note_largest_tile(key, int(host[1]), n)
                if n > cap:
                    self.capacity_hint[key] = n
                    raise RasterizerCapacityOverflow(
                        f"rasterizer forward for (P,H,W)={key} needed {n} instances, workspace held {cap}: that frame "
                        f"was rendered empty; re-run it (the capacity hint is now {n})")


DEFAULT_STATE = RasterState()
# the default state's tables under their historical module names (tests and scripts reach for them)
_CAPACITY_HINT = DEFAULT_STATE.capacity_hint
_BIN_HINT = DEFAULT_STATE.bin_hint
_SPLIT_HINT = DEFAULT_STATE.split_hint
_PENDING = DEFAULT_STATE.pending


def _note_largest_tile(key, largest: int, n: int = 0) -> None:
    DEFAULT_STATE.note_largest_tile(key, largest, n)


_INSTANCE_LIMIT = 2 ** 31 - 1      # the device saturates its (tile, Gaussian) instance count here (rdg_scan_block_sums_kernel)


def _too_many(key) -> str:
    return (f"rasterizer forward for (P,H,W)={key}: 2^31 - 1 or more (tile, Gaussian) instances -- the instance index is "
            "32 bits wide, as upstream's; the frame was rendered empty.  Scales this large usually mean a diverged run")


def poll_overflow(block: bool = False, state: Optional[RasterState] = None) -> None:
    (state or DEFAULT_STATE).poll_overflow(block)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    projmatrix: torch.Tensor
    sh_degree: int
    prefiltered: bool
    debug: bool
    enable_cov_grad: bool = True
    enable_sh_grad: bool = True


def _c_settings(rs: GaussianRasterizationSettings, P: int, M: int, state: Optional["RasterState"] = None) -> _lib.RdgRasterSettings:
    state = state if state is not None else DEFAULT_STATE
    s = _lib.RdgRasterSettings()
    s.P = P
    s.M = M
    s.sh_degree = int(rs.sh_degree)
    s.image_height = int(rs.image_height)
    s.image_width = int(rs.image_width)
    s.tanfovx = float(rs.tanfovx)
    s.tanfovy = float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    s.enable_cov_grad = int(bool(rs.enable_cov_grad))
    s.enable_sh_grad = int(bool(rs.enable_sh_grad))
    s.render_normal = int(state.mode("render_normal"))
    s.bin_mode = 1 if state.mode("force_radix") else 0
    s.num_rendered_stats = 0
    s.list_hints = 0
    s.cull = int(state.mode("cull"))
    return s


def _f32c(t: Optional[torch.Tensor], name: str, dev) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"rodygs_amd rasterizer: `{name}` must be a CUDA/HIP tensor (no CPU fallback exists)")
    if t.device != dev:
        raise RuntimeError(f"rodygs_amd rasterizer: `{name}` is on {t.device}, expected {dev}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"rodygs_amd rasterizer: `{name}` must be float32, got {t.dtype}")
    return t.detach().contiguous()


def _empty(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if (t is None or t.numel() == 0) else t


def _composite_backward(L, ctx, bg, geom, binning, image, g_color, g_depth, g_alpha, gws, g_normal=None):
    """The compositing half of backward: float-atomic accumulation, or the deterministic two-pass form."""
    if not ctx.deterministic:
        _lib.check(L.rdg_composite_backward(C.byref(ctx.cs), _lib.ptr(bg), _lib.ptr(geom), _lib.ptr(binning),
                                            ctx.capacity, _lib.ptr(image), _lib.ptr(g_color), _lib.ptr(g_depth),
                                            _lib.ptr(g_alpha), _lib.ptr(g_normal), _lib.ptr(gws), _lib.stream_ptr()),
                   "rdg_composite_backward")
        return
    # D is known on the host except in deferred-overflow mode and under graph capture, where the capacity bounds it (the
    # library clears only the rows of the frame's own instances: it reads D on the device).  The workspace is kept on
    # the state and reused while it is large enough -- 256 B per instance is 1 GB at P = 1 M
    n_inst = ctx.num_rendered if 0 <= ctx.num_rendered <= ctx.capacity else ctx.capacity
    need = L.rdg_det_bytes(n_inst)
    st = ctx.state
    det = st.det_ws
    if det is None or det.numel() < need or det.device != gws.device or st.graph_capture or GRAPH_CAPTURE:
        det = torch.empty(need, dtype=torch.uint8, device=gws.device)
        if not (st.graph_capture or GRAPH_CAPTURE):
            st.det_ws = det
    _lib.check(L.rdg_composite_backward_det(C.byref(ctx.cs), _lib.ptr(bg), _lib.ptr(geom), _lib.ptr(binning),
                                            ctx.capacity, _lib.ptr(image), _lib.ptr(g_color), _lib.ptr(g_depth),
                                            _lib.ptr(g_alpha), _lib.ptr(g_normal), _lib.ptr(gws), _lib.ptr(det), n_inst,
                                            _lib.stream_ptr()), "rdg_composite_backward_det")


def _bind_densify_stats(ctx, P: int) -> None:
    """grad_sinks["densify"] = {"grad_accum", "denom", "max_radii" (float32, `rows` elements each, any may be missing),
    "row0" (default 0), "rows" (default: the arrays' length)}: the per-Gaussian backward kernel updates the densification
    statistics of the reference's train loop (/root/reference/src/trainer/rodygs.py:316-341, rodygs_static.py:317-319)
    itself -- RdgRasterSettings.densify_*.  Applied by the FIRST backward through a forward only: a second backward
    through the same graph (retain_graph) must not count the frame twice."""
    cs = ctx.cs
    cs.densify_grad_accum = cs.densify_denom = cs.densify_max_radii = None
    cs.densify_rows = cs.densify_row0 = 0
    sink = None if ctx.grad_sinks is None else ctx.grad_sinks.get("densify")
    if sink is None or getattr(ctx, "densify_done", False):
        return
    ctx.densify_done = True
    row0 = int(sink.get("row0", 0))
    arrs = {k: sink.get(k) for k in ("grad_accum", "denom", "max_radii")}
    rows = sink.get("rows")
    for k, t in arrs.items():
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"grad_sinks['densify']['{k}'] must be a contiguous float32 GPU tensor")
        rows = t.numel() if rows is None else rows
        if t.numel() != rows:
            raise RuntimeError("grad_sinks['densify']: the statistics arrays must all hold `rows` elements")
    if rows is None:
        return
    rows = int(rows)
    if row0 < 0 or row0 + rows > P:
        raise RuntimeError(f"grad_sinks['densify']: rows [{row0}, {row0 + rows}) do not lie inside the {P} Gaussians")
    cs.densify_grad_accum = _lib.ptr(arrs["grad_accum"])
    cs.densify_denom = _lib.ptr(arrs["denom"])
    cs.densify_max_radii = _lib.ptr(arrs["max_radii"])
    cs.densify_row0, cs.densify_rows = row0, rows


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                raster_settings, grad_sinks=None, state=None):
        L = _lib.lib()
        ctx.grad_sinks = grad_sinks
        state = state if state is not None else DEFAULT_STATE
        ctx.state = state
        dev = means3D.device
        sh, colors_precomp = _empty(sh), _empty(colors_precomp)
        scales, rotations, cov3Ds_precomp = _empty(scales), _empty(rotations), _empty(cov3Ds_precomp)
        m3 = _f32c(means3D, "means3D", dev)
        shs = _f32c(sh, "shs", dev)
        col = _f32c(colors_precomp, "colors_precomp", dev)
        op = _f32c(opacities, "opacities", dev)
        sc = _f32c(scales, "scales", dev)
        ro = _f32c(rotations, "rotations", dev)
        cov = _f32c(cov3Ds_precomp, "cov3Ds_precomp", dev)
        vm = _f32c(viewmatrix, "viewmatrix", dev)
        pm = _f32c(raster_settings.projmatrix, "projmatrix", dev)
        bg = _f32c(raster_settings.bg, "bg", dev)
        P = m3.shape[0]
        if m3.dim() != 2 or m3.shape[1] != 3:
            raise RuntimeError("means3D must be [P,3]")
        if op.numel() != P:
            raise RuntimeError("opacities must be [P,1]")
        if vm.numel() != 16 or pm.numel() != 16 or bg.numel() != 3:
            raise RuntimeError("viewmatrix/projmatrix must be [4,4] and bg [3]")
        M = 0
        if shs is not None:
            if shs.dim() != 3 or shs.shape[0] != P or shs.shape[2] != 3:
                raise RuntimeError("shs must be [P,K,3]")
            M = shs.shape[1]
        if P == 0:
            # empty cloud: empty tensors have NULL data pointers, which the C-ABI reads as "not provided";
            # hand it never-dereferenced placeholders so the normal path renders the background
            ph = torch.zeros(16, 3, dtype=torch.float32, device=dev)
            shs, col, cov, sc, ro, M = ph, None, None, ph, ph, 16
            op = ph if op.numel() == 0 else op
            m3 = ph
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        cs = _c_settings(raster_settings, P, M, state)
        n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
        # the backward follows the mode the forward ran in (the state may be switched between the two)
        ctx.deterministic = det_mode = state.mode("deterministic")

        with torch.cuda.device(dev):
            u8 = dict(dtype=torch.uint8, device=dev)
            geom = torch.empty(L.rdg_geom_bytes(P), **u8)
            image = torch.empty(L.rdg_image_bytes(H, W), **u8)
            color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
            depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            normal = torch.empty(3, H, W, dtype=torch.float32, device=dev)
            alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            radii = torch.empty(P, dtype=torch.int32, device=dev)
            # a backward will follow: its gradient rows are cleared by the compositing forward (RdgRasterSettings
            # .zero_grad_ws: free there, a 12 us launch at the head of the backward otherwise)
            gws = None
            if PREZERO_GRAD_ROWS and P > 0 and not det_mode and any(ctx.needs_input_grad):
                gws = torch.empty(L.rdg_grad_bytes(P), **u8)
                cs.zero_grad_ws = gws.data_ptr()
            # [0] = D (instances), [1] = largest tile list: both written by the forward on every path (the block-sum scan
            # writes [0] even for an empty cloud, the tile scan / tile max kernels write [1] also on capacity overflow),
            # so no fill launch
            nren = torch.empty(2, dtype=torch.int32, device=dev)
            key = (P, H, W)
            cs.num_rendered_stats = 1
            capture = GRAPH_CAPTURE or state.graph_capture
            stream = _lib.stream_ptr()
            with state.lock:
                cs.bin_mode = max(int(cs.bin_mode), int(state.bin_hint.get(key, 0)))
                # deterministic mode: the choice must not depend on what the previous frame looked like (the split path
                # associates the transmittance product differently: same result to the last bits only)
                cs.list_hints = 1 if det_mode else int(state.split_hint.get(key, 0))
                cap = max(int(state.capacity_hint.get(key, 0) * 1.25) + 4096, 4 * P + 4096)
                if capture and key not in state.capacity_hint:
                    raise RuntimeError("graph capture needs a warm-up forward of this (P, H, W) on the same RasterState first")
                deferred = state.mode("deferred_overflow_check") and key in state.capacity_hint and not capture
                if deferred:
                    state.poll_overflow(block=False)
                host = None
                if deferred:
                    # D and the largest list are ALSO written to pinned host memory by the binning stage itself
                    # (RdgRasterSettings.num_rendered_host): no copy and no event on the stream; -1 = not there yet
                    host = state.pinned_slot()
                    host[0], host[1] = -1, -1
                    if NREN_HOST_MIRROR:
                        cs.num_rendered_host = host.data_ptr()
                if state.nren_max is not None and state.nren_max_key in (None, key):
                    cs.num_rendered_max = state.nren_max.data_ptr()     # sticky maximum of D (graph replay)
            while True:
                binning = torch.empty(L.rdg_binning_bytes(cap, n_tiles), **u8)
                rc = L.rdg_rasterize_forward(C.byref(cs), _lib.ptr(bg), _lib.ptr(m3), _lib.ptr(shs), _lib.ptr(col),
                                             _lib.ptr(op), _lib.ptr(sc), _lib.ptr(ro), _lib.ptr(cov), _lib.ptr(vm),
                                             _lib.ptr(pm), _lib.ptr(geom), _lib.ptr(binning), cap, _lib.ptr(image),
                                             _lib.ptr(color), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(alpha),
                                             _lib.ptr(radii), _lib.ptr(nren), stream)
                _lib.check(rc, "rdg_rasterize_forward")
                state.last_nren = (nren, key, cap)
                if capture:
                    n = -1
                    break
                if deferred:
                    # opt-in: no host wait at all.  D goes to pinned memory asynchronously and is checked by
                    # poll_overflow() at the next forward / on demand; on overflow the device has rendered an
                    # empty scene (every tile range zero) and RasterizerCapacityOverflow is raised then.
                    if not NREN_HOST_MIRROR:           # the copy-engine form: a blit and an event on the stream per frame
                        host.copy_(nren, non_blocking=True)
                    with state.lock:
                        state.pending.append((torch.cuda.current_stream(dev), host, key, cap, nren))
                    n = -1
                    break
                # one host read AFTER the whole forward is queued (upstream stalls mid-pipeline instead)
                n, largest = (int(v) for v in nren.tolist())
                if n >= _INSTANCE_LIMIT:
                    raise RuntimeError(_too_many(key))
                with state.lock:
                    # an owner that checks only now and then (GraphedStep.check, the deferred mode) may have left a HIGHER
                    # record of this shape: keep it, decayed like everywhere else, instead of following this one frame down
                    checked = (state.nren_max is not None and state.nren_max_key in (None, key)) or bool(state.pending)
                    state.capacity_hint[key] = max(n, int(state.capacity_hint.get(key, 0) * HINT_DECAY)) if checked else n
                    state.note_instances(key, n)
                    state.note_largest_tile(key, largest, n)
                if n <= cap:
                    break
                cap = int(n * 1.25) + 4096
        state.last_image = (image, H, W)
        # outputs the loss never touched (normal, alpha, usually depth) arrive as None in backward instead of as
        # zero-filled images the kernel would then read
        ctx.set_materialize_grads(False)
        ctx.raster_settings = raster_settings
        ctx.empty_cloud = (P == 0)
        ctx.cs = cs
        ctx.gws = gws                 # zeroed gradient rows for the FIRST backward through this graph
        ctx.capacity = cap
        ctx.num_rendered = n
        ctx.has = (shs is not None, col is not None, sc is not None, cov is not None)
        ctx.save_for_backward(m3, shs, col, op, sc, ro, cov, vm, pm, bg, radii, geom, binning, image)
        extra = torch.empty(0, dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(radii, extra)
        return color, depth, normal, alpha, radii, extra

    @staticmethod
    def backward(ctx, g_color, g_depth, g_normal, g_alpha, g_radii, g_extra):
        if g_normal is not None and not ctx.cs.render_normal:
            raise RuntimeError("rodygs_amd rasterizer: rendered_normal received an upstream gradient, but the forward ran "
                               "with rasterizer.RENDER_NORMAL = False (the image is all zeros and has no graph)")
        if ctx.empty_cloud or (g_color is None and g_depth is None and g_alpha is None and g_normal is None):
            return (None,) * 12
        L = _lib.lib()
        m3, shs, col, op, sc, ro, cov, vm, pm, bg, radii, geom, binning, image = ctx.saved_tensors
        dev = m3.device
        P = m3.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)

        def gc(t):
            return None if t is None else t.to(torch.float32).contiguous()

        # rendered_normal is composited from per-Gaussian normals that are constants of the graph: its gradient reaches the
        # inputs through the compositing weights only (no RoDyGS loss reads it, src/trainer/rodygs.py:272-309)
        g_color, g_depth, g_alpha, g_normal = gc(g_color), gc(g_depth), gc(g_alpha), gc(g_normal)
        with torch.cuda.device(dev):
            gws, ctx.gws = ctx.gws, None          # the rows the forward cleared serve one backward
            _bind_densify_stats(ctx, P)
            ctx.cs.grad_rows_zeroed = 1 if (gws is not None and not ctx.deterministic) else 0
            if gws is None:
                gws = torch.empty(L.rdg_grad_bytes(P), dtype=torch.uint8, device=dev)
            d_m3 = torch.empty(P, 3, **f32)
            d_m2 = torch.empty(P, 3, **f32)
            d_op = torch.empty_like(op)
            fused_adam = ctx.grad_sinks is not None and ctx.grad_sinks.get("shs_adam") is not None
            d_sh = torch.empty_like(shs) if (shs is not None and not fused_adam) else None
            sink_sh = None if ctx.grad_sinks is None else ctx.grad_sinks.get("shs")
            if sink_sh is not None and shs is not None:
                # the kernel overwrites every element of dL/dshs: let it write straight into the caller's buffer
                # (e.g. a flat gradient bucket) and return nothing through autograd for that input
                if sink_sh.shape != shs.shape or not sink_sh.is_contiguous() or sink_sh.dtype != torch.float32:
                    raise RuntimeError("grad_sinks['shs'] must be a contiguous float32 tensor shaped like shs")
                d_sh = sink_sh
            d_col = torch.empty_like(col) if col is not None else None
            d_sc = torch.empty_like(sc) if sc is not None else None
            d_ro = torch.empty_like(ro) if ro is not None else None
            d_cov = torch.empty_like(cov) if cov is not None else None
            d_vm = torch.empty(4, 4, **f32)
            ctx.cs.aux_stream = None
            st_ = ctx.state
            if st_.graph_capture and st_.aux_stream is not None and st_.pose_grad is not None and not ctx.deterministic:
                ctx.cs.aux_stream = st_.aux_stream.cuda_stream
                d_vm = st_.pose_grad
                st_.keep_alive.append(gws)
            elif (st_.pose_fork_eager is not None and not st_.graph_capture and not GRAPH_CAPTURE and not ctx.deterministic
                  and gws is not None):
                # the eager step's fork (trainstep.DynamicScene.train_step): same branch, no graph -- the allocator must know
                # that the gradient workspace is read on the second stream after this function returns
                aux_, d_vm = st_.pose_fork_eager
                ctx.cs.aux_stream = aux_.cuda_stream
                gws.record_stream(aux_)
                vm.record_stream(aux_)        # rdg_pose_finalize reads the view matrix there; this node's saved tensors die with it
            fused = None if ctx.grad_sinks is None else ctx.grad_sinks.get("shs_adam")
            if fused is not None:
                # optimizer in backward for the SH features: dL/dshs never leaves the kernel's LDS tile; the kernel
                # updates `param` (which must BE the tensor passed as shs), exp_avg and exp_avg_sq in place.
                # The saved shs shares that storage (no autograd version bump), so a second backward through this
                # node -- loss.backward(retain_graph=True) twice, /root/reference/src/trainer/rodygs.py:310 -- would
                # differentiate at already-stepped parameters and step them again: one shot only.
                if getattr(ctx, "shs_adam_consumed", False):
                    raise RuntimeError("grad_sinks['shs_adam']: this forward's backward already applied the SH Adam "
                                       "step; a second backward through the same graph is not allowed with the "
                                       "optimizer-in-backward sink (use grad_sinks['shs'] or no sink for retain_graph)")
                ctx.shs_adam_consumed = True
                if shs is None or col is not None or cov is not None or sc is None:
                    raise RuntimeError("grad_sinks['shs_adam'] needs shs + scales/rotations inputs")
                prm = fused["param"]
                if prm.data_ptr() != shs.data_ptr() or prm.numel() != shs.numel():
                    raise RuntimeError("grad_sinks['shs_adam']['param'] must be the storage passed to the rasterizer as shs")
                _composite_backward(L, ctx, bg, geom, binning, image, g_color, g_depth, g_alpha, gws, g_normal)
                if fused.get("step_scalars") is not None:
                    # graph replay: the bias corrections are read from device memory (RdgStepScalars)
                    _lib.check(L.rdg_preprocess_backward_adam_dev(
                        C.byref(ctx.cs), _lib.ptr(m3), _lib.ptr(shs), _lib.ptr(op), _lib.ptr(sc), _lib.ptr(ro),
                        _lib.ptr(vm), _lib.ptr(pm), _lib.ptr(radii), _lib.ptr(geom), _lib.ptr(gws), _lib.ptr(d_m3),
                        _lib.ptr(d_m2), _lib.ptr(d_op), _lib.ptr(d_sc), _lib.ptr(d_ro), _lib.ptr(d_vm),
                        _lib.ptr(fused["exp_avg"]), _lib.ptr(fused["exp_avg_sq"]), int(fused["head_len"]),
                        float(fused["lr_head"]), float(fused["lr_tail"]), float(fused["betas"][0]),
                        float(fused["betas"][1]), float(fused["eps"]), _lib.ptr(fused["step_scalars"]),
                        _lib.stream_ptr()), "rdg_preprocess_backward_adam_dev")
                    return d_m3, d_m2, None, None, d_op, d_sc, d_ro, None, d_vm, None, None, None
                step = fused["step"]() if callable(fused["step"]) else int(fused["step"])
                _lib.check(L.rdg_preprocess_backward_adam(
                    C.byref(ctx.cs), _lib.ptr(m3), _lib.ptr(shs), _lib.ptr(op), _lib.ptr(sc), _lib.ptr(ro), _lib.ptr(vm),
                    _lib.ptr(pm), _lib.ptr(radii), _lib.ptr(geom), _lib.ptr(gws), _lib.ptr(d_m3), _lib.ptr(d_m2),
                    _lib.ptr(d_op), _lib.ptr(d_sc), _lib.ptr(d_ro), _lib.ptr(d_vm), _lib.ptr(fused["exp_avg"]),
                    _lib.ptr(fused["exp_avg_sq"]), int(fused["head_len"]), float(fused["lr_head"]),
                    float(fused["lr_tail"]), float(fused["betas"][0]), float(fused["betas"][1]), float(fused["eps"]),
                    step, _lib.stream_ptr()), "rdg_preprocess_backward_adam")
                return d_m3, d_m2, None, None, d_op, d_sc, d_ro, None, d_vm, None, None, None
            if ctx.deterministic:
                _composite_backward(L, ctx, bg, geom, binning, image, g_color, g_depth, g_alpha, gws, g_normal)
                rc = L.rdg_preprocess_backward(C.byref(ctx.cs), _lib.ptr(m3), _lib.ptr(shs), _lib.ptr(col), _lib.ptr(op),
                                               _lib.ptr(sc), _lib.ptr(ro), _lib.ptr(cov), _lib.ptr(vm), _lib.ptr(pm),
                                               _lib.ptr(radii), _lib.ptr(geom), _lib.ptr(gws), _lib.ptr(d_m3),
                                               _lib.ptr(d_m2), _lib.ptr(d_sh), _lib.ptr(d_col), _lib.ptr(d_op),
                                               _lib.ptr(d_sc), _lib.ptr(d_ro), _lib.ptr(d_cov), _lib.ptr(d_vm),
                                               _lib.stream_ptr())
                _lib.check(rc, "rdg_preprocess_backward")
            else:
                rc = L.rdg_rasterize_backward(C.byref(ctx.cs), _lib.ptr(bg), _lib.ptr(m3), _lib.ptr(shs), _lib.ptr(col),
                                              _lib.ptr(op), _lib.ptr(sc), _lib.ptr(ro), _lib.ptr(cov), _lib.ptr(vm),
                                              _lib.ptr(pm), _lib.ptr(radii), _lib.ptr(geom), _lib.ptr(binning),
                                              ctx.capacity, _lib.ptr(image), _lib.ptr(g_color), _lib.ptr(g_depth),
                                              _lib.ptr(g_alpha), _lib.ptr(g_normal), _lib.ptr(gws), _lib.ptr(d_m3),
                                              _lib.ptr(d_m2),
                                              _lib.ptr(d_sh), _lib.ptr(d_col), _lib.ptr(d_op), _lib.ptr(d_sc),
                                              _lib.ptr(d_ro), _lib.ptr(d_cov), _lib.ptr(d_vm), _lib.stream_ptr())
                _lib.check(rc, "rdg_rasterize_backward")
        if sink_sh is not None:
            d_sh = None
            ready = ctx.grad_sinks.get("on_shs_ready")
            if ready is not None:
                ready()          # the sink now holds the final dL/dshs (stream-ordered): e.g. start its all-reduce
        return d_m3, d_m2, d_sh, d_col, d_op, d_sc, d_ro, d_cov, d_vm, None, None, None


def last_num_rendered(state: Optional[RasterState] = None):
    """(nren, key, capacity): the device tensor [D, largest tile list] the most recent forward (of ``state``) wrote, the
    (P, H, W) key of its hints and the capacity it ran with.  The owner of a captured graph reads it between replays:
    D > capacity means that frame was rendered empty and the graph must be re-captured with the raised hint."""
    state = state or DEFAULT_STATE
    if state.last_nren is None:
        raise RuntimeError("no rasterizer forward has run yet")
    return state.last_nren


def last_compositing_state(state: Optional[RasterState] = None):
    """(final_T[H,W] float32, n_contrib[H,W] int32) of the most recent forward -- the per-pixel state the backward
    replays from (``rdg_image_export``).  sum(n_contrib) is S, the pixel-splat pairs the forward walked."""
    state = state or DEFAULT_STATE
    if state.last_image is None:
        raise RuntimeError("no rasterizer forward has run yet")
    image, H, W = state.last_image
    with torch.cuda.device(image.device):
        final_T = torch.empty(H, W, dtype=torch.float32, device=image.device)
        n_contrib = torch.empty(H, W, dtype=torch.int32, device=image.device)
        _lib.check(_lib.lib().rdg_image_export(H, W, _lib.ptr(image), _lib.ptr(final_T), _lib.ptr(n_contrib),
                                               _lib.stream_ptr()), "rdg_image_export")
    return final_T, n_contrib


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                        raster_settings, grad_sinks=None, state=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, viewmatrix, raster_settings, grad_sinks, state)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, state: Optional[RasterState] = None):
        """``state`` (outside the reference surface): the caller-owned ``RasterState`` this rasterizer's frame-to-frame
        memory lives in; None = the process default (what the reference's rasterizer-per-call pattern gets)."""
        super().__init__()
        self.raster_settings = raster_settings
        self.state = state

    def markVisible(self, positions: torch.Tensor, viewmatrix: torch.Tensor) -> torch.Tensor:
        """Frustum test of the upstream API (no reference caller): view-space z > 0.2."""
        with torch.no_grad():
            v = viewmatrix.reshape(16)
            z = (v[2] * positions[:, 0] + v[6] * positions[:, 1]) + v[10] * positions[:, 2] + v[14]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None, viewmatrix=None, extra_attrs=None, grad_sinks=None):
        """Reference call signature (renderer.py:87-101).  ``grad_sinks`` is an extension outside the reference
        surface (``extra_attrs``: see ``_composite_extra``): {"shs": tensor} makes backward write dL/dshs into that tensor instead of returning it;
        {"shs_adam": {param, exp_avg, exp_avg_sq, head_len, lr_head, lr_tail, betas, eps, step}} makes backward apply
        the Adam step of the SH features itself (``rdg_preprocess_backward_adam``) -- the parameters change DURING
        backward, so use it only where backward runs exactly once per optimiser step and nothing else needs dL/dshs
        (rodygs_amd/trainstep.py does, for the single-GPU photometric step); {"densify": {...}} makes backward update
        the densification statistics of the train loop (see ``_bind_densify_stats``)."""
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or (
                (scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if viewmatrix is None:
            raise Exception("viewmatrix must be given (it is a differentiable forward argument in the pose branch)")
        out = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                  viewmatrix, self.raster_settings, grad_sinks, self.state)
        if extra_attrs is None:
            return out
        return (*out[:5], self._composite_extra(extra_attrs, means3D, means2D, opacities, scales, rotations,
                                                cov3Ds_precomp, viewmatrix))

    def _composite_extra(self, extra_attrs, means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, viewmatrix):
        """``extra_attrs`` [P,E] (an upstream kwarg no RoDyGS caller passes; semantics assumed, SURVEY.md section 7 open
        question 3): per-Gaussian attributes composited with the colour's own blending weights, no background term ->
        ``extra`` [E,H,W].  The compositing kernels carry three colour channels, so the attributes go through them three
        at a time as precomputed colours over a black background -- E / 3 more passes of the same forward, each a node of
        the autograd graph, so their gradients reach the attributes and, through the weights, every geometric input
        (they accumulate on ``means2D`` and ``viewmatrix`` with the main pass's).  Not a fast path: it exists so that a
        caller of the upstream surface finds the argument honoured."""
        P = means3D.shape[0]
        if extra_attrs.dim() != 2 or extra_attrs.shape[0] != P:
            raise RuntimeError("extra_attrs must be [P,E]")
        E = extra_attrs.shape[1]
        rs = self.raster_settings._replace(bg=torch.zeros_like(self.raster_settings.bg))
        planes = []
        for c0 in range(0, E, 3):
            chunk = extra_attrs[:, c0:c0 + 3].to(torch.float32)
            if chunk.shape[1] < 3:
                chunk = torch.cat([chunk, chunk.new_zeros(P, 3 - chunk.shape[1])], dim=1)
            img = rasterize_gaussians(means3D, means2D, None, chunk.contiguous(), opacities, scales, rotations,
                                      cov3Ds_precomp, viewmatrix, rs, None, self.state)[0]
            planes.append(img[:min(3, E - c0)])
        if not planes:
            return torch.empty(0, int(rs.image_height), int(rs.image_width), dtype=torch.float32, device=means3D.device)
        return torch.cat(planes, dim=0)
