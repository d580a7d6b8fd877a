"""Gaussian-sharded frame data parallelism: the 8-GPU train step without a parameter
all-reduce.

Plain frame-DP replicates the cloud, renders one camera per rank and all-reduces 75 floats per Gaussian (300 MB at
1 M Gaussians): over point-to-point xGMI that exchange costs about as much as the whole backward pass.  Here the
Gaussians -- parameters, Adam moments, gradients -- are SHARDED over the ranks and the cameras stay one per rank:

  owner stage     every rank runs the per-Gaussian half of the rasterizer (deformation + activations, projection, SH
                  colour) on ITS slice of the cloud for the cameras of ALL ranks: same work as one camera over the
                  whole cloud, and it leaves one 64-byte splat record per (camera, Gaussian);
  all-to-all #1   records go to the rank that renders the camera (56 MB in, 56 MB out per GPU at 1 M / 8 ranks; an
                  all-to-all drives all seven xGMI links of a GPU at once);
  camera stage    tile binning, sort, compositing, loss and compositing backward over all records of the camera --
                  exactly the kernels of the single-GPU path -- leaving one 64-byte gradient row per Gaussian;
  all-to-all #2   gradient rows go back to the owners;
  owner stage     per-Gaussian backward for every camera, gradients summed over cameras locally, fused Adam on the
                  slice.  Only the MLP + camera-pose bucket (~0.1 MB) is all-reduced.

The two row formats and the *_views entry points are described in include/rodygs_hip.h.  Every collective is an
equal-split ``all_to_all_single`` / ``all_reduce`` of ``torch.distributed`` (RCCL on the GPUs, gloo in the CPU
tests of the exchange pattern); ``run_virtual_step`` drives several ranks inside one process for the single-GPU
parity tests, ``HostStagedExchange`` lets real processes share one GPU over gloo.  Densification runs per slice
(``densify``), and the config-5 loss set (depth, motion regularisers, rigidity) is supported (``full_losses``).  The reference has no distributed code: this is the build's own capability.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib, rasterizer
from .deform import MLPBasisNetwork, _birth_order
from .dp import FlatParams, frame_for
from .rasterizer import GaussianRasterizationSettings, _c_settings

_ROW = 16          # floats per splat record / gradient row (64 bytes)


class DistExchange:
    """The collectives of a sharded step over the default process group: the two equal-split all-to-alls and the small
    all-reduce of every step, plus the all-gather / reduce-scatter pair of a rigidity step."""

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        dist.all_to_all_single(recv, send)

    def all_to_all_rows(self, recv: torch.Tensor, send: torch.Tensor, stride: int, r0: int, r1: int) -> None:
        """Rows [r0, r1) of every shard of the equal-split all-to-all (shard = ``stride`` rows of 64 B): the row range of
        my slice under camera v goes to rank v, into the same rows of my shard there.  One grouped send / receive."""
        world, me = dist.get_world_size(), dist.get_rank()
        sv, rv = send.view(world, stride, -1), recv.view(world, stride, -1)
        ops = []
        for p in range(world):
            if p != me:                 # grouped point-to-point (RCCL fuses the batch into one launch; works on gloo too)
                ops.append(dist.P2POp(dist.isend, sv[p, r0:r1], p))
                ops.append(dist.P2POp(dist.irecv, rv[p, r0:r1], p))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        rv[me, r0:r1].copy_(sv[me, r0:r1])
        for q in reqs:
            q.wait()

    def all_reduce(self, t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def all_gather(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        dist.all_gather_into_tensor(out, mine)

    def reduce_scatter(self, out: torch.Tensor, full: torch.Tensor) -> None:
        dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM)


class HostStagedExchange(DistExchange):
    """The same five collectives staged through host memory, for process groups whose backend cannot move device
    tensors (gloo): lets the real multi-process ``train_step`` run with several ranks sharing ONE GPU
    (tests/test_gpu_parity.py) -- a debugging / test transport, not a performance path."""

    def all_to_all(self, recv, send):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu())
        recv.copy_(r)

    def all_to_all_rows(self, recv, send, stride, r0, r1):
        world = dist.get_world_size()
        sv, rv = send.view(world, stride, -1), recv.view(world, stride, -1)
        r = torch.empty(world, r1 - r0, sv.shape[2], dtype=recv.dtype)
        dist.all_to_all_single(r, sv[:, r0:r1].contiguous().cpu())
        rv[:, r0:r1].copy_(r)

    def all_reduce(self, t):
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        t.copy_(c)

    def all_gather(self, out, mine):
        parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine.cpu())
        out.copy_(torch.cat(parts).view(out.shape))

    def reduce_scatter(self, out, full):
        c = full.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n = out.shape[0]
        out.copy_(c[dist.get_rank() * n:(dist.get_rank() + 1) * n])


def shard_rows(P: int, world: int):
    """(per, stride): Gaussians per rank (the last rank may hold fewer) and the 256-aligned row stride of a shard in
    the exchanged buffers."""
    per = (P + world - 1) // world
    return per, (per + 255) // 256 * 256


def chunk_ranges(stride: int, chunks: int):
    """Row ranges [r0, r1) (multiples of 256, the same on every rank) that cut a shard of ``stride`` rows into at most
    ``chunks`` pieces (the pipelined all-to-all #1).  An empty shard (P = 0: stride 0) is ONE empty range, so that the
    sequence of collectives stays the same on every rank."""
    blocks = stride // 256
    if blocks == 0:
        return [(0, 0)]
    per = -(-blocks // max(1, min(chunks, blocks)))
    return [(b * 256, min(b + per, blocks) * 256) for b in range(0, blocks, per)]


class ShardedDynamicScene:
    """One rank's slice of a ``trainstep.DynamicScene`` (same parameters, same step arithmetic, gradients summed over
    the cameras of all ranks before Adam) -- build it with ``from_replica``."""

    @classmethod
    def from_replica(cls, ds, rank: int, world: int, exchange=None) -> "ShardedDynamicScene":
        return cls(ds, rank, world, exchange)

    def __init__(self, ds, rank: int, world: int, exchange=None):
        from .trainstep import _MLP_SINK_ORDER, bind_module_to_flat
        if not 1 <= world <= 16:
            raise ValueError("world must be 1..16 (RDG_MAX_VIEWS)")
        # full_losses: the config-5 loss set of trainstep.DynamicScene (photometric + Pearson depth on the camera rank,
        # motion L1 / sparsity on the owner's slice, basis regulariser on the replicated table, rigidity every 5th step
        # on a gathered copy of the cloud)
        self.full_losses = bool(ds.full_losses)
        if self.full_losses:
            self.loss_terms, self.depth_terms, self.rigidity = ds.loss_terms, ds.depth_terms, ds.rigidity
            if "coeff" in self.rigidity[2].mode:
                raise NotImplementedError("RigidityLoss mode 'coeff' needs the DC colours of the whole cloud")
            self.gt_depth = ds.gt_depth
        self._rng_state = None
        T = ds.T
        L = _lib.lib()
        if not L.rdg_dyn_getter_views_supported(16, T, world):
            raise NotImplementedError(f"motion table of {T} birth times x {world} views does not fit the fused getter")
        dev = ds.device
        self.device, self.rank, self.world = dev, rank, world
        self.ex = exchange if exchange is not None else DistExchange()
        self.H, self.W, self.T, self.sh_degree = ds.H, ds.W, T, ds.sh_degree
        self.tanfovx, self.tanfovy = ds.tanfovx, ds.tanfovy
        self.spatial_lr_scale = ds.spatial_lr_scale
        self.P_total = ds.P
        per, stride = shard_rows(ds.P, world)
        lo, hi = min(rank * per, ds.P), min((rank + 1) * per, ds.P)
        n = hi - lo
        self.lo, self.n, self.per = lo, n, per
        # ---- my slice of the parameters (+ fresh Adam moments copied from the replica's)
        spec = {k: ((n, *ds.fp.shapes[k][1:]), ds.fp.lr[k]) for k in ds.fp.names}
        fp = FlatParams(spec, dev)
        with torch.no_grad():
            for k in ds.fp.names:
                fp[k].copy_(ds.fp[k][lo:hi])
                o, m = fp.offsets[k]
                so, _ = ds.fp.offsets[k]
                row = m // max(n, 1)
                for dst, src in ((fp.exp_avg, ds.fp.exp_avg), (fp.exp_avg_sq, ds.fp.exp_avg_sq)):
                    dst[o:o + m].copy_(src[so + lo * row:so + hi * row])
        fp.step_count = ds.fp.step_count
        self.fp, self.row_lr = fp, dict(ds.row_lr)
        self.K = ds.fp.shapes["features"][1]
        self.time_ind = ds.time_ind[lo:hi].to(torch.int64).contiguous()
        # ---- replicated small bucket: deformation MLP + camera poses
        self.net = MLPBasisNetwork(128, 16, 26, False).to(dev)
        self.sp = bind_module_to_flat(self.net, 0.0016, dev, {"cam_q": ((T, 4), 1e-5), "cam_t": ((T, 3), 1e-6)})
        with torch.no_grad():
            self.sp.flat.copy_(ds.sp.flat)
            self.sp.exp_avg.copy_(ds.sp.exp_avg)
            self.sp.exp_avg_sq.copy_(ds.sp.exp_avg_sq)
        self.net.grad_sinks = [self.sp[k].grad for k in _MLP_SINK_ORDER]
        self.time_batch_embeddings = ds.time_batch_embeddings
        self.frame_embeddings = ds.frame_embeddings
        self.proj_t, self.bg = ds.proj_t, ds.bg
        self.gt = ds.gt
        self._rows_cache = {}
        if isinstance(self.ex, DistExchange) and world > 1:
            # the small bucket is REPLICATED state: refuse to start from replicas that differ (e.g. a model initialised
            # from an unseeded random stream on every rank)
            chk = self.sp.flat.double().sum().reshape(1)
            allc = torch.empty(world, dtype=torch.float64, device=dev)
            self.ex.all_gather(allc, chk)
            if float(allc.max() - allc.min()) != 0.0:
                raise RuntimeError("ShardedDynamicScene: the replicated MLP / pose parameters differ between ranks "
                                   f"(checksums {allc.tolist()}); build the replica with the same seed on every rank")
        # all-to-all #1 in this many row chunks, each overlapped with the projection of the next (1 = one exchange after
        # the whole owner stage)
        self.a2a_chunks = max(1, int(os.environ.get("RDG_A2A_CHUNKS", "1")))
        self.stats = None                # DensifyStats of my slice once track_densification() is called
        self.counts = [min(per, max(ds.P - r * per, 0)) for r in range(world)]      # slice sizes of all ranks
        self._time_ind_full = None
        self._alloc(stride)
        self.frames: List[int] = []

    def _alloc(self, stride: int) -> None:
        """Persistent buffers for ``self.n`` Gaussians of mine and a shard row stride of ``stride`` (the same on every
        rank): owner side = my Gaussians x all cameras, camera side = all Gaussians x my camera."""
        L, dev, Wn, n, T = _lib.lib(), self.device, self.world, self.n, self.T
        self.stride, self.rows = stride, Wn * stride
        u8 = dict(dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        R = self.rows
        with torch.cuda.device(dev):
            self.geom_own = torch.zeros(L.rdg_geom_bytes(R), **u8)      # records of unused rows stay zero = invisible
            self.geom_cam = torch.zeros(L.rdg_geom_bytes(R), **u8)
            self.grad_own = torch.zeros(L.rdg_grad_bytes(R), **u8)
            self.grad_cam = torch.zeros(L.rdg_grad_bytes(R), **u8)
            self.radii_own = torch.zeros(R, dtype=torch.int32, device=dev)
            self.radii_cam = torch.zeros(R, dtype=torch.int32, device=dev)
            self.nren = torch.zeros(2, dtype=torch.int32, device=dev)   # [0] = D, [1] = largest tile list
            self.rec_own = self.geom_own[:R * 64].view(torch.float32)
            self.rec_cam = self.geom_cam[:R * 64].view(torch.float32)
            self.row_own = self.grad_own[:R * 64].view(torch.float32)
            self.row_cam = self.grad_cam[:R * 64].view(torch.float32)
            # activated + deformed Gaussians of every camera, and the gradients coming back for them
            self.m3, self.g_m3 = torch.zeros(Wn, stride, 3, **f32), torch.zeros(Wn, stride, 3, **f32)
            self.ro, self.g_ro = torch.zeros(Wn, stride, 4, **f32), torch.zeros(Wn, stride, 4, **f32)
            self.sc, self.g_sc = torch.zeros(max(n, 1), 3, **f32), torch.zeros(Wn, stride, 3, **f32)   # time-independent
            self.op, self.g_op = torch.zeros(max(n, 1), 1, **f32), torch.zeros(Wn, stride, 1, **f32)
            self.d_m2 = torch.zeros(Wn, stride, 3, **f32)
            self.views = torch.zeros(Wn, 16, **f32)
            self.d_views = torch.zeros(Wn, 16, **f32)
            self.d_bases = torch.zeros(Wn, T + 1, 16, 7, **f32)
            self.sorted_ws = torch.empty(L.rdg_deform_sorted_views_ws_bytes(max(n, 1), Wn), **u8)
            if getattr(self, "image_ws", None) is None:
                self.image_ws = torch.empty(L.rdg_image_bytes(self.H, self.W), **u8)
                self.color = torch.empty(3, self.H, self.W, **f32)
                self.depth = torch.empty(1, self.H, self.W, **f32)
                self.normal = torch.empty(3, self.H, self.W, **f32)
                self.alpha = torch.empty(1, self.H, self.W, **f32)
                self.d_img = torch.empty(3, self.H, self.W, **f32)
                self.loss_ws = torch.empty(L.rdg_loss_ws_bytes(3, self.H, self.W), **u8)
                self.loss3 = torch.zeros(3, **f32)
        self._binning = None
        self._capacity = 0
        rs = GaussianRasterizationSettings(self.H, self.W, self.tanfovx, self.tanfovy, self.bg, 1.0, self.proj_t,
                                           self.sh_degree, False, False, True, True)
        self._rs, self._cs_rows = rs, {}                # settings of a row range of the owner stage, by row count
        self.cs_own = _c_settings(rs, n, self.K)        # per-Gaussian stages: my n Gaussians (per camera)
        self.cs_cam = _c_settings(rs, R, self.K)        # compositing stages: every row of the gathered records
        self.key = (R, self.H, self.W)

    # ---- the four local phases of a step; the collectives sit between them -------------------------------------------
    def _emb_rows(self, frames: Sequence[int]) -> torch.Tensor:
        key = tuple(frames)
        r = self._rows_cache.get(key)
        if r is None:
            if len(self._rows_cache) > 512:
                self._rows_cache.clear()
            idx = torch.tensor(list(frames), dtype=torch.int64, device=self.device)
            r = torch.cat([self.time_batch_embeddings, self.frame_embeddings[idx]], dim=0).contiguous()
            self._rows_cache[key] = r
        return r

    def chunk_ranges(self, chunks: int):
        return chunk_ranges(self.stride, chunks)

    def phase_owner_forward(self, step: int, perm: Sequence[int], chunks: int = 1, exchange: bool = False) -> None:
        """Deformation + activations + projection of MY Gaussians for the cameras of every rank -> ``rec_own``.
        ``chunks`` > 1: the slice is processed in row ranges (same arithmetic, same bits); with ``exchange`` the records of
        a range go on the wire (all-to-all #1, on a side stream) while the next range is projected -- on return
        ``rec_cam`` is complete on the current stream."""
        if chunks > 1 or exchange:
            return self._owner_forward_chunked(step, perm, chunks, exchange)
        L, Wn, T, n, dev = _lib.lib(), self.world, self.T, self.n, self.device
        self.frames = [frame_for(step, r, Wn, perm) for r in range(Wn)]
        self._frames_c = (C.c_int32 * Wn)(*self.frames)
        fp = self.fp
        with torch.cuda.device(dev):
            st = _lib.stream_ptr()
            # ONE pass of the MLP over the T birth-time rows + the W frame times
            allb = self.net.motion_basis(self._emb_rows(self.frames))                   # [T+W,16,7]
            self._allb = allb
            self._bases_all = torch.cat([allb[:T].unsqueeze(0).expand(Wn, -1, -1, -1), allb[T:].unsqueeze(1)], dim=1)
            b = self._bases_all.detach()
            if n == 0:                  # an empty slice (fewer Gaussians than ranks): its record rows stay zero
                return
            # my Gaussians at the times of all W cameras: parameters read once, one launch
            _lib.check(L.rdg_dyn_getter_views_forward(n, T, Wn, self.stride, _lib.ptr(fp["motion_coeff"]),
                                                      _lib.ptr(self.time_ind), _lib.ptr(b), float(self.spatial_lr_scale),
                                                      _lib.ptr(fp["xyz"]), _lib.ptr(fp["scaling"]),
                                                      _lib.ptr(fp["rotation"]), _lib.ptr(fp["opacity"]),
                                                      _lib.ptr(self.m3), _lib.ptr(self.sc), _lib.ptr(self.ro),
                                                      _lib.ptr(self.op), st), "rdg_dyn_getter_views_forward")
            sp = self.sp
            _lib.check(L.rdg_pose_views_forward(T, Wn, self._frames_c, _lib.ptr(sp["cam_q"]), _lib.ptr(sp["cam_t"]),
                                                _lib.ptr(self.views), st), "rdg_pose_views_forward")
            _lib.check(L.rdg_preprocess_forward_views(C.byref(self.cs_own), Wn, self.stride, _lib.ptr(self.m3),
                                                      _lib.ptr(fp["features"]), _lib.ptr(self.op), _lib.ptr(self.sc),
                                                      _lib.ptr(self.ro), _lib.ptr(self.views), _lib.ptr(self.proj_t),
                                                      _lib.ptr(self.geom_own), _lib.ptr(self.radii_own), st),
                       "rdg_preprocess_forward_views")

    def _owner_forward_chunked(self, step: int, perm: Sequence[int], chunks: int, exchange: bool) -> None:
        L, Wn, T, n, dev = _lib.lib(), self.world, self.T, self.n, self.device
        self.frames = [frame_for(step, r, Wn, perm) for r in range(Wn)]
        self._frames_c = (C.c_int32 * Wn)(*self.frames)
        fp, sp = self.fp, self.sp
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream()
            st = _lib.stream_ptr()
            allb = self.net.motion_basis(self._emb_rows(self.frames))                   # [T+W,16,7]
            self._allb = allb
            self._bases_all = torch.cat([allb[:T].unsqueeze(0).expand(Wn, -1, -1, -1), allb[T:].unsqueeze(1)], dim=1)
            b = self._bases_all.detach()
            _lib.check(L.rdg_pose_views_forward(T, Wn, self._frames_c, _lib.ptr(sp["cam_q"]), _lib.ptr(sp["cam_t"]),
                                                _lib.ptr(self.views), st), "rdg_pose_views_forward")
            if exchange and getattr(self, "_comm_stream", None) is None:
                self._comm_stream = torch.cuda.Stream(device=dev)
            for r0, r1 in self.chunk_ranges(chunks):
                m = min(r1, n) - r0                     # my rows in this range (the rest of the range is zero padding)
                if m > 0:
                    _lib.check(L.rdg_dyn_getter_views_forward(m, T, Wn, self.stride, _lib.ptr(fp["motion_coeff"][r0:]),
                                                              _lib.ptr(self.time_ind[r0:]), _lib.ptr(b),
                                                              float(self.spatial_lr_scale), _lib.ptr(fp["xyz"][r0:]),
                                                              _lib.ptr(fp["scaling"][r0:]), _lib.ptr(fp["rotation"][r0:]),
                                                              _lib.ptr(fp["opacity"][r0:]), _lib.ptr(self.m3[0, r0:]),
                                                              _lib.ptr(self.sc[r0:]), _lib.ptr(self.ro[0, r0:]),
                                                              _lib.ptr(self.op[r0:]), st), "rdg_dyn_getter_views_forward")
                    cs = self._cs_rows.get(m)
                    if cs is None:
                        cs = self._cs_rows[m] = _c_settings(self._rs, m, self.K)
                    _lib.check(L.rdg_preprocess_forward_views_rows(C.byref(cs), Wn, self.stride, r0, _lib.ptr(self.m3),
                                                                   _lib.ptr(fp["features"]), _lib.ptr(self.op),
                                                                   _lib.ptr(self.sc), _lib.ptr(self.ro),
                                                                   _lib.ptr(self.views), _lib.ptr(self.proj_t),
                                                                   _lib.ptr(self.geom_own), _lib.ptr(self.radii_own), st),
                               "rdg_preprocess_forward_views_rows")
                if exchange:
                    ready = torch.cuda.Event()
                    ready.record(main)
                    with torch.cuda.stream(self._comm_stream):
                        self._comm_stream.wait_event(ready)
                        self.ex.all_to_all_rows(self.rec_cam, self.rec_own, self.stride, r0, r1)
            if exchange:
                done = torch.cuda.Event()
                done.record(self._comm_stream)
                main.wait_event(done)

    def _composite_forward(self) -> None:
        L, dev = _lib.lib(), self.device
        n_tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        hint = rasterizer._CAPACITY_HINT
        # every step is another camera: 1.5x the last instance count, and the workspace only ever grows
        cap = max(int(hint.get(self.key, 0) * 1.5) + 4096, 4 * self.P_total + 4096, self._capacity)
        deferred = rasterizer.DEFAULT_STATE.mode("deferred_overflow_check") and self.key in hint
        if deferred:
            rasterizer.poll_overflow(block=False)
        st = _lib.stream_ptr()
        host = None
        if deferred:                # the binning stage mirrors (D, largest list) into pinned memory: rasterizer.poll_overflow
            host = rasterizer.DEFAULT_STATE.pinned_slot()
            host[0], host[1] = -1, -1
        self.cs_cam.num_rendered_host = None if host is None else host.data_ptr()
        while True:
            if self._binning is None or self._capacity != cap:
                self._binning = torch.empty(L.rdg_binning_bytes(cap, n_tiles), dtype=torch.uint8, device=dev)
                self._capacity = cap
            self.cs_cam.num_rendered_stats = 1
            # RDG_BIN_MODE=radix / bucket decides here as on the single-GPU path (note_largest_tile honours "bucket")
            self.cs_cam.bin_mode = 1 if rasterizer.DEFAULT_STATE.mode("force_radix") else int(rasterizer._BIN_HINT.get(self.key, 0))
            _lib.check(L.rdg_composite_forward(C.byref(self.cs_cam), _lib.ptr(self.bg), _lib.ptr(self.geom_cam),
                                               _lib.ptr(self.radii_cam), _lib.ptr(self._binning), cap,
                                               _lib.ptr(self.image_ws), _lib.ptr(self.nren), _lib.ptr(self.color),
                                               _lib.ptr(self.depth), _lib.ptr(self.normal), _lib.ptr(self.alpha), st),
                       "rdg_composite_forward")
            if deferred:
                rasterizer._PENDING.append((torch.cuda.current_stream(dev), host, self.key, cap, self.nren))
                break
            D, largest = (int(v) for v in self.nren.tolist())
            hint[self.key] = D
            rasterizer._note_largest_tile(self.key, largest, D)
            if D <= cap:
                break
            cap = int(D * 1.5) + 4096
        rasterizer.DEFAULT_STATE.last_image = (self.image_ws, self.H, self.W)

    # private random stream of this rank (Pearson boxes, rigidity sample): lets several virtual ranks in one process draw
    # exactly what they would draw in their own processes
    def seed_rng(self, seed: int) -> None:
        keep = (torch.get_rng_state(), torch.cuda.get_rng_state(self.device))
        torch.manual_seed(seed)
        self._rng_state = (torch.get_rng_state(), torch.cuda.get_rng_state(self.device))
        torch.set_rng_state(keep[0])
        torch.cuda.set_rng_state(keep[1], self.device)

    class _Rng:
        def __init__(self, scene):
            self.s = scene

        def __enter__(self):
            s = self.s
            if s._rng_state is not None:
                self.keep = (torch.get_rng_state(), torch.cuda.get_rng_state(s.device))
                torch.set_rng_state(s._rng_state[0])
                torch.cuda.set_rng_state(s._rng_state[1], s.device)

        def __exit__(self, *a):
            s = self.s
            if s._rng_state is not None:
                s._rng_state = (torch.get_rng_state(), torch.cuda.get_rng_state(s.device))
                torch.set_rng_state(self.keep[0])
                torch.cuda.set_rng_state(self.keep[1], s.device)

    def phase_camera(self) -> torch.Tensor:
        """Records of MY camera have arrived in ``rec_cam``: bin, composite, loss, compositing backward ->
        ``row_cam`` (one gradient row per Gaussian of the whole cloud).  Returns the camera's loss."""
        L, dev = _lib.lib(), self.device
        frame = self.frames[self.rank]
        with torch.cuda.device(dev):
            st = _lib.stream_ptr()
            _lib.check(L.rdg_geom_from_records(C.byref(self.cs_cam), _lib.ptr(self.geom_cam), _lib.ptr(self.radii_cam),
                                               _lib.ptr(self.nren), st), "rdg_geom_from_records")
            self._composite_forward()
            gt = self.gt[frame]
            _lib.check(L.rdg_photometric_loss_forward(3, self.H, self.W, _lib.ptr(self.color), _lib.ptr(gt), 0.2,
                                                      _lib.ptr(self.loss_ws), _lib.ptr(self.loss3), st),
                       "rdg_photometric_loss_forward")
            _lib.check(L.rdg_photometric_loss_backward(3, self.H, self.W, _lib.ptr(self.color), _lib.ptr(gt), 0.2,
                                                       _lib.ptr(self.loss_ws), None, _lib.ptr(self.d_img), st),
                       "rdg_photometric_loss_backward")
            g_depth, loss = None, self.loss3[0]
            if self.full_losses:
                with self._Rng(self):
                    depth_leaf = self.depth.detach().requires_grad_(True)
                    ld = sum(w * mod(depth_leaf, self.gt_depth[frame]) for w, mod in self.depth_terms)
                    ld.backward()
                g_depth, loss = depth_leaf.grad.contiguous(), loss + ld.detach()
            _lib.check(L.rdg_composite_backward(C.byref(self.cs_cam), _lib.ptr(self.bg), _lib.ptr(self.geom_cam),
                                                _lib.ptr(self._binning), self._capacity, _lib.ptr(self.image_ws),
                                                _lib.ptr(self.d_img), _lib.ptr(g_depth), None, None,
                                                _lib.ptr(self.grad_cam), st), "rdg_composite_backward")
        return loss

    def phase_owner_backward(self) -> None:
        """Gradient rows of MY Gaussians from every camera have arrived in ``row_own``: per-Gaussian backward over all
        cameras (two launches; every parameter gradient leaves its kernel already summed over the cameras, straight
        into the flat gradient bucket), MLP + pose gradients into the small bucket (partial
        sums over my slice -- the caller all-reduces ``sp.flat_grad``)."""
        L, Wn, T, n, dev, fp = _lib.lib(), self.world, self.T, self.n, self.device, self.fp
        with torch.cuda.device(dev):
            st = _lib.stream_ptr()
            self._loss_owner = None
            if n == 0:                  # empty slice: no parameter gradients; no contribution to bases or poses
                self.d_bases.zero_()
                self.d_views.zero_()
                return
            _lib.check(L.rdg_preprocess_backward_views(
                C.byref(self.cs_own), Wn, self.stride, _lib.ptr(self.m3), _lib.ptr(fp["features"]), _lib.ptr(self.op),
                _lib.ptr(self.sc), _lib.ptr(self.ro), _lib.ptr(self.views), _lib.ptr(self.proj_t),
                _lib.ptr(self.radii_own), _lib.ptr(self.geom_own), _lib.ptr(self.grad_own), _lib.ptr(self.g_m3),
                _lib.ptr(self.d_m2), _lib.ptr(fp["features"].grad), _lib.ptr(self.g_op), _lib.ptr(self.g_sc), _lib.ptr(self.g_ro),
                _lib.ptr(self.d_views), st), "rdg_preprocess_backward_views")
            if self.stats is not None and n:
                # add_densification_stats (rodygs.py:319-341) for each camera of the step: the screen-space gradient norm
                # and the radius of my Gaussians where they were visible -- all local, no collective
                vis = self.radii_own.view(Wn, self.stride)[:, :n] > 0
                g2 = torch.norm(self.d_m2[:, :n, :2], dim=-1)
                self.stats.xyz_gradient_accum += (g2 * vis).sum(0).unsqueeze(1)
                self.stats.denom += vis.sum(0).unsqueeze(1).to(self.stats.denom.dtype)
                rad = (self.radii_own.view(Wn, self.stride)[:, :n] * vis).max(0).values.to(self.stats.max_radii2D.dtype)
                self.stats.max_radii2D = torch.maximum(self.stats.max_radii2D, rad)
            b = self._bases_all.detach()
            if n:
                order, inv, _ = _birth_order(self.time_ind)
                # all cameras in one pass: the five parameter gradients come out summed over the cameras, straight
                # into the flat gradient bucket
                _lib.check(L.rdg_dyn_getter_views_backward(
                    n, T, Wn, self.stride, _lib.ptr(fp["motion_coeff"]), _lib.ptr(self.time_ind), _lib.ptr(b),
                    float(self.spatial_lr_scale), _lib.ptr(fp["scaling"]), _lib.ptr(fp["rotation"]),
                    _lib.ptr(fp["opacity"]), _lib.ptr(self.g_m3), _lib.ptr(self.g_sc), _lib.ptr(self.g_ro),
                    _lib.ptr(self.g_op), _lib.ptr(fp["xyz"].grad), _lib.ptr(fp["scaling"].grad),
                    _lib.ptr(fp["rotation"].grad), _lib.ptr(fp["opacity"].grad), _lib.ptr(fp["motion_coeff"].grad),
                    _lib.ptr(self.d_bases), _lib.ptr(order), _lib.ptr(inv), _lib.ptr(self.sorted_ws), st),
                    "rdg_dyn_getter_views_backward")
            else:
                self.d_bases.zero_()
            self._loss_owner = None
            if self.full_losses and n:
                # motion L1 / sparsity (means over ALL P x 16 coefficients): my slice's share, times the N cameras of
                # the step (every rank adds these terms in the replicated formulation); accumulates on top of the
                # getter's gradient
                from .motion_losses import fused_motion_l1_sparsity
                share = float(Wn) * n / float(self.P_total)
                lm = share * fused_motion_l1_sparsity(fp["motion_coeff"], self.loss_terms["motion_l1"][0],
                                                      self.loss_terms["motion_sparsity"][0],
                                                      grad_sink=fp["motion_coeff"].grad)
                lm.backward()
                self._loss_owner = lm.detach()

    def rigidity_due(self, step: int) -> bool:
        return self.full_losses and step % self.rigidity[1] == 0

    def pack_for_rigidity(self) -> torch.Tensor:
        """[per,19] = (xyz, motion coefficients) of my slice, zero rows after the first n: my part of the all-gather."""
        n = self.n
        buf = torch.zeros(self.per, 19, dtype=torch.float32, device=self.device)
        buf[:n, :3] = self.fp["xyz"].detach()
        buf[:n, 3:] = self.fp["motion_coeff"].detach().reshape(n, 16)
        return buf

    def pack_time_ind(self) -> torch.Tensor:
        buf = torch.zeros(self.per, dtype=torch.int64, device=self.device)
        buf[:self.n] = self.time_ind
        return buf

    def _valid_rows(self, padded: torch.Tensor) -> torch.Tensor:
        """[W*per, ...] gathered layout -> [P, ...] without the padding rows of short slices."""
        if all(c == self.per for c in self.counts):
            return padded
        return torch.cat([padded[r * self.per:r * self.per + c] for r, c in enumerate(self.counts)])

    def phase_small_backward(self, step: int, gathered: Optional[torch.Tensor] = None,
                             time_ind_gathered: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """What remains of backward after the per-Gaussian kernels: (on rigidity steps) RigidityLoss of MY camera's time
        on the gathered cloud, the basis regulariser, then ONE pass of the MLP backward for everything that reached the
        motion bases, and the pose backward.  Returns the [W*per,19] gradient of the gathered cloud for the
        reduce-scatter (rigidity steps) or None."""
        L, Wn, T, dev = _lib.lib(), self.world, self.T, self.device
        roots, root_grads, full_grad = [self._bases_all], [self.d_bases], None
        extra = None
        with torch.cuda.device(dev):
            st = _lib.stream_ptr()
            if self.full_losses:
                allb, scene = self._allb, self

                class _Model:                                    # what the reference's loss modules read
                    unique_times = list(range(T))

                    @staticmethod
                    def get_total_motion_table():
                        return allb[:T]

                    @staticmethod
                    def get_motion_for_times(timesteps, time_indices=None):
                        return allb[:T][time_indices.to(allb.device)]

                w, mod = self.loss_terms["motion_basis_reg"]
                extra = w * mod(_Model)
                leaves = None
                if gathered is not None:
                    from .deform import gaussian_deformation_packed
                    if time_ind_gathered is not None:
                        self._time_ind_full = self._valid_rows(time_ind_gathered).contiguous()
                    full = self._valid_rows(gathered)
                    xyz_full = full[:, :3].contiguous().requires_grad_(True)
                    coeff_full = full[:, 3:].contiguous().requires_grad_(True)
                    leaves = (xyz_full, coeff_full)
                    dxyz, _ = gaussian_deformation_packed(coeff_full, self._time_ind_full, self._bases_all[self.rank],
                                                          self.spatial_lr_scale)
                    _Model._xyz, _Model._motion_coeff = xyz_full, coeff_full.view(-1, 1, 16)
                    _Model._features_dc = torch.zeros(1, 1, 3, device=dev).expand(full.shape[0], 1, 3)
                    wr, _, rig = self.rigidity
                    with self._Rng(self):
                        extra = extra + wr * rig(_Model, dxyz)
                roots.append(extra)
                root_grads.append(None)
            torch.autograd.backward(roots, root_grads)      # MLP backward: overwrites its ten sinks in sp.flat_grad
            if self.full_losses and gathered is not None:
                g = torch.cat([leaves[0].grad, leaves[1].grad], dim=1)
                if all(c == self.per for c in self.counts):
                    full_grad = g.contiguous()
                else:
                    full_grad = torch.zeros(Wn * self.per, 19, dtype=torch.float32, device=dev)
                    o = 0
                    for r, c in enumerate(self.counts):
                        full_grad[r * self.per:r * self.per + c] = g[o:o + c]
                        o += c
            self._bases_all = self._allb = None
            sp = self.sp
            _lib.check(L.rdg_pose_views_backward(T, Wn, self._frames_c, _lib.ptr(sp["cam_q"]), _lib.ptr(sp["cam_t"]),
                                                 _lib.ptr(self.d_views), _lib.ptr(sp["cam_q"].grad),
                                                 _lib.ptr(sp["cam_t"].grad), st), "rdg_pose_views_backward")
        self._loss_small = None if extra is None else extra.detach()
        return full_grad

    def add_rigidity_grads(self, mine: torch.Tensor) -> None:
        """[per,19] = my rows of the rigidity gradient summed over the ranks (reduce-scatter result)."""
        n = self.n
        self.fp["xyz"].grad += mine[:n, :3]
        self.fp["motion_coeff"].grad.view(n, 16).add_(mine[:n, 3:])

    def phase_update(self) -> None:
        from .trainstep import fused_adam_
        fused_adam_(self.fp, row_lr=self.row_lr, extra=(self.sp,))

    # ---- one step over the process group ---------------------------------------------------------------------------
    def train_step(self, step: int, perm: Sequence[int]) -> torch.Tensor:
        if self.a2a_chunks > 1:         # records of a row range on the wire while the next range is projected
            self.phase_owner_forward(step, perm, chunks=self.a2a_chunks, exchange=True)
        else:
            self.phase_owner_forward(step, perm)
            self.ex.all_to_all(self.rec_cam, self.rec_own)
        loss = self.phase_camera()
        self.ex.all_to_all(self.row_own, self.row_cam)
        self.phase_owner_backward()
        gathered = tig = None
        if self.rigidity_due(step):
            gathered = torch.empty(self.world * self.per, 19, dtype=torch.float32, device=self.device)
            self.ex.all_gather(gathered, self.pack_for_rigidity())
            if self._time_ind_full is None:
                tig = torch.empty(self.world * self.per, dtype=torch.int64, device=self.device)
                self.ex.all_gather(tig, self.pack_time_ind())
        full_grad = self.phase_small_backward(step, gathered, tig)
        if full_grad is not None:
            mine = torch.empty(self.per, 19, dtype=torch.float32, device=self.device)
            self.ex.reduce_scatter(mine, full_grad)
            self.add_rigidity_grads(mine)
        self.ex.all_reduce(self.sp.flat_grad)
        self.phase_update()
        return self.step_loss(loss)

    def step_loss(self, camera_loss: torch.Tensor) -> torch.Tensor:
        """This rank's summand of the step loss: its camera's terms + the replicated terms + its slice's share of the
        per-Gaussian regularisers (the sum over ranks equals the sum over ranks of the replicated formulation)."""
        out = camera_loss.detach()
        for t in (getattr(self, "_loss_owner", None), getattr(self, "_loss_small", None)):
            if t is not None:
                out = out + t
        return out

    # ---- densification (rodygs.py:319-362 / rodygs_static.py:170-319) on the slices -------------------------------------
    def track_densification(self) -> None:
        from .densify import DensifyStats
        self.stats = DensifyStats.zeros(self.n, self.device)

    def densify_local(self, max_grad: float = 0.0002, min_opacity: float = 0.005, extent: Optional[float] = None,
                      max_screen_size=None, percent_dense: float = 0.01) -> dict:
        """densify_and_prune on MY slice only: every Gaussian has one owner, and its statistics were gathered there
        over all cameras, so no rank needs another rank's decisions.  Call ``reshard`` (collective) afterwards."""
        from .densify import densify_and_prune
        if self.stats is None:
            raise RuntimeError("call track_densification() first")
        res = densify_and_prune(self.fp, self.stats, {"time_ind": self.time_ind}, max_grad, min_opacity,
                                extent if extent is not None else self.spatial_lr_scale, max_screen_size, percent_dense)
        self.fp, self.stats, self.time_ind = res.fp, res.stats, res.per_point["time_ind"].contiguous()
        self.n = self.fp.shapes["xyz"][0]
        return {"n": self.n, "cloned": res.n_clone, "split": res.n_split, "pruned": res.n_pruned}

    def reshard(self, counts: Sequence[int]) -> None:
        """New common row stride after the slices changed size (slices stay where they are: a rank's Gaussians are
        rows [rank*stride, rank*stride + n) of the gathered buffers, whatever n the other ranks have)."""
        self.counts = [int(c) for c in counts]
        self.per, self.P_total = max(self.counts), sum(self.counts)
        self._time_ind_full = None
        self._alloc((self.per + 255) // 256 * 256)

    def densify(self, **kw) -> dict:
        info = self.densify_local(**kw)
        cnt = torch.tensor([self.n], dtype=torch.int64, device=self.device)
        counts = torch.empty(self.world, dtype=torch.int64, device=self.device)
        self.ex.all_gather(counts, cnt)
        self.reshard(counts.tolist())
        info["P"] = self.P_total
        return info

    # ---- inspection ---------------------------------------------------------------------------------------------------
    def visible_count(self) -> int:
        return int((self.radii_cam > 0).sum().item())

    def gather_flat(self) -> FlatParams:
        """The whole cloud as ONE FlatParams on every rank (parameters and both Adam moments, rank order); collective."""
        P = self.P_total
        spec = {k: ((P, *self.fp.shapes[k][1:]), self.fp.lr[k]) for k in self.fp.names}
        full = FlatParams(spec, self.device)
        full.step_count = self.fp.step_count
        with torch.no_grad():
            for k in self.fp.names:
                o, m = self.fp.offsets[k]
                fo, fm = full.offsets[k]
                shape = self.fp.shapes[k]
                for src, dst in ((self.fp.flat, full.flat), (self.fp.exp_avg, full.exp_avg),
                                 (self.fp.exp_avg_sq, full.exp_avg_sq)):
                    pad = torch.zeros(self.per, *shape[1:], dtype=torch.float32, device=self.device)
                    pad[:self.n] = src[o:o + m].view(shape)
                    got = torch.empty(self.world * self.per, *shape[1:], dtype=torch.float32, device=self.device)
                    self.ex.all_gather(got, pad)
                    dst[fo:fo + fm].copy_(self._valid_rows(got).reshape(-1))
        return full

    def export_state_dict(self, iteration: int) -> dict:
        """The reference's checkpoint dictionary (rodygs_amd.checkpoint.export_state_dict) of the gathered cloud, with the
        deformation network, the birth time of every Gaussian and the camera tables; collective, identical on all ranks."""
        from .checkpoint import export_state_dict
        full = self.gather_flat()
        ti = torch.zeros(self.per, dtype=torch.int64, device=self.device)
        ti[:self.n] = self.time_ind
        got = torch.empty(self.world * self.per, dtype=torch.int64, device=self.device)
        self.ex.all_gather(got, ti)
        times = (torch.arange(self.T, dtype=torch.float32, device=self.device) / self.T)[self._valid_rows(got)]
        return export_state_dict(full, iteration, self.sh_degree, self.spatial_lr_scale, deform_network=self.net,
                                 gaussian_to_time=times, cameras=(self.sp["cam_q"].detach(), self.sp["cam_t"].detach()),
                                 feature_lr_rest=self.row_lr["features"][2])

    def gather_params(self) -> Optional[dict]:
        """{name: full [P,...] tensor} assembled from all ranks in rank order (checkpointing, tests); collective."""
        counts, width = self.counts, self.per
        out = {}
        for k in self.fp.names:
            mine = self.fp[k].detach()
            pad = torch.zeros(width, *mine.shape[1:], dtype=mine.dtype, device=mine.device)
            pad[:self.n] = mine
            full = torch.empty(self.world * width, *mine.shape[1:], dtype=mine.dtype, device=mine.device)
            self.ex.all_gather(full, pad)
            out[k] = self._valid_rows(full)
        return out


def run_virtual_densify(scenes: Sequence["ShardedDynamicScene"], **kw) -> List[dict]:
    """``densify`` for virtual ranks in one process (the two scalar all-reduces done by hand)."""
    infos = [s.densify_local(**kw) for s in scenes]
    counts = [s.n for s in scenes]
    for s, info in zip(scenes, infos):
        s.reshard(counts)
        info["P"] = sum(counts)
    return infos


def run_virtual_step(scenes: Sequence[ShardedDynamicScene], step: int, perm: Sequence[int]) -> List[torch.Tensor]:
    """All ranks of a sharded step inside ONE process (same device, same stream): the exchanges become block copies.
    Used by the single-GPU parity tests; the arithmetic per rank is exactly that of ``train_step``."""
    Wn = len(scenes)
    for s in scenes:
        s.phase_owner_forward(step, perm)
    for c in range(Wn):
        for s in range(Wn):
            scenes[c].rec_cam.view(Wn, -1)[s].copy_(scenes[s].rec_own.view(Wn, -1)[c])
    losses = [s.phase_camera() for s in scenes]
    for s in range(Wn):
        for c in range(Wn):
            scenes[s].row_own.view(Wn, -1)[c].copy_(scenes[c].row_cam.view(Wn, -1)[s])
    for s in scenes:
        s.phase_owner_backward()
    gathered = tig = None
    if scenes[0].rigidity_due(step):
        gathered = torch.cat([s.pack_for_rigidity() for s in scenes])
        if scenes[0]._time_ind_full is None:
            tig = torch.cat([s.pack_time_ind() for s in scenes])
    full_grads = [s.phase_small_backward(step, gathered, tig) for s in scenes]
    if full_grads[0] is not None:
        total = torch.stack(full_grads).sum(0)
        per = scenes[0].per
        for r, s in enumerate(scenes):
            s.add_rigidity_grads(total[r * per:(r + 1) * per])
    total = torch.stack([s.sp.flat_grad for s in scenes]).sum(0)
    for s in scenes:
        s.sp.flat_grad.copy_(total)
        s.phase_update()
    return [s.step_loss(x).clone() for s, x in zip(scenes, losses)]
