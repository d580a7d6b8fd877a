"""bench.py -- the driver's measurement contract for the rodygs_amd hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): 1 M dynamic Gaussians + time-deformation
MLP, 1920x1080, SH degree 3, 100-frame synthetic video (SURVEY.md §8d generator, seed 777).  A "step" is one
full train step on one camera per GPU: deformation -> rasterize forward -> 0.8 L1 + 0.2 D-SSIM -> backward ->
fused Adam over every parameter.  Inputs are resident in HBM before the timed region.  Weak scaling: every GPU
renders its own frame, value = frames (train steps x GPUs) per second.  N > 1 (--dp-mode): "allreduce" is the
BASELINE north_star formulation -- replicated cloud, RCCL all-reduce of the flat gradient bucket, overlapped with
backward and Adam; "shard" keeps the Gaussians sharded over the ranks and exchanges 64-byte splat records / gradient
rows with two all-to-alls per step (rodygs_amd/sharded.py); the default "both" times exactly K steps of each, back to
back, prints the faster one as `value` (its name in config.parallelism) and both under `dp_modes`.

The JSON line also carries
  roofline     : the dominant kernel (render backward) -- ALGORITHMIC bytes per launch / its average duration,
                 measured live with hipEvents recorded by the library on the launch stream;
  cpu_baseline : the PyTorch-CPU oracle timed on this box's host cores on a bounded sample of the same frame.
"""
import argparse
import json
import os
import sys
import time

# --loop: the cloud grows by a few percent at every densification and with it every per-Gaussian tensor of the step; with the
# allocator's sizes rounded up to sixteenths of a power of two the blocks of the old cloud serve the new one (otherwise the first
# step after every densification allocates ~30 new blocks from the driver: 6 ms).  Read when the allocator initialises.
if "--loop" in sys.argv:
    os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "roundup_power2_divisions:16")
    os.environ.setdefault("PYTORCH_CUDA_ALLOC_CONF", os.environ["PYTORCH_HIP_ALLOC_CONF"])

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SQ_FILE = "r06_pmc_sq_counters.json"     # rocprofv3 --pmc SQ_* pass of this bench command (scripts/pmc_summary.py)
N_SIMD = 1024                            # 256 CUs x 4 SIMDs
# measured issue cost of a wave64 VALU instruction with every SIMD saturated (scripts/valu_probe.hip,
# profiles/r02_valu_issue_cost.txt), in cycles per instruction and SIMD
VALU_ISSUE_CLASSES = {"f32 mul/add/fma/mov with VGPR sources": 2.3, "every other VALU instruction": 4.2,
                      "transcendental": 8.4}
PMC_FILE = "r06_pmc_hbm_traffic.json"   # rocprofv3 --pmc passes of this bench command (scripts/pmc_hbm_traffic.py)


def kernel_source_hash(files=("rdg_render.hip", "rdg_common.h")):
    """sha256 over the sources of the dominant kernel: stamps a PMC file (scripts/pmc_hbm_traffic.py) and decides whether
    the traffic figure in it still belongs to the kernel that runs."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "rodygs_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def metric_label(P, W, H):
    """The metric string from the arguments: BASELINE.json's wording at its own shape, the actual shape otherwise."""
    pts = f"{P // 1000000}M" if P % 1000000 == 0 else (f"{P // 1000}k" if P % 1000 == 0 else str(P))
    res = {(1920, 1080): "1080p", (3840, 2160): "4K"}.get((W, H), f"{W}x{H}")
    return f"train-step fps at {pts} dynamic Gaussians / {res} (fwd+bwd+Adam, one camera per GPU per step)"


def workload_label(P, W, H, frames, full_losses, world, variant="uniform"):
    """Name the workload from the arguments (BASELINE.json configs by shape), never from a hard-coded string."""
    base = f"{P} dynamic Gaussians + deformation MLP, {W}x{H}, SH3, {frames}-frame synthetic video"
    if variant != "uniform":
        return base + f" (scene variant '{variant}': not a BASELINE config)"
    if (P, W, H) == (1000000, 1920, 1080) and not full_losses:
        return base + (" (BASELINE configs[2])" if world == 1 else f" (BASELINE configs[3] shape on {world} GPUs)")
    if (P, W, H) == (4000000, 3840, 2160) and full_losses:
        return base + " + depth and motion-regularisation losses (BASELINE configs[4] shape)"
    if (P, W, H) == (100000, 1920, 1080):
        return base + " (BASELINE configs[1] size, run as a dynamic train step)"
    return base + " (not a BASELINE config)"


def wire_bytes(P, K, world, sharded, frames=100, mlp_params=68656):
    """Payload each GPU puts on the wire per step (what the collectives are asked to move, before the algorithm's own
    factor: a ring all-reduce sends 2 (N - 1) / N of its payload per GPU, an all-to-all (N - 1) / N):
      allreduce: the flat gradient bucket of the replicated cloud -- (11 + 3 K + 16) floats per Gaussian -- plus the MLP
                 and camera-pose bucket;
      shard:     two all-to-alls of 64-byte rows (splat records out, gradient rows back) over the P / N Gaussians a rank
                 owns x N cameras, plus the all-reduce of the MLP + pose bucket."""
    small = (mlp_params + frames * 7) * 4
    if not sharded:
        payload = P * (11 + 3 * K + 16) * 4 + small
        return {"all_reduce_payload_bytes": payload, "ring_bytes_sent_per_gpu": int(payload * 2 * (world - 1) / world)}
    per = -(-P // world)
    stride = (per + 255) // 256 * 256
    a2a = stride * world * 64                       # one all-to-all's send buffer on a rank (all cameras of its slice)
    return {"all_to_all_payload_bytes": 2 * a2a, "all_to_all_bytes_sent_per_gpu": int(2 * a2a * (world - 1) / world),
            "all_reduce_payload_bytes": small}


def _spread_order(n):
    """0 .. n-1 ordered by the base-2 radical inverse (bit-reversed counting, also for n that is no power of two)."""
    def rinv(i):
        r, f = 0.0, 0.5
        while i:
            r += f * (i & 1)
            i >>= 1
            f *= 0.5
        return r
    return sorted(range(n), key=rinv)


def stage_bytes(P, K, V, D, H, W, world=1, sharded=False, sh_adam_in_backward=False, radix_binning=False,
                densify_stats=False):
    """ALGORITHMIC HBM bytes per stage and step (SURVEY.md §8d formulas, adjusted to the algorithm that actually runs
    -- DESIGN.md §4 'Roofline accounting'):
      * binning: bucket binning moves, per tile instance, a 4-B rank (written by the count pass, read by the scatter
        pass), an 8-B (depth, id) composite (written by the scatter, read by the per-tile sort) and the 4-B sorted id
        = 28 B, plus two passes over 40 B of per-Gaussian geometry and the per-tile counters / ranges; the radix
        path (depth first: the Gaussians sorted once, their instances partitioned by tile id) is priced by its own passes;
      * optimizer in backward: the SH features' Adam step happens inside the per-Gaussian backward kernel -- their
        28 B/float leave the Adam launch; the kernel no longer writes dL/dshs (-12K B) and instead reads and writes the
        parameter and both moments (24 B/float) of those 3K floats;
      * densification statistics: the per-Gaussian backward reads and writes max_radii2D, xyz_gradient_accum and denom of
        every visible Gaussian (24 B);
      * deformation: priced as the fused getter that runs (deformation + activations), 160 B forward / 256 B backward per
        Gaussian (round 5; VERDICT r03 / r04: the survey's 96 B understated what the stage has to move)."""
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    n_tile_pass = (max(tiles - 1, 1).bit_length() + 7) // 8      # 8-bit passes over the tile-id bits (two at 1080p and 4K)
    n_adam = 59 + 16                                    # floats per Gaussian: 11 geometry + 3K SH (K=16) + 16 coeff
    own = P // world if sharded else P                  # Gaussian-sharded frame-DP: a rank steps its slice only
    sb = {
        "preprocess": P * (44 + 12 * K) + V * 48 + P * 8,
        # radix binning, depth first: P 4-byte keys + ids sorted in four passes (20 B per pair and pass: 4 counted, 8 read,
        # 8 written), D (tile id, Gaussian id) pairs written once, partitioned in n_tile_pass passes, their ids read once
        "binning": (P * (8 + 20 * 4) + D * (8 + 20 * n_tile_pass + 4) + tiles * 8) if radix_binning
                   else (D * 28 + P * 80 + tiles * 16),
        "render_fwd": D * 44 + H * W * 40,
        "render_bwd": D * 44 + H * W * 40 + V * 40,
        "preprocess_bwd": P * (44 + 12 * K) * 2 + V * 48,
        # the fused getter (deformation + activations in one kernel each way, what the step runs): forward reads xyz 12 +
        # scaling 12 + rotation 16 + opacity 4 + coefficients 64 + birth index 8 and writes the four activated tensors (44);
        # backward reads their gradients (44), scaling / rotation / opacity (32), the birth index (8) and the coefficients
        # (64, for the basis gradient) and writes five parameter gradients (12 + 12 + 16 + 4 + 64).  (SURVEY 8d's 96 B each
        # way is the bare deformation without the activations; the birth-sorted copy the reduction reads is overhead.)
        "deform_fwd": P * 160, "deform_bwd": P * 256,
        "adam": 28 * n_adam * own,
    }
    if sh_adam_in_backward:
        sb["adam"] -= 28 * 3 * K * own
        sb["preprocess_bwd"] += (24 * 3 * K - 12 * K) * P
    if densify_stats:
        sb["preprocess_bwd"] += 24 * V
    return sb


def _host_cpu():
    """(logical cores, model name) of the box the cpu_baseline leg runs on."""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return os.cpu_count(), model


FRAME_KEYS = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")


def _cpu_train_step(frame, gt, sh_degree, budget_s):
    """One CPU train step built from the oracle (kind "port") on EXACTLY the rasterizer inputs in `frame` (FRAME_KEYS +
    H, W, tanfovx, tanfovy, projmatrix): raw parameters -> activations -> oracle rasterizer forward -> 0.8 L1 + 0.2 D-SSIM
    (torch) against `gt` -> backward -> torch.optim.Adam (eps 1e-15) over every Gaussian parameter.  The activations are
    computed and differentiated, but the rasterizer is fed `frame`'s tensors bit for bit (value = captured + (act - act
    .detach())): exp(log(s)) is not s to the last bit, and one ulp in a scale can move a radius -- the parity check below
    needs the two implementations on identical inputs.  The compositing runs over interleaved eighths of the tile grid
    until `budget_s` is spent; if tiles remain, their share is extrapolated by splat instances and the dict says so.
    Returns (timing dict, oracle results of the frame for the parity check -- None if tiles were left out)."""
    from oracle import rasterizer_oracle as O   # checker / baseline only -- never on the product path
    from rodygs_amd.losses import photometric_loss     # torch expression (host mirror pinned by golden G6)
    H, W = frame["H"], frame["W"]
    P = frame["means3D"].shape[0]
    st = O.OracleSettings(H, W, frame["tanfovx"], frame["tanfovy"], torch.zeros(3), 1.0, frame["projmatrix"], sh_degree)
    op0 = frame["opacities"].clamp(1e-6, 1 - 1e-6)
    params = {"xyz": frame["means3D"].clone(), "features": frame["shs"].clone(), "scaling": torch.log(frame["scales"]),
              "rotation": frame["rotations"].clone(), "opacity": torch.log(op0 / (1 - op0))}
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam([{"params": [v], "lr": 1e-3} for v in params.values()], lr=0.0, eps=1e-15)
    vm = frame["viewmatrix"].clone().requires_grad_(True)
    m2 = torch.zeros(P, 3, requires_grad=True)
    t0 = time.perf_counter()
    act = {"means3D": params["xyz"], "shs": params["features"], "opacities": torch.sigmoid(params["opacity"]),
           "scales": torch.exp(params["scaling"]), "rotations": torch.nn.functional.normalize(params["rotation"], dim=1)}
    x = {k: (frame[k] + (a - a.detach())) for k, a in act.items()}      # the frame's own bits, the activations' graph
    for v in x.values():
        v.retain_grad()
    geom = O.preprocess(x["means3D"], m2, x["opacities"], vm, st, shs=x["shs"], scales=x["scales"],
                        rotations=x["rotations"])
    binning = O.bin_and_sort(geom)
    t_pre = time.perf_counter() - t0
    gx, gy = geom["grid"]
    n_tiles = gx * gy
    ranges = binning["ranges"].astype("int64")
    total_pairs = int(binning["num_rendered"])
    color = torch.zeros(3, H, W)
    other = {"depth": torch.zeros(1, H, W), "alpha": torch.zeros(1, H, W)}
    final_T, n_contrib = torch.ones(H, W), torch.zeros(H, W, dtype=torch.int32)
    done_tiles, done_pairs = 0, 0
    t1 = time.perf_counter()
    for c in range(8):
        subset = list(range(c, n_tiles, 8))
        img = O.render_tiles(geom, binning, st.bg, H, W, tile_subset=subset)
        color = color + img["color"]
        with torch.no_grad():      # tiles outside the subset: zero images, T = 1, no contributor
            other = {k: v + img[k] for k, v in other.items()}
            final_T, n_contrib = torch.minimum(final_T, img["final_T"]), torch.maximum(n_contrib, img["n_contrib"])
        done_tiles += len(subset)
        done_pairs += int((ranges[subset, 1] - ranges[subset, 0]).sum())
        if time.perf_counter() - t1 > 0.4 * budget_s:     # backward costs about as much again
            break
    loss = photometric_loss(color, gt, 0.2)
    loss.backward()
    t_tiles = time.perf_counter() - t1
    t2 = time.perf_counter()
    opt.step()
    t_adam = time.perf_counter() - t2
    scale = total_pairs / max(done_pairs, 1)
    timing = {"seconds": t_pre + t_tiles * scale + t_adam, "t_pre": t_pre, "t_tiles": t_tiles, "t_adam": t_adam,
              "tiles_done": done_tiles, "n_tiles": n_tiles, "pairs_done": done_pairs, "pairs": total_pairs,
              "extrapolated": done_tiles < n_tiles}
    if timing["extrapolated"]:
        return timing, None
    orc = {"images": {"color": color.detach(), **other}, "final_T": final_T, "n_contrib": n_contrib, "radii": geom["radii"],
           "D": total_pairs, "loss": float(loss.detach()), "vals_sorted": binning["vals_sorted"], "ranges": binning["ranges"],
           "grid_x": gx, "grads": {**{k: x[k].grad for k in x}, "viewmatrix": vm.grad, "means2D": m2.grad}}
    return timing, orc


def hip_frame(frame, gt, sh_degree, dev):
    """The product path on the rasterizer inputs of `frame`: GaussianRasterizer forward, the fused 0.8 L1 + 0.2 D-SSIM
    kernel against `gt`, backward; everything the parity check compares, on the host.  Runs on its own RasterState (no
    deferred check: D is read back), outside every timed region."""
    from rodygs_amd.losses import fused_photometric_loss
    from rodygs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, RasterState,
                                       last_compositing_state)
    H, W = frame["H"], frame["W"]
    P = frame["means3D"].shape[0]
    st = RasterState()
    ins = {k: frame[k].to(dev).clone().requires_grad_(True) for k in FRAME_KEYS}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    rs = GaussianRasterizationSettings(H, W, frame["tanfovx"], frame["tanfovy"], torch.zeros(3, device=dev), 1.0,
                                       frame["projmatrix"].to(dev).contiguous(), sh_degree, False, False, True, True)
    out = GaussianRasterizer(rs, state=st)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                           scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
    fT, nc = last_compositing_state(st)
    loss = fused_photometric_loss(out[0], gt.to(dev), 0.2)
    loss.backward()
    torch.cuda.synchronize()
    return {"images": {"color": out[0].detach().cpu(), "depth": out[1].detach().cpu(), "alpha": out[3].detach().cpu()},
            "final_T": fT.cpu(), "n_contrib": nc.cpu(), "radii": out[4].cpu(), "D": int(st.capacity_hint[(P, H, W)]),
            "loss": float(loss), "grads": {**{k: v.grad.cpu() for k, v in ins.items()}, "means2D": m2.grad.cpu()}}


def _parity(hip, orc, gt):
    """parity_check entry: the HIP frame against the oracle frame the cpu_baseline leg just computed (every tile, every
    gradient entry; oracle.parity.full_frame_report: 1e-4 per column, misses only where a witnessed pixel flip explains
    them, 5e-3 there)."""
    if hip is None:
        return {"skipped": "no HIP frame was captured for this workload"}
    if orc is None:
        return {"skipped": "the oracle composited only part of the tile grid within --cpu-budget"}
    from oracle.parity import full_frame_report
    # the loss is 0.8 L1 + 0.2 D-SSIM: a pixel whose two renderings (equal to 1e-6) lie on different sides of the ground
    # truth has dL/dpixel of opposite sign in the L1 term -- witnessed here, passed on as the loss's own kink
    hc, oc = hip["images"]["color"].double(), orc["images"]["color"].double()
    kink = (torch.sign(hc - gt.double()) != torch.sign(oc - gt.double())).any(dim=0)
    rep = full_frame_report(hip, orc, orc["vals_sorted"], orc["ranges"], orc["grid_x"], loss_kink=kink)
    rep["loss_hip"], rep["loss_oracle"] = hip["loss"], orc["loss"]
    return rep


def cpu_baseline(frame, gt, hip, sh_degree, dev, budget_s=60.0, threads=16):
    """cpu_baseline leg (N = 1, rank 0): the oracle-built CPU train step on this box's host cores.
      value    : the bench frame itself -- the rasterizer inputs of the first frame of the timed region's cycle, as they
                 stand after the timed steps (deformed, activated, that frame's pose), and its ground truth --, one full
                 step; compositing over as many interleaved eighths of the tile grid as fit in `budget_s` (all of them
                 on a box with enough cores -- then nothing is extrapolated; otherwise by splat instances, and `sample`
                 says so);
      c2_full  : BASELINE configs[1] size (100 k Gaussians, 1080p, SH3) timed IN FULL, forward + backward + Adam,
                 median of 3 -- no extrapolation.
    Returns (cpu_baseline dict, parity_check dict): the oracle's image and gradients of both frames are not thrown away
    but compared with the HIP path's on the same inputs (`hip`: hip_frame() of the bench frame)."""
    from rodygs_amd.synthetic import synthetic_scene
    P, H, W = frame["means3D"].shape[0], frame["H"], frame["W"]
    # the oracle's tensors are a few hundred KB each: beyond ~16 threads torch's intra-op fork/join costs more than it
    # buys (measured on the 128-core GPU box); `cores` reports the threads actually used
    cores = max(1, min(torch.get_num_threads(), int(threads)))
    torch.set_num_threads(cores)
    r, orc = _cpu_train_step(frame, gt, sh_degree, budget_s)
    parity = {"bench_frame": _parity(hip, orc, gt)}
    del orc
    how = ("every tile composited: no extrapolation" if not r["extrapolated"] else
           f"{r['tiles_done']}/{r['n_tiles']} tiles holding {r['pairs_done']}/{r['pairs']} splat instances composited, "
           f"the rest extrapolated by instances")
    c2 = synthetic_scene(100000, 1920, 1080, 3, seed=777)
    gt2 = torch.rand(3, 1080, 1920, generator=torch.Generator().manual_seed(1))
    hip2 = hip_frame(c2, gt2, sh_degree, dev)
    runs = []
    for i in range(3):
        r2, orc2 = _cpu_train_step(c2, gt2, sh_degree, 1e9)
        runs.append(r2["seconds"])
        if i == 0:
            parity["c2"] = _parity(hip2, orc2, gt2)
        del orc2
    runs.sort()
    n_cpu, model = _host_cpu()
    base = {"value": 1.0 / r["seconds"], "unit": "frames/s", "cores": cores, "kind": "port",
            "host": {"cpu_count": n_cpu, "cpu_model": model, "torch_threads_used": cores},
            "sample": f"oracle train step (activations + rasterizer fwd/bwd + 0.8 L1 + 0.2 D-SSIM + torch Adam eps 1e-15; "
                      f"no deformation MLP) of the bench frame (the deformed, activated cloud and pose of the timed "
                      f"region's first frame), {P} Gaussians {W}x{H}: per-Gaussian stage + binning "
                      f"{r['t_pre']:.1f}s, compositing fwd+bwd {r['t_tiles']:.1f}s ({how}), Adam {r['t_adam']:.1f}s",
            "c2_full": {"value": 1.0 / runs[1], "unit": "frames/s", "seconds_median_of_3": runs[1], "runs_s": runs,
                        "workload": "100000 Gaussians, 1920x1080, SH3 (BASELINE configs[1] size), every tile, "
                                    "fwd + bwd + Adam, no extrapolation"}}
    return base, parity


def capture_bench_frame(ds, frame_idx):
    """The rasterizer inputs the train step hands to the rasterizer for video frame `frame_idx`, as the parameters stand
    now (deformation MLP + per-Gaussian deformation + activations, that frame's learnable pose), copied to the host: what
    the cpu_baseline leg times the oracle on and what hip_frame() renders for the parity check."""
    from rodygs_amd.model_ops import pose_view_matrix
    with torch.no_grad():
        xyz, opacity, scaling, rot, feats = ds.gaussians_at(frame_idx)
        vm = pose_view_matrix(ds.cam_q, ds.cam_t, int(frame_idx))
        fr = {"means3D": xyz, "shs": feats, "opacities": opacity, "scales": scaling, "rotations": rot, "viewmatrix": vm}
        fr = {k: v.detach().float().cpu().contiguous().clone() for k, v in fr.items()}
    fr.update(H=ds.H, W=ds.W, tanfovx=ds.tanfovx, tanfovy=ds.tanfovy, projmatrix=ds.proj_t.detach().cpu().clone())
    return fr, ds.gt[int(frame_idx)].detach().cpu().clone()


def run_mode(args, mode, rank, world, dev, backend, scene, target):
    """Build the scene for one N > 1 formulation ("allreduce" | "shard"; "single" at N = 1), warm up, time EXACTLY
    args.steps steps between barriers, and collect the per-stage table from a few extra steps.  Returns a dict of raw
    measurements (every rank; dt is the MAX over ranks)."""
    from rodygs_amd import _lib, rasterizer
    from rodygs_amd.trainstep import DynamicScene
    P, W, H = args.points, args.width, args.height
    # rows of the cloud along the Z curve of their positions (RDG_SPATIAL_ORDER=0: the generator's random order)
    spatial_order = os.environ.get("RDG_SPATIAL_ORDER", "1") != "0"
    ds = DynamicScene(scene, num_frames=args.frames, sh_degree=3, device=dev, seed=777, full_losses=args.full_losses,
                      spatial_order=spatial_order)
    # the statistics every reference iteration below densify_until_iter keeps (max_radii2D, xyz_gradient_accum, denom:
    # /root/reference/src/trainer/rodygs.py:316-341) are part of the step: updated inside the per-Gaussian backward kernel
    densify_stats = not args.no_densify_stats
    if densify_stats:
        ds.track_densification()
    if args.no_normal:
        ds.raster_state.render_normal = False
    n_gt = min(args.gt_frames * world, args.frames)
    perm = sorted(set(int(round(i * args.frames / n_gt)) % args.frames for i in range(n_gt)))
    # Visit order: the reference draws its frames from a random permutation of the video (src/data/dataloader.py:28-71).  Its
    # deterministic stand-in here: the orbit positions in bit-reversed order (0, 8, 4, 12, 2, ...), so that ANY window of
    # steps samples the orbit evenly -- walking the orbit in order, a 20-step window was one pass plus the four frames it
    # happened to start on (the heavy side of the orbit: 1.63 ms per step where 100 steps read 1.53 on the same box).
    perm = [perm[j] for j in _spread_order(len(perm))]
    ds.make_ground_truth(target, perm)
    sharded = mode == "shard"
    ss = None
    if sharded:
        from rodygs_amd.sharded import HostStagedExchange, ShardedDynamicScene
        try:
            ss = ShardedDynamicScene.from_replica(ds, rank, world, None if backend == "nccl" else HostStagedExchange())
        except (NotImplementedError, ValueError) as e:
            # a configuration the sharded step does not cover (decided from sizes every rank shares, so all ranks take
            # this branch together): the replicated formulation runs instead
            if rank == 0:
                print(f"bench.py: sharded frame-DP unavailable ({e}); using the replicated formulation", file=sys.stderr)
            sharded = False
    if sharded:
        ds.fp = ds.sync = ds.m2 = None           # the replica's full-size buffers are not needed any more
        torch.cuda.empty_cache()
        train_step = lambda st_: ss.train_step(st_, perm)                       # noqa: E731
    else:
        train_step = lambda st_: ds.train_step(st_, rank, world, perm)          # noqa: E731
    graphed = None
    use_graph = bool(args.graph and world == 1 and not sharded)
    if sharded and densify_stats:
        ss.track_densification()
    rstate = rasterizer.DEFAULT_STATE if sharded else ds.raster_state     # whose frame-to-frame memory the steps use

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rstate.deferred_overflow_check = False
    step = 0
    for _ in range(args.warmup):
        train_step(step)
        step += 1
    # After the warm-up the instance count D of every frame is known to within a few percent: stop reading it back
    # inside the forward (no host wait in the step).  Capacity is 1.25x the last D; an overflow would render that
    # frame empty and raise RasterizerCapacityOverflow at the next forward / at the final poll below.
    points_after = None
    if args.densify_first and not sharded:
        if not densify_stats:
            raise SystemExit("--densify-first needs the densification statistics (drop --no-densify-stats)")
        # one densify-and-prune on the statistics of the steps so far (thresholds: the reference's rule at the 80 % quantile of
        # the mean screen-space gradient, so that about a fifth of the visible cloud is cloned or split): the timed steps
        # then run on a cloud whose rows were rebuilt, re-sorted along the Z curve and appended to -- not a BASELINE
        # config, a check that the headline does not depend on a cloud that never densified
        for _ in range(max(0, 24 - args.warmup)):
            train_step(step)
            step += 1
        with torch.no_grad():
            g_mean = (ds.stats.xyz_gradient_accum / ds.stats.denom.clamp_min(1)).reshape(-1)
            seen = ds.stats.denom.reshape(-1) > 0
            thr = float(torch.quantile(g_mean[seen][:2000000], 0.8)) if bool(seen.any()) else 1e9
        info = ds.densify(max_grad=thr, min_opacity=0.005, percent_dense=0.01)
        points_after = int(info["P"])
        rstate = ds.raster_state
        for _ in range(args.warmup):             # the new (P, H, W) needs its own capacity hint before the deferred check
            train_step(step)
            step += 1
    rstate.deferred_overflow_check = True
    # Untimed settling steps in the configuration the timed region runs in (deferred check, its pinned slots and hints in
    # place; clocks and allocator warm): W = 5 warm-up steps are 8 ms of GPU work, and the FIRST bench run on a fresh box
    # has read 1.93 ms per step where every later run of the same binary reads 1.63.  Not part of W, not timed.
    # The interpreter's cyclic collector runs a full (generation-2) pass once the start-up garbage has piled up -- a 40 ms
    # host stall that would land somewhere in a 20-step window: collect now and keep the survivors out of later passes.
    # BEFORE the settling steps, not after them: the GPU idles while the host collects, its clocks fall, and the first
    # four or five steps after such a pause run 8-15 % slow (kernel durations in a rocprofv3 trace: 1.74, 1.77, 1.69,
    # 1.66 ms, then 1.55) -- a 20-step window read 1.63-1.65 ms per step where 100 steps read 1.53-1.55.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.settle):
        train_step(step)
        step += 1
    if use_graph:
        # capture after the warm-up (capacity and binning hints are known); the timed region replays the graph
        from rodygs_amd.trainstep import GraphedStep
        rstate.deferred_overflow_check = False
        rstate.poll_overflow(block=True)
        graphed = GraphedStep(ds, perm, warmup=2, first_step=step)
        step = graphed.next_step
        train_step_eager = train_step
        train_step = lambda st_: graphed.step()                                  # noqa: E731
    # Inside the timed region only the dominant kernel is bracketed by hipEvents (every timed stage costs ~10 us of
    # stream gap); the per-stage table is taken from a few extra steps afterwards.  (A replayed graph cannot bracket one
    # of its kernels: the dominant kernel's time is then taken from the eager steps after the timed region.)
    _lib.timing_enable(not use_graph, stages=["render_bwd"])
    _lib.timing_reset()
    deferred_in_timed_region = rstate.mode("deferred_overflow_check")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(step)
        step += 1
    sync()
    dt = time.perf_counter() - t0
    if graphed is not None:
        graphed.check()                      # raises if ANY replayed frame outgrew the captured capacity (sticky record)
        step = graphed.next_step
        graphed.close()
        train_step = train_step_eager
    rstate.poll_overflow(block=True)
    dom = _lib.stage_times()["render_bwd"]
    _lib.timing_enable(True)
    _lib.timing_reset()
    n_stage_steps = min(5, args.steps)
    for _ in range(n_stage_steps):
        train_step(step)
        step += 1
    sync()
    rstate.poll_overflow(block=True)
    # milliseconds PER STEP of every stage: a stage that is entered more than once per step (the optimiser since round 6: the rows'
    # launch and the MLP + pose launch) counts with the sum of its entries, which is what its bytes per step are priced against
    stages = {k: (ms, n_stage_steps if n else 0) for k, (ms, n) in _lib.stage_times().items()}
    if graphed is None:
        stages["render_bwd"] = dom           # the dominant kernel: per LAUNCH over the timed region (what `roofline` prices)
    _lib.timing_enable(False)
    rstate.deferred_overflow_check = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"mode": mode, "sharded": sharded, "dt": dt, "loss": float(loss.item()), "spatial_order": spatial_order,
           "graph": graphed is not None, "densify_stats": densify_stats, "deferred": deferred_in_timed_region,
           # the binning algorithm the timed steps ran: forced by RDG_BIN_MODE, or what the per-frame rule has picked
           "radix": bool(rstate.mode("force_radix") or any(rstate.bin_hint.values())),
           "deterministic": rstate.mode("deterministic"), "render_normal": rstate.mode("render_normal"),
           "points_after_densify": points_after,
           "per_stage": {k: (ms / n if n else 0.0) for k, (ms, n) in stages.items()}}
    if rank == 0:
        if sharded:
            out["D"] = int(rstate.capacity_hint.get(ss.key, 0))
            out["V"] = ss.visible_count()
            with torch.no_grad():
                _, n_contrib = rasterizer.last_compositing_state(rstate)
                out["S"] = int(n_contrib.sum(dtype=torch.int64).item())
        else:
            # D, V and S: means over the frames the timed region cycles through (each forward reads its instance count
            # back: deferred mode is off again) -- the roofline divides by an average over the same frames
            dsum = vsum = ssum = drefsum = 0
            with torch.no_grad():
                for f in perm:
                    o, _ = ds.render(f)
                    vsum += int((o[4] > 0).sum().item())
                    dsum += int(rstate.capacity_hint.get((ds.P, H, W), 0))
                    _, n_contrib = rasterizer.last_compositing_state(rstate)
                    ssum += int(n_contrib.sum(dtype=torch.int64).item())
                # the same frames under the REFERENCE's tile rectangles (RdgRasterSettings.cull = 0): what the reference
                # algorithm would have binned, sorted and staged -- its own RasterState, nothing of the timed path is touched
                ref_state = rasterizer.RasterState(cull=False)
                ds.raster_state = ref_state
                try:
                    for f in perm:
                        ds.render(f)
                        drefsum += int(ref_state.capacity_hint.get((ds.P, H, W), 0))
                finally:
                    ds.raster_state = rstate
            out["D"], out["V"], out["S"] = dsum // len(perm), vsum // len(perm), ssum // len(perm)
            out["D_ref"] = drefsum // len(perm)
            out["cull"] = bool(rstate.mode("cull"))
    if rank == 0 and world == 1 and not sharded and not args.no_cpu_baseline:
        # the frame the cpu_baseline leg times the oracle on and the parity check compares: first frame of the cycle
        out["bench_frame"] = capture_bench_frame(ds, perm[0])
    del ds, ss, train_step
    gc.unfreeze()
    gc.collect()
    torch.cuda.empty_cache()
    return out


def run_loop(args, dev, scene, target):
    """--loop K: the train LOOP, not the step -- K steps with densify_and_prune every --densify-interval steps, the reference's
    cadence (/root/reference/src/trainer/rodygs.py:343-356, src/trainer/rodygs_static.py:285-301: every 100 iterations, minimum
    opacity 0.005, extent = spatial_lr_scale, no screen-size limit below opacity_reset_interval).  The gradient threshold is the
    --densify-quantile quantile of the mean screen-space gradient (taken on the device: no read-back), so that a fixed share of
    the cloud is cloned / split whatever the synthetic data's gradient scale is (--densify-grad-threshold: the reference's
    absolute 0.0002 instead).  Everything between the first and the last step is inside the clock: the densifications, the first
    steps on the new buffers (allocator, Adam segment table, birth-order sort, new workspaces), the graph re-captures.
    Reported: sustained fps; per segment P and ms / step (first 20 steps / the rest); per densification its wall time (with
    --loop-profile: per phase, synchronising) and the P trajectory; the steady-state fps of the same window (the segments' settled
    step times) and sustained / steady."""
    import gc
    from rodygs_amd import rasterizer
    from rodygs_amd.trainstep import DynamicScene, GraphedStep
    P, W, H = args.points, args.width, args.height
    spatial_order = os.environ.get("RDG_SPATIAL_ORDER", "1") != "0"
    ds = DynamicScene(scene, num_frames=args.frames, sh_degree=3, device=dev, seed=777, spatial_order=spatial_order)
    fixed = args.fixed_capacity > 1.0
    if fixed:
        ds.fix_capacity(args.fixed_capacity)
    else:
        ds.track_densification()
    n_gt = min(args.gt_frames, args.frames)
    perm = sorted(set(int(round(i * args.frames / n_gt)) % args.frames for i in range(n_gt)))
    perm = [perm[j] for j in _spread_order(len(perm))]
    ds.make_ground_truth(target, perm)
    ds.raster_state.deferred_overflow_check = False
    regrown = 0
    graphed = None

    def densify_now(tm=None):
        # one densify_and_prune with the loop's threshold rule; fixed capacity: in place, growing the buffers when the dead rows run out
        nonlocal regrown
        if not fixed:
            return ds.densify(max_grad=threshold(), min_opacity=0.005, percent_dense=0.01, want_decisions=False, timings=tm)
        thr = args.densify_grad_threshold if args.densify_grad_threshold > 0 else ds.live_gradient_quantile(args.densify_quantile)
        info = ds.densify_inplace(max_grad=thr, min_opacity=0.005, percent_dense=0.01)
        if info is None:
            # the dead rows do not suffice: the live rows move into larger buffers (the captured graph, if any, goes: its
            # addresses are the old buffers'), and the densification is repeated there
            nonlocal graphed
            regrown += 1
            if graphed is not None:
                graphed.close()
                graphed = None
            stats = ds.stats
            live = (~ds.dead).nonzero().squeeze(1)
            ds.fix_capacity(args.fixed_capacity)
            for a, b in ((ds.stats.xyz_gradient_accum, stats.xyz_gradient_accum), (ds.stats.denom, stats.denom),
                         (ds.stats.max_radii2D, stats.max_radii2D)):
                a[:live.numel()] = b.index_select(0, live)
            thr = args.densify_grad_threshold if args.densify_grad_threshold > 0 else ds.live_gradient_quantile(args.densify_quantile)
            info = ds.densify_inplace(max_grad=thr, min_opacity=0.005, percent_dense=0.01)
            if info is None:
                raise RuntimeError("fixed capacity exhausted right after growing it: --fixed-capacity is too small for one densification")
        return info
    step = 0
    for _ in range(args.warmup):
        ds.train_step(step, 0, 1, perm)
        step += 1
    ds.raster_state.deferred_overflow_check = True
    gc.collect()
    gc.freeze()
    for _ in range(args.settle):
        ds.train_step(step, 0, 1, perm)
        step += 1

    def threshold():
        with torch.no_grad():
            if args.densify_grad_threshold > 0:
                return args.densify_grad_threshold
            g_mean = (ds.stats.xyz_gradient_accum / ds.stats.denom.clamp_min(1)).reshape(-1)
            return torch.quantile(g_mean, args.densify_quantile)

    # The first densification of a process pays for the lazy initialisation of ~40 framework kernels (180 ms): like the first
    # train steps it belongs to the warm-up.  One real densification on the statistics of the steps so far, untimed, then the
    # statistics start afresh and a few more untimed steps settle the new cloud; the P trajectory below starts from its result.
    warm = densify_now()
    for _ in range(min(args.settle, 20)):
        ds.train_step(step, 0, 1, perm)
        step += 1
    ds.stats.xyz_gradient_accum.zero_(); ds.stats.denom.zero_(); ds.stats.max_radii2D.zero_()

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    K, interval, head = args.loop, args.densify_interval, 20
    segments, densifies = [], []
    done = 0
    graphed = None
    graph_pool = cap_tm = None
    overflows = 0
    t_begin = sync()
    while done < K:
        n = min(interval, K - done)
        t_cap = 0.0
        if args.graph and graphed is None:
            t0 = sync()
            ds.raster_state.deferred_overflow_check = False
            ds.raster_state.poll_overflow(block=True)
            cap_tm = {} if args.loop_profile else None
            graphed = GraphedStep(ds, perm, warmup=1, first_step=step, timings=cap_tm,      # (its eager warm-up step is one
                                  pool=graph_pool)                                            #  of the segment's n)
            graph_pool = graphed.pool()
            eager_done = graphed.next_step - step
            step = graphed.next_step
            t_cap = (sync() - t0) * 1e3
        else:
            eager_done = 0
        t0 = sync()
        t_head = None
        for i in range(eager_done, n):
            if i == head:
                t_head = sync()
            if graphed is not None:
                graphed.step()
            else:
                try:
                    ds.train_step(step, 0, 1, perm)
                except rasterizer.RasterizerCapacityOverflow:
                    # an earlier frame of the deferred check outgrew its workspace (it was rendered empty; the hint is raised now):
                    # counted and reported -- a line with overflows is not a valid measurement
                    overflows += 1
            step += 1
        t1 = sync()
        if graphed is not None:
            keep_graph = fixed             # fixed capacity: the captured step survives the densification that follows
            try:
                graphed.check()
            except rasterizer.RasterizerCapacityOverflow:
                overflows += 1           # counted like the eager path's; the next segment re-captures with the raised hint
                keep_graph = False
            finally:
                step = graphed.next_step
                if not keep_graph:
                    graphed.close()
                    graphed = None
                    ds.raster_state.deferred_overflow_check = True
        try:
            ds.raster_state.poll_overflow(block=True)
        except rasterizer.RasterizerCapacityOverflow:
            overflows += 1
        n_head = min(head, n) - eager_done
        segments.append({"P": ds.P_live if fixed else ds.P, "rows": ds.P, "steps": n, "graph_capture_ms": t_cap, "graph_capture_phases_ms": cap_tm,
                         "ms_per_step_first_20": ((t_head or t1) - t0) * 1e3 / max(n_head, 1),
                         "ms_per_step_settled": ((t1 - t_head) * 1e3 / (n - head)) if (t_head is not None and n > head) else None})
        done += n
        if done < K:
            t0 = sync()
            tm = ({"_nosync": True} if args.loop_profile == "host" else {}) if args.loop_profile else None
            t_q = time.perf_counter()
            p0 = ds.P_live if fixed else ds.P
            if tm is not None:
                tm["threshold_host"] = (time.perf_counter() - t_q) * 1e3
            info = densify_now(tm)
            t1 = sync()
            densifies.append({"after_step": done, "ms": (t1 - t0) * 1e3, "P_before": p0, "P_after": info["P"],
                              "cloned": info["cloned"], "split": info["split"], "pruned": info["pruned"], "phases_ms": tm})
    t_end = sync()
    if graphed is not None:
        graphed.close()
    ds.raster_state.deferred_overflow_check = False
    gc.unfreeze()
    total = t_end - t_begin
    settled = [(sg["steps"], sg["ms_per_step_settled"]) for sg in segments if sg["ms_per_step_settled"]]
    steady_ms = sum(n * ms for n, ms in settled) / max(sum(n for n, _ in settled), 1)
    sustained = K / total
    mean_P = sum(sg["P"] * sg["steps"] for sg in segments) / K
    return {"metric": f"sustained train-LOOP fps at {P} initial dynamic Gaussians / {W}x{H} (fwd+bwd+Adam+statistics, "
                      f"densify_and_prune every {interval} steps)",
            "value": sustained, "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": args.warmup,
            "ms_per_step": total / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_label(P, W, H, args.frames, False, 1, args.scene) + f", train loop with "
                       f"densification every {interval} steps", "points": P, "width": W, "height": H, "scene": args.scene,
                       "graph_replay": bool(args.graph), "densify_interval": interval,
                       # > 0: the cloud lives in buffers of a fixed number of rows (dead rows + in-place densification); with
                       # --graph the step is captured once for the whole loop
                       "fixed_capacity": args.fixed_capacity if fixed else 0.0, "rows": ds.P if fixed else None,
                       "graph_captures": sum(1 for sg in segments if sg["graph_capture_ms"] > 0), "capacity_regrown": regrown,
                       "points_after_warmup_densification": int(warm["P"]),
                       "densify_threshold": (args.densify_grad_threshold if args.densify_grad_threshold > 0 else
                                             f"{args.densify_quantile} quantile of the mean screen-space gradient"),
                       "row_order": "z-curve of the canonical positions, kept through every densification" if spatial_order
                       else "generator (random)", "deferred_overflow_check": True,
                       "allocator": os.environ.get("PYTORCH_HIP_ALLOC_CONF", "default")},
            "gaussians_per_s": sustained * mean_P,
            "loop": {"sustained_fps": sustained, "steady_state_fps_of_the_window": 1e3 / steady_ms if steady_ms else None,
                     "sustained_over_steady": (sustained * steady_ms / 1e3) if steady_ms else None,
                     "capacity_overflows": overflows, "mean_P": mean_P, "P_trajectory": [sg["P"] for sg in segments], "segments": segments,
                     "densifications": densifies,
                     "densify_ms_mean": (sum(d_["ms"] for d_ in densifies) / len(densifies)) if densifies else None}}


def run_reference_iteration(args, dev):
    """--iteration reference: the iteration a RoDyGS user runs (/root/reference/src/trainer/rodygs.py:157-179) -- a static
    sub-step and a dynamic sub-step, each rendering the CONCATENATED cloud (static ‖ dynamic + deformation), with the
    reference's stale-gradient semantics (trainstep.ReferenceIteration) -- next to the per-step headline, which BASELINE quotes
    on the dynamic cloud alone.  --points Gaussians are split evenly between the two clouds, so every sub-step renders --points
    Gaussians.  value = iterations / s (one iteration = two optimiser steps, two renders)."""
    import gc
    from rodygs_amd.refiter import FusedReferenceIteration, GraphedIteration
    from rodygs_amd.synthetic import synthetic_scene
    from rodygs_amd.trainstep import ReferenceIteration
    P, W, H = args.points, args.width, args.height
    ps = P // 2
    # "reference": the fused bookkeeping (refiter.FusedReferenceIteration: same semantics, tested against the other);
    # "reference-v1": the reference-shaped class (autograd accumulation, f_dc / f_rest split, clears)
    cls = ReferenceIteration if args.iteration == "reference-v1" else FusedReferenceIteration
    ri = cls(synthetic_scene(ps, W, H, 3, seed=777, variant=args.scene),
             synthetic_scene(P - ps, W, H, 3, seed=778, variant=args.scene), num_frames=args.frames, device=dev)
    n_gt = min(args.gt_frames, args.frames)
    perm = sorted(set(int(round(i * args.frames / n_gt)) % args.frames for i in range(n_gt)))
    perm = [perm[j] for j in _spread_order(len(perm))]
    ri.make_ground_truth(synthetic_scene(max(P // 4, 1000), W, H, 3, seed=1234, variant=args.scene), perm)
    it = 0
    for _ in range(args.warmup):
        ri.iteration(it, perm)
        it += 1
    ri.raster_state.deferred_overflow_check = True
    gc.collect()
    gc.freeze()
    for _ in range(args.settle // 2):
        ri.iteration(it, perm)
        it += 1
    graphed = None
    if args.graph:
        if cls is ReferenceIteration:
            raise SystemExit("--graph with --iteration reference-v1: only the fused iteration can be captured")
        ri.raster_state.deferred_overflow_check = False
        ri.raster_state.poll_overflow(block=True)
        graphed = GraphedIteration(ri, perm, first_iteration=it, warmup=2)
        it = graphed.next_iteration
        for _ in range(5):
            graphed.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graphed is not None:
            loss = graphed.step()
        else:
            loss = ri.iteration(it, perm)
            it += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if graphed is not None:
        graphed.check()
        graphed.close()
    ri.raster_state.poll_overflow(block=True)
    ri.raster_state.deferred_overflow_check = None
    gc.unfreeze()
    return {"metric": f"reference-iteration rate at {ps} static + {P - ps} dynamic Gaussians / {W}x{H} (static sub-step + dynamic "
                      f"sub-step, each fwd+bwd over the concatenated cloud + its own Adam)",
            "value": args.steps / dt, "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{ps} static + {P - ps} dynamic Gaussians + deformation MLP, {W}x{H}, SH3, {args.frames}-frame "
                                   f"synthetic video: the reference's iteration (not a BASELINE config: its metric is per step on "
                                   f"the dynamic cloud)", "points": P, "width": W, "height": H, "scene": args.scene,
                       "sub_steps_per_iteration": 2, "gaussians_rendered_per_sub_step": P,
                       "bookkeeping": ("reference-shaped (autograd accumulation into .grad, f_dc / f_rest split, zero_grad)"
                                       if cls is ReferenceIteration else
                                       "fused (two gradient buffers written in turn, Adam on their sum, one feature tensor)"),
                       "graph_replay": graphed is not None,
                       "stale_gradient_semantics": "kept (a sub-step's backward accumulates in both clouds; only its own "
                                                   "trainer steps and clears)", "densify_stats": True,
                       "deferred_overflow_check": True},
            "ms_per_sub_step": dt / args.steps * 1e3 / 2, "losses": [float(v) for v in loss]}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks the way the driver's torchrun command does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)
    as a CHILD process, pass its output through (rank 0's JSON line is the only line on stdout) and return its exit status.
    This parent never initialises a GPU (torch.cuda.device_count() does not, on this image) and never exec()s: a process
    that has touched the GPU must not be replaced, and the children need fresh ones anyway.  Fewer visible devices than
    ranks is an error (exit 2) unless RDG_ONE_DEVICE=1 (functional check of the N > 1 flow on one device, with
    RDG_DIST_BACKEND=gloo: RCCL refuses two ranks on one device)."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < n and not os.environ.get("RDG_ONE_DEVICE"):
        print(f"bench.py: --gpus {n} needs {n} visible devices, this node shows {n_dev} "
              f"(RDG_ONE_DEVICE=1 RDG_DIST_BACKEND=gloo runs every rank on device 0 as a functional check)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL across processes needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


def live_pmc(P, W, H, scene, timeout_s=120.0):
    """Hardware counters of THIS run's box and build: three child runs of a short bench under `rocprofv3 --pmc ...` --
    FETCH_SIZE, WRITE_SIZE (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes for the HBM bytes; FETCH_SIZE
    doubled per its gfx950 note) and one pass of the SQ instruction counters -- counters only, no trace domain next to them; the
    program after `--` is python3 itself (no shell, no env wrapper).  Parsed like scripts/pmc_hbm_traffic.py / pmc_summary.py.
    Returns (tables, note): tables = {counter: {kernel (template arguments stripped): (mean per launch, launches)}} or None."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found on this box"
    env = dict(os.environ, TMPDIR="/tmp")
    tables = {}
    tmp = tempfile.mkdtemp(prefix="rdg_pmc_", dir="/tmp")
    t0 = time.perf_counter()
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE")):
            d = os.path.join(tmp, counters[0])
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "4", "--warmup", "2", "--settle", "4", "--points", str(P), "--width", str(W), "--height", str(H),
                   "--scene", scene, "--no-cpu-baseline", "--no-sub-records"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {' '.join(counters)} pass failed (exit {r.returncode}): {(r.stderr or '')[-200:]}"
            tot = {c: collections.defaultdict(lambda: [0.0, 0]) for c in counters}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] not in tot:
                        continue
                    k = re.sub(r"<.*>", "", row["Kernel_Name"].split("(")[0].replace("void ", "")).strip()
                    a = tot[row["Counter_Name"]][k]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
            for c in counters:
                tables[c] = {k: (v[0] / v[1], v[1]) for k, v in tot[c].items() if v[1] and k.startswith("rdg_")}
        if "rdg_render_bwd_kernel" not in tables.get("FETCH_SIZE", {}):
            return None, "the PMC passes ran but hold no rdg_render_bwd_kernel rows"
        return tables, (f"LIVE: rocprofv3 --pmc child passes of this run (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU "
                        f"GRBM_GUI_ACTIVE; {tables['FETCH_SIZE']['rdg_render_bwd_kernel'][1]} launches of the dominant kernel, "
                        f"{time.perf_counter() - t0:.0f} s); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950 correction)")
    except Exception as e:                                    # noqa: BLE001
        return None, f"live PMC passes failed: {type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _child_json(argv, timeout_s, script="bench.py"):
    """Run `python <script> argv...` as a CHILD process (never exec: this process has initialised the GPU) and return the last
    JSON line of its stdout, or {"error": ...}.  Sub-records must never take the headline down with them."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, script)] + [str(a) for a in argv]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"exit {r.returncode}: {(r.stderr or r.stdout)[-300:]}", "seconds": time.perf_counter() - t0}
        out = json.loads(lines[-1])
        out["_seconds"] = time.perf_counter() - t0
        return out
    except Exception as e:                                    # noqa: BLE001  (timeout, bad JSON, ...)
        return {"error": f"{type(e).__name__}: {e}"[:300], "seconds": time.perf_counter() - t0}


def sub_records(budget_s=170.0):
    """What DESIGN.md section 5 claims beside the headline, measured in THE SAME driver run: each a short
    child run of this file's own modes, outside the headline's timed region, reduced to a few numbers.
      loop                : 300 steps with the reference's densification cadence (every 100): sustained / steady fps, overflows;
      loop_100k           : the same loop at the reference's real cloud size, eager and as ONE captured graph over a cloud of fixed
                            capacity (DynamicScene.fix_capacity: dead rows, densification in place);
      reference_iteration : the iteration a RoDyGS user runs (static + dynamic sub-step), at 0.5 M + 0.5 M and 0.1 M + 0.1 M
                            (refiter.FusedReferenceIteration; the 0.1 M form also as one replayed hipGraph; the reference-shaped
                            bookkeeping of trainstep.ReferenceIteration next to it);
      graph_100k          : the reference's real cloud size, eager step against the replayed hipGraph;
      psnr_delta          : the teacher-forced protocol of scripts/psnr_delta.py, 100 steps with a densification at 20 k points /
                            320x240 (the 100 k / 1080p form takes 410 s of oracle: profiles/r03_psnr_teacher_forced_100k_1080p.json)."""
    t_all = time.perf_counter()
    out = {}

    def left():
        return budget_s - (time.perf_counter() - t_all)

    def pick(d, keys):
        return {k: d.get(k) for k in keys} if "error" not in d else d

    lp = _child_json(["--loop", "300", "--densify-interval", "100", "--settle", "20", "--warmup", "3"], max(20.0, min(90.0, left())))
    if "error" not in lp:
        L = lp["loop"]
        out["loop"] = {"steps": lp["steps"], "densifications_in_the_clock": len(L["densifications"]),
                       "sustained_fps": L["sustained_fps"], "steady_state_fps": L["steady_state_fps_of_the_window"],
                       "sustained_over_steady": L["sustained_over_steady"], "capacity_overflows": L["capacity_overflows"],
                       "P_trajectory": L["P_trajectory"], "densify_ms_mean": L["densify_ms_mean"], "seconds": lp["_seconds"]}
    else:
        out["loop"] = lp
    # the reference's real cloud size in the loop: eager against ONE captured graph over a cloud of fixed capacity
    l100 = {}
    for name, extra in (("eager", []), ("one_graph_fixed_capacity", ["--graph", "--fixed-capacity", "1.2"])):
        if left() < 12:
            l100[name] = {"error": "sub-record budget spent"}
            continue
        r = _child_json(["--loop", "300", "--points", "100000", "--densify-interval", "100", "--settle", "20", "--warmup", "3"] + extra,
                        max(12.0, min(45.0, left())))
        l100[name] = ({"sustained_fps": r["loop"]["sustained_fps"], "sustained_over_steady": r["loop"]["sustained_over_steady"],
                       "capacity_overflows": r["loop"]["capacity_overflows"], "graph_captures": r["config"].get("graph_captures"),
                       "seconds": r["_seconds"]} if "error" not in r else r)
    out["loop_100k"] = l100
    ri = {}
    for name, pts, mode, extra in (("0.5M+0.5M", 1000000, "reference", []), ("0.1M+0.1M", 200000, "reference", []),
                                   ("0.1M+0.1M graph", 200000, "reference", ["--graph"]),
                                   ("0.5M+0.5M reference-shaped bookkeeping", 1000000, "reference-v1", [])):
        if left() < 15:
            ri[name] = {"error": "sub-record budget spent"}
            continue
        r = _child_json(["--iteration", mode, "--points", pts, "--steps", "40", "--settle", "20", "--warmup", "3"] + extra,
                        max(15.0, min(60.0, left())))
        ri[name] = ({"iterations_per_s": r["value"], "ms_per_iteration": r["ms_per_step"], "ms_per_sub_step": r["ms_per_sub_step"],
                     "bookkeeping": r["config"].get("bookkeeping"), "graph_replay": r["config"].get("graph_replay", False),
                     "seconds": r["_seconds"]} if "error" not in r else r)
    out["reference_iteration"] = ri
    g = {}
    for name, extra in (("eager", []), ("graph", ["--graph"])):
        if left() < 12:
            g[name] = {"error": "sub-record budget spent"}
            continue
        r = _child_json(["--points", "100000", "--steps", "200", "--settle", "20", "--warmup", "3", "--no-cpu-baseline"] + extra,
                        max(12.0, min(45.0, left())))
        g[name] = {"fps": r["value"], "ms_per_step": r["ms_per_step"], "seconds": r["_seconds"]} if "error" not in r else r
    out["graph_100k"] = g
    if left() > 25:
        d = _child_json(["--teacher-forced", "--points", "20000", "--width", "320", "--height", "240", "--steps", "100"],
                        max(25.0, min(90.0, left())), script=os.path.join("scripts", "psnr_delta.py"))
        if "error" not in d:
            out["psnr_delta"] = {"protocol": "teacher-forced (scripts/psnr_delta.py): at every state of the oracle's training the HIP "
                                             "gradient is computed too; drift = accumulated PSNR difference of the next states",
                                 "drift_db": d["drift_db"], "abs_sum_db": d["abs_sum_db"], "steps": d["steps"], "gate_db": 0.05,
                                 "within_gate": abs(d["drift_db"]) <= 0.05, "points": d["config"]["points"],
                                 "image": f"{d['config']['width']}x{d['config']['height']}", "psnr_end_db": d["psnr_end_db"],
                                 "vs": "the oracle (the reference CUDA rasterizer cannot run here: no source, no NVIDIA GPU)",
                                 "seconds": d["_seconds"]}
        else:
            out["psnr_delta"] = d
    else:
        out["psnr_delta"] = {"error": "sub-record budget spent"}
    out["seconds_total"] = time.perf_counter() - t_all
    return out


def preflight(n):
    """`bench.py --gpus N --preflight`: what the first real multi-GPU run should know BEFORE it spends its time -- devices,
    peer access, RCCL, the exact bytes each formulation puts on the wire per step -- printed as one JSON line; exit 0 if N ranks
    can run here, 2 otherwise.  Touches the devices (peer-access queries), launches nothing."""
    import ctypes
    P, K = 1000000, 16
    info = {"preflight": True, "gpus_requested": n, "devices_visible": torch.cuda.device_count(),
            "hip": getattr(torch.version, "hip", None), "torch": torch.__version__,
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "wire_bytes_per_step_at_1M": {"allreduce": wire_bytes(P, K, max(n, 2), False), "shard": wire_bytes(P, K, max(n, 2), True)}}
    try:
        info["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                    # noqa: BLE001
        info["rccl"] = f"unavailable: {e}"
    nd = info["devices_visible"]
    names, peer = [], []
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        for i in range(nd):
            names.append(torch.cuda.get_device_name(i))
            row = []
            for j in range(nd):
                can = ctypes.c_int(0)
                rc = hip.hipDeviceCanAccessPeer(ctypes.byref(can), i, j) if i != j else 0
                row.append(1 if i == j else (int(can.value) if rc == 0 else -1))
            peer.append(row)
    except Exception as e:                                    # noqa: BLE001
        info["peer_access_error"] = f"{type(e).__name__}: {e}"
    info["devices"], info["hipDeviceCanAccessPeer"] = names, peer
    ok = nd >= n and all(all(v == 1 for v in row[:n]) for row in peer[:n]) if peer else nd >= n
    info["ok"] = bool(ok)
    if not ok:
        info["why"] = (f"{nd} visible devices for {n} ranks" if nd < n else "some device pairs have no peer access: RCCL would "
                       "fall back to host staging (expect a fraction of the xGMI rate)")
    print(json.dumps(info), flush=True)
    return 0 if ok else 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: about six passes over the frames that have ground truth (--gt-frames: 16 spread over the orbit) -- the
    # kernels take 1.46 ... 1.67 ms with the frame, a 20-step window measures whichever stretch of them it lands on
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=40, help="untimed steps after the warm-up, before the timed region")
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--gt-frames", type=int, default=16, help="distinct frames with ground truth resident in HBM")
    ap.add_argument("--scene", choices=["uniform", "sheets", "dense"], default="uniform",
                    help="rodygs_amd.synthetic variant: 'uniform' = the SURVEY 8d generator (the headline workload); "
                         "'sheets' = opaque depth sheets (early termination); 'dense' = 3x the projected sigma (D ~ 9x)")
    ap.add_argument("--graph", action="store_true", default=os.environ.get("RDG_GRAPH", "0") == "1",
                    help="N = 1: replay the step as ONE captured hipGraph (trainstep.GraphedStep) instead of "
                         "launching its ~50 kernels from Python -- what makes the step kernel-bound at the size of the "
                         "reference's real clouds (~100 k points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the compact loop / reference-iteration / 100 k graph / PSNR-delta records the default N = 1 headline run "
                         "appends (sub_records: child runs after the timed region)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes the default N = 1 headline run takes roofline.traffic from")
    ap.add_argument("--preflight", action="store_true",
                    help="print devices, peer access, RCCL version and the wire bytes of both frame-DP formulations, and exit")
    ap.add_argument("--no-normal", action="store_true",
                    help="N = 1: the scene's RasterState leaves the normal channels out (render_normal=False): no RoDyGS loss reads "
                         "rendered_normal (gt_normal is always None), the upstream rasterizer composites it regardless.  A side line, "
                         "not the headline: config.render_normal says which one a line is")
    ap.add_argument("--densify-first", action="store_true",
                    help="run ONE densify-and-prune (on the statistics of ~24 untimed steps) before the timed region: the "
                         "timed steps then train a cloud that went through the row surgery (not the headline)")
    ap.add_argument("--no-densify-stats", action="store_true",
                    help="leave the per-iteration densification statistics (max_radii2D, xyz_gradient_accum, denom) out of "
                         "the step; by default they are part of it, as in every reference iteration below densify_until_iter")
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch intra-op threads of the cpu_baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=60.0,
                    help="seconds of CPU compositing the cpu_baseline leg may spend on the bench frame before it "
                         "extrapolates the remaining tiles")
    ap.add_argument("--full-losses", action="store_true",
                    help="config-5 loss set (depth, motion regularisers, rigidity every 5th step) instead of the "
                         "photometric-only step the headline metric is quoted on")
    ap.add_argument("--loop", type=int, default=0,
                    help="N = 1: time K steps of the train LOOP with a densify_and_prune every --densify-interval steps "
                         "(run_loop) instead of the steady-state step; prints one JSON line with the `loop` table")
    ap.add_argument("--densify-interval", type=int, default=100)
    ap.add_argument("--fixed-capacity", type=float, default=0.0,
                    help="--loop: keep the cloud in buffers of (this factor) x P rows for good (DynamicScene.fix_capacity: dead rows, "
                         "densify_and_prune in place) -- with --graph the step is then captured ONCE and keeps replaying across the "
                         "densifications (nothing it refers to moves or changes size); 0 = off")
    ap.add_argument("--densify-quantile", type=float, default=0.97)
    ap.add_argument("--densify-grad-threshold", type=float, default=0.0,
                    help="> 0: the reference's absolute threshold on the mean screen-space gradient (its configs: 0.0002) "
                         "instead of the quantile rule")
    ap.add_argument("--loop-profile", nargs="?", const="sync", default=None, choices=["sync", "host"],
                    help="--loop: per-phase times of every densification; 'sync' (default) synchronises the device at every "
                         "phase boundary, 'host' records host time only (where the host blocks)")
    ap.add_argument("--iteration", choices=["step", "reference", "reference-v1"], default="step",
                    help="'reference': time the reference's ITERATION -- static sub-step + dynamic sub-step over the "
                         "two-segment cloud with its stale-gradient semantics (run_reference_iteration) -- instead of the step")
    ap.add_argument("--dp-mode", choices=["both", "allreduce", "shard"], default=os.environ.get("RDG_DP_MODE", "both"),
                    help="N > 1 formulation: 'allreduce' = BASELINE north_star: replicated cloud, frames over the GPUs, "
                         "RCCL all-reduce of the Gaussian / pose gradients (bucketed, overlapped with backward and Adam); "
                         "'shard' = Gaussians sharded over the ranks, 64-byte splat records / gradient rows exchanged "
                         "with two all-to-alls (rodygs_amd/sharded.py); 'both' (default) times EXACTLY --steps steps of "
                         "each, back to back, reports the faster one as `value` and both under `dp_modes`")
    args = ap.parse_args()

    if args.preflight:
        raise SystemExit(preflight(args.gpus))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches a GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE="
                         f"{os.environ.get('WORLD_SIZE', '1')} ranks; they must agree (n_gpus in the JSON line is the number of "
                         f"ranks that ran)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rodygs_amd hot path has no CPU fallback")
    # functional checks of the N > 1 flow on a 1-GPU box: RDG_ONE_DEVICE=1 puts every rank on cuda:0 and
    # RDG_DIST_BACKEND=gloo moves the collectives through the host (RCCL refuses two ranks on one device)
    backend = os.environ.get("RDG_DIST_BACKEND", "nccl")
    if os.environ.get("RDG_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RDG_FORCE_SHARD=1: run the sharded step on a 1-rank group too (single-GPU check of the collective path)
    force_shard = bool(os.environ.get("RDG_FORCE_SHARD")) and world == 1
    if world > 1 or force_shard:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from rodygs_amd.synthetic import synthetic_scene
    from rodygs_amd import _lib
    _lib.lib()

    P, W, H = args.points, args.width, args.height
    scene = synthetic_scene(P, W, H, 3, seed=777, variant=args.scene)
    target = synthetic_scene(max(P // 4, 1000), W, H, 3, seed=1234, variant=args.scene)
    if args.iteration in ("reference", "reference-v1"):
        if world != 1:
            raise SystemExit("--iteration reference is a single-GPU measurement")
        print(json.dumps(run_reference_iteration(args, dev)))
        return
    if args.loop > 0:
        if world != 1:
            raise SystemExit("--loop is a single-GPU measurement")
        print(json.dumps(run_loop(args, dev, scene, target)))
        return
    if force_shard:
        modes = ["shard"]
    elif world == 1:
        modes = ["single"]
    else:
        modes = ["allreduce", "shard"] if args.dp_mode == "both" else [args.dp_mode]
    runs = [run_mode(args, m, rank, world, dev, backend, scene, target) for m in modes]

    if rank == 0:
        best = min(runs, key=lambda r: r["dt"])
        sharded, dt, per_stage = best["sharded"], best["dt"], best["per_stage"]
        D, V, S, spatial_order = best["D"], best["V"], best["S"], best["spatial_order"]
        D_ref = best.get("D_ref", D)
        graph_replay = best["graph"]
        fps = args.steps * world / dt
        # dominant kernel: render backward.  Algorithmic bytes per launch (DESIGN.md §5 / SURVEY.md §8d):
        #   D*44 (sorted id + 40-B splat features) + H*W*40 (5 upstream-gradient channels, final_T, n_contrib, +pad
        #   as in the survey formula) + V*40 (10 accumulated floats per visible Gaussian)
        dom = "render_bwd"
        dom_ms = per_stage[dom]
        alg_bytes = D * 44 + H * W * 40 + V * 40
        # HBM traffic of that kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs, gfx950 correction applied); only valid for the workload it was taken on
        # and for the kernel source it was taken with (kernel_source_hash): a stale file gives null, not a wrong number
        traffic, traffic_note = None, "no PMC file for this workload"
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            wl = pmc["workload"]
            if (wl["points"], wl["width"], wl["height"], wl.get("scene", "uniform")) != (P, W, H, args.scene):
                traffic_note = f"profiles/{PMC_FILE} was taken on another workload"
            elif pmc.get("kernel_source_hash") != kernel_source_hash():
                traffic_note = (f"profiles/{PMC_FILE} was taken with kernel sources {pmc.get('kernel_source_hash')}, "
                                f"the build has {kernel_source_hash()}: re-run scripts/refresh_profiles.sh")
            else:
                traffic = pmc["kernels"]["rdg_render_bwd_kernel"]["hbm_bytes_corrected"]
                traffic_note = (f"profiles/{PMC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                f"kernel sources {pmc['kernel_source_hash']}")
        except Exception as e:
            traffic, traffic_note = None, f"profiles/{PMC_FILE} unreadable: {e}"
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # whole-step and per-stage algorithmic bytes for the algorithm that actually ran
        # (stage_bytes), and the FP32-VALU fraction of the two compositing kernels
        K = 16
        sh_adam_in_backward = bool(world == 1 and not sharded and not args.full_losses
                                   and os.environ.get("RDG_FUSE_SH_ADAM", "1") != "0")
        radix_binning = bool(best["radix"])
        P_eff = best["points_after_densify"] or P
        sb = stage_bytes(P_eff, K, V, D, H, W, world=world, sharded=sharded, sh_adam_in_backward=sh_adam_in_backward,
                         radix_binning=radix_binning, densify_stats=best["densify_stats"])
        sms = dict(per_stage)
        sms["binning"] = per_stage["scan_dup"] + per_stage["sort"] + per_stage["ranges"]
        stage_roofline = {k: {"algorithmic_bytes": b, "ms": sms[k],
                              "hbm_frac": (b / (sms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if sms[k] > 0 else None}
                          for k, b in sb.items()}
        b_step = sum(sb.values())
        # VALU-issue roofline of the two compositing kernels (they are issue-bound, not HBM-bound): wave instructions
        # per launch from the SQ counters (committed PMC pass, valid only for the kernel sources it was taken with) over
        # the kernel's SIMD-cycles = cycles per VALU instruction and SIMD, to be read against the measured issue classes
        valu = {"issue_classes_cycles_per_inst": VALU_ISSUE_CLASSES, "source": None}
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", SQ_FILE)))
            meta = sq.get("_meta", {})
            wl = meta.get("workload") or {}
            if meta.get("kernel_source_hash") != kernel_source_hash():
                valu["source"] = f"profiles/{SQ_FILE} is stale (kernel sources changed): no VALU-issue figure"
            elif (wl.get("points"), wl.get("width"), wl.get("height"), wl.get("scene", "uniform")) != (P, W, H, args.scene):
                valu["source"] = f"profiles/{SQ_FILE} was taken on another workload"
            else:
                valu["source"] = f"profiles/{SQ_FILE} (rocprofv3 --pmc SQ_INSTS_VALU ... pass of this command)"
                for stage, prefix in (("render_fwd", "void rdg_render_fwd_kernel"), ("render_bwd", "void rdg_render_bwd_kernel")):
                    row = next((v for k_, v in sq.items() if k_.startswith(prefix)), None)
                    if row is None or per_stage[stage] <= 0:
                        continue
                    # GRBM_GUI_ACTIVE counts per XCD (8 of them): the kernel's duration in shader-clock cycles
                    cycles = row["GRBM_GUI_ACTIVE"] / 8.0
                    cpi = cycles * N_SIMD / row["SQ_INSTS_VALU"]
                    valu[stage] = {"valu_wave_insts_per_launch": row["SQ_INSTS_VALU"],
                                   "salu_wave_insts_per_launch": row["SQ_INSTS_SALU"],
                                   "kernel_cycles": cycles, "cycles_per_valu_inst_per_simd": cpi,
                                   # every slot filled with the cheapest class would be 2.3: the fraction of the
                                   # issue-rate ceiling this kernel's instruction stream reaches
                                   "frac_of_issue_ceiling": VALU_ISSUE_CLASSES["f32 mul/add/fma/mov with VGPR sources"] / cpi}
        except Exception as e:
            valu["source"] = f"profiles/{SQ_FILE} unreadable: {e}"
        if world == 1:
            parallelism = "single GPU (frame-dp1)"
        elif sharded:
            parallelism = f"frame-dp{world}, Gaussian-sharded: splat records / gradient rows all-to-all (DESIGN.md §6)"
        else:
            parallelism = (f"frame-dp{world}, replicated cloud + RCCL all-reduce of Gaussian / pose gradients "
                           f"(BASELINE north_star formulation)")
        res = {
            "metric": metric_label(P, W, H),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_label(P, W, H, args.frames, args.full_losses, world, args.scene), "points": P,
                       "width": W, "height": H, "scene": args.scene,
                       "frames": args.frames,
                       # switches that shape the number: no read-back of the instance count inside the timed steps
                       # (capacity from the warm-up; an overflow renders that frame empty and raises afterwards), and
                       # whether every rank shares one device (functional check of the N > 1 flow, not a measurement)
                       "deferred_overflow_check": best["deferred"], "untimed_settle_steps": args.settle, "frame_order": "orbit positions, bit-reversed", "one_device": bool(os.environ.get("RDG_ONE_DEVICE")),
                       # True: the timed steps are replays of ONE captured hipGraph (trainstep.GraphedStep); the
                       # dominant kernel's avg_ms then comes from eager steps after the timed region
                       "graph_replay": graph_replay,
                       # the statistics of /root/reference/src/trainer/rodygs.py:316-341 are kept by every timed step
                       "densify_stats": best["densify_stats"],
                       # --densify-first: Gaussians after the one densify-and-prune that ran before the timed region
                       "points_after_densify": best["points_after_densify"],
                       "deterministic_backward": best["deterministic"], "render_normal": best["render_normal"],
                       "parallelism": parallelism,
                       # D = (tile, Gaussian) instances of the REFERENCE algorithm's rectangles on these frames (what its key
                       # stream holds); D_composited = the instances this build bins, sorts and composites (tight rectangles,
                       # RdgRasterSettings.cull: the rest cannot blend in any pixel of their tile); num_rendered_D = the count the
                       # byte formulas above are evaluated with (= D_composited)
                       "D": D_ref, "D_composited": D, "tight_tile_rectangles": best.get("cull", False),
                       "num_rendered_D": D, "visible_V": V,
                       "losses": "full (config 5 set)" if args.full_losses else "photometric",
                       # single-GPU photometric step: the per-Gaussian backward kernel applies the Adam update of the SH
                       # features itself (RDG_FUSE_SH_ADAM=0 restores the separate launch; same bits either way)
                       "sh_adam_in_backward": sh_adam_in_backward,
                       "binning": "radix" if radix_binning else "bucket",
                       "row_order": "z-curve of the canonical positions" if spatial_order else "generator (random)"},
            "gaussians_per_s": fps * (best["points_after_densify"] or P),
            "loss": best["loss"],
            "stage_ms": per_stage,
            "step_roofline": {"algorithmic_bytes_per_step": b_step, "achieved_GBps": b_step * fps / world / 1e9,
                              "frac_of_8TBps": b_step * fps / world / 1e9 / HBM_PEAK_GBPS,
                              "frac_of_6.3TBps_achievable": b_step * fps / world / 1e9 / 6300.0,
                              "pixel_splat_pairs_S": S},
            "stage_roofline": stage_roofline,
            "render_valu_issue": valu,
            "roofline": {"bound": "hbm", "kernel": "rdg_render_bwd_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_ms": dom_ms,
                         "note": "the compositing kernels are VALU-issue bound (render_valu_issue); the HBM fraction is "
                                 "reported because north_star asks for it",
                         "second_roofline": {"bound": "valu_issue", **(valu.get("render_bwd") or {}),
                                             "source": valu["source"]}},
        }
        if world > 1:
            # every formulation that was timed (each EXACTLY --steps steps between barriers, max over ranks)
            res["dp_modes"] = {r["mode"] + ("" if r["sharded"] == (r["mode"] == "shard") else " (fell back to replicated)"):
                               {"value": args.steps * world / r["dt"], "unit": "frames/s",
                                "ms_per_step": r["dt"] / args.steps * 1e3,
                                "wire": wire_bytes(P, 16, world, r["sharded"], args.frames)} for r in runs}
            # what the process group really is (the first hardware run of N > 1 should explain itself)
            res["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                    "rccl": (".".join(str(v) for v in torch.cuda.nccl.version())
                                             if dist.get_backend() == "nccl" else None),
                                    "devices_visible": torch.cuda.device_count(),
                                    "one_device": bool(os.environ.get("RDG_ONE_DEVICE"))}
        if not args.no_cpu_baseline and world == 1:      # contract: the CPU leg runs at N = 1 only
            try:
                if best.get("bench_frame") is not None:
                    frame, gt = best["bench_frame"]
                    hip = hip_frame(frame, gt, 3, dev)
                else:                                    # (RDG_FORCE_SHARD: no replica to capture from) the canonical cloud
                    frame, gt, hip = scene, torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)), None
                res["cpu_baseline"], res["parity_check"] = cpu_baseline(frame, gt, hip, 3, dev, budget_s=args.cpu_budget,
                                                                        threads=args.cpu_threads)
            except Exception as e:  # the baseline must never take the GPU number down with it
                n_cpu, model = _host_cpu()
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(),
                                       "host": {"cpu_count": n_cpu, "cpu_model": model},
                                       "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
        # the default headline run (BASELINE configs[2], N = 1, no mode flag) also carries the compact records of what DESIGN.md
        # section 5 claims next to the step: the loop, the reference's iteration, the 100 k graph step, the PSNR delta -- child runs
        # after everything above, outside every timed region of this process (its GPU memory is released first)
        default_run = (world == 1 and (P, W, H, args.scene) == (1000000, 1920, 1080, "uniform") and not args.full_losses
                       and not args.graph and not args.no_normal and not args.densify_first and not args.no_cpu_baseline
                       and not force_shard)
        if default_run and not args.no_sub_records:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            try:
                res["sub_records"] = sub_records()
                pd = res["sub_records"].get("psnr_delta", {})
                if "drift_db" in pd:
                    res["psnr_delta_db"] = pd["drift_db"]          # BASELINE.json metric: "... PSNR delta vs ref"
            except Exception as e:                            # noqa: BLE001
                res["sub_records"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if default_run and not args.no_live_pmc:
            # roofline.traffic and the VALU-issue figures from the counters of THIS box and THIS build (the committed
            # profiles/ files stay the fallback, labelled as such)
            tables, note = live_pmc(P, W, H, args.scene)
            if tables is not None:
                def hbm(k):
                    return 2 * tables["FETCH_SIZE"].get(k, (0.0, 0))[0] * 1024 + tables["WRITE_SIZE"].get(k, (0.0, 0))[0] * 1024
                live = hbm("rdg_render_bwd_kernel")
                res["roofline"]["traffic"], res["roofline"]["traffic_source"] = live, note
                res["roofline"]["traffic_over_algorithmic"] = live / alg_bytes
                res["hbm_traffic_per_kernel_live"] = {k: hbm(k) for k in sorted(set(tables["FETCH_SIZE"]) | set(tables["WRITE_SIZE"]))}
                lv = {"issue_classes_cycles_per_inst": VALU_ISSUE_CLASSES, "source": note}
                for stage, k in (("render_fwd", "rdg_render_fwd_kernel"), ("render_bwd", "rdg_render_bwd_kernel")):
                    try:
                        insts, salu = tables["SQ_INSTS_VALU"][k][0], tables["SQ_INSTS_SALU"][k][0]
                        cycles = tables["GRBM_GUI_ACTIVE"][k][0] / 8.0       # counted per XCD (8 of them)
                        cpi = cycles * N_SIMD / insts
                        lv[stage] = {"valu_wave_insts_per_launch": insts, "salu_wave_insts_per_launch": salu,
                                     "kernel_cycles": cycles, "cycles_per_valu_inst_per_simd": cpi,
                                     "frac_of_issue_ceiling": VALU_ISSUE_CLASSES["f32 mul/add/fma/mov with VGPR sources"] / cpi}
                    except (KeyError, ZeroDivisionError):
                        pass
                if "render_bwd" in lv:
                    res["render_valu_issue"] = lv
                    res["roofline"]["second_roofline"] = {"bound": "valu_issue", **lv["render_bwd"], "source": note}
            else:
                res["roofline"]["traffic_live_note"] = note
    # The JSON line is the LAST line on stdout: RCCL writes its version banner through C stdio, which sits in a buffer until the
    # process exits when stdout is a pipe -- after Python's print.  Every rank flushes C stdio, the ranks meet, then rank 0 prints.
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
