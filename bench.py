"""bench.py -- the driver's measurement contract for the rodygs_amd hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): 1 M dynamic Gaussians + time-deformation
MLP, 1920x1080, SH degree 3, 100-frame synthetic video (SURVEY.md §8d generator, seed 777).  A "step" is one
full train step on one camera per GPU: deformation -> rasterize forward -> 0.8 L1 + 0.2 D-SSIM -> backward ->
fused Adam over every parameter.  Inputs are resident in HBM before the timed region.  Weak scaling: every GPU
renders its own frame, value = frames (train steps x GPUs) per second.  N > 1 (--dp-mode): "shard" (default) keeps
the Gaussians sharded over the ranks and exchanges 64-byte splat records / gradient rows with two all-to-alls per
step (rodygs_amd/sharded.py); "allreduce" replicates the cloud and all-reduces the flat gradient bucket (RCCL,
overlapped with backward and Adam).

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

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3  # same guide: FP32 vector peak


def cpu_baseline(scene, sh_degree, sample_tiles=96, max_seconds=40.0):
    """Oracle (kind "port") on a bounded sample: full per-Gaussian stage + binning for the frame, then forward +
    backward compositing of `sample_tiles` evenly spaced tiles, extrapolated to the whole image."""
    from oracle import rasterizer_oracle as O   # checker / baseline only -- never on the product path
    P, H, W = scene["means3D"].shape[0], scene["H"], scene["W"]
    st = O.OracleSettings(H, W, scene["tanfovx"], scene["tanfovy"], torch.zeros(3), 1.0, scene["projmatrix"], sh_degree)
    names = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")
    ins = {k: scene[k].clone().requires_grad_(True) for k in names}
    m2 = torch.zeros(P, 3, requires_grad=True)
    t0 = time.perf_counter()
    geom = O.preprocess(ins["means3D"], m2, ins["opacities"], ins["viewmatrix"], st, shs=ins["shs"],
                        scales=ins["scales"], rotations=ins["rotations"])
    binning = O.bin_and_sort(geom)
    t_pre = time.perf_counter() - t0
    gx, gy = geom["grid"]
    n_tiles = gx * gy
    stride = max(1, n_tiles // sample_tiles)
    subset = list(range(stride // 2, n_tiles, stride))[:sample_tiles]
    t1 = time.perf_counter()
    img = O.render_tiles(geom, binning, st.bg, H, W, tile_subset=subset)
    loss = img["color"].sum() + 0.1 * img["depth"].sum()
    loss.backward()
    t_tiles = time.perf_counter() - t1
    pairs = int(sum((binning["ranges"][t, 1] - binning["ranges"][t, 0]) for t in subset))
    total_pairs = int(binning["num_rendered"])
    # tile time scales with the splat instances in the tile, not with the tile count
    est = t_pre + t_tiles * (total_pairs / max(pairs, 1))
    return {"value": 1.0 / est, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fwd+bwd of one frame: full preprocess+binning of {P} Gaussians ({t_pre:.1f}s) + "
                      f"{len(subset)}/{n_tiles} tiles holding {pairs}/{total_pairs} splat instances ({t_tiles:.1f}s), "
                      f"extrapolated by instances; rasterizer only (no MLP/loss/Adam)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--gt-frames", type=int, default=16, help="distinct frames with ground truth resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-losses", action="store_true",
                    help="config-5 loss set (depth, motion regularisers, rigidity every 5th step) instead of the "
                         "photometric-only step the headline metric is quoted on")
    ap.add_argument("--dp-mode", choices=["shard", "allreduce"], default=os.environ.get("RDG_DP_MODE", "shard"),
                    help="N > 1: 'shard' = Gaussians sharded over the ranks, splat records / gradient rows exchanged "
                         "with two all-to-alls (rodygs_amd/sharded.py); 'allreduce' = replicated cloud, overlapped "
                         "bucketed all-reduce of the 75-float-per-Gaussian gradient")
    args = ap.parse_args()

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

    from oracle import rasterizer_oracle as O          # synthetic-scene generator + cpu_baseline only
    from rodygs_amd import _lib
    from rodygs_amd.trainstep import DynamicScene
    _lib.lib()

    P, W, H = args.points, args.width, args.height
    scene = O.synthetic_scene(P, W, H, 3, seed=777)
    target = O.synthetic_scene(max(P // 4, 1000), W, H, 3, seed=1234)
    ds = DynamicScene(scene, num_frames=args.frames, sh_degree=3, device=dev, seed=777, full_losses=args.full_losses)
    n_gt = min(args.gt_frames * world, args.frames)
    gt_frames = [int(round(i * args.frames / n_gt)) % args.frames for i in range(n_gt)]
    gt_frames = sorted(set(gt_frames))
    ds.make_ground_truth(target, gt_frames)
    perm = gt_frames
    sharded = (world > 1 or force_shard) and args.dp_mode == "shard"
    if sharded:
        from rodygs_amd.sharded import HostStagedExchange, ShardedDynamicScene
        try:
            ss = ShardedDynamicScene.from_replica(ds, rank, world, None if backend == "nccl" else HostStagedExchange())
        except (NotImplementedError, ValueError) as e:
            # a configuration the sharded step does not cover (decided from sizes every rank shares, so all ranks take
            # this branch together): the replicated formulation runs instead
            if rank == 0:
                print(f"bench.py: sharded frame-DP unavailable ({e}); using --dp-mode allreduce", file=sys.stderr)
            sharded = False
    if sharded:
        ds.fp = ds.sync = ds.m2 = None           # the replica's full-size buffers are not needed any more
        torch.cuda.empty_cache()
        train_step = lambda st_: ss.train_step(st_, perm)                       # noqa: E731
    else:
        train_step = lambda st_: ds.train_step(st_, rank, world, perm)          # noqa: E731

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        train_step(step)
        step += 1
    from rodygs_amd import rasterizer
    # After the warm-up the instance count D of every frame is known to within a few percent: stop reading it back
    # inside the forward (no host wait in the step).  Capacity is 1.25x the last D; an overflow would render that
    # frame empty and raise RasterizerCapacityOverflow at the next forward / at the final poll below.
    rasterizer.DEFERRED_OVERFLOW_CHECK = True
    # Inside the timed region only the dominant kernel is bracketed by hipEvents (every timed stage costs ~10 us of
    # stream gap); the per-stage table is taken from a few extra steps afterwards.
    _lib.timing_enable(True, stages=["render_bwd"])
    _lib.timing_reset()
    # the interpreter's cyclic collector runs a full (generation-2) pass once the start-up garbage has piled up -- a
    # 40 ms host stall that would land somewhere in a 20-step window: collect now, and keep the survivors out of later
    # passes
    import gc
    gc.collect()
    gc.freeze()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(step)
        step += 1
    sync()
    dt = time.perf_counter() - t0
    rasterizer.poll_overflow(block=True)
    dom = _lib.stage_times()["render_bwd"]
    _lib.timing_enable(True)
    _lib.timing_reset()
    for _ in range(min(5, args.steps)):
        train_step(step)
        step += 1
    sync()
    rasterizer.poll_overflow(block=True)
    stages = _lib.stage_times()
    stages["render_bwd"] = dom
    _lib.timing_enable(False)
    rasterizer.DEFERRED_OVERFLOW_CHECK = False
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        if sharded:
            D = int(rasterizer._CAPACITY_HINT.get(ss.key, 0))
            V = ss.visible_count()
        else:
            D = int(rasterizer._CAPACITY_HINT.get((P, H, W), 0))
            with torch.no_grad():
                out, _ = ds.render(perm[0])
                V = int((out[4] > 0).sum().item())
        fps = args.steps * world / dt
        per_stage = {k: (ms / n if n else 0.0) for k, (ms, n) in stages.items()}
        # dominant kernel: render backward.  Algorithmic bytes per launch (DESIGN.md §5 / SURVEY.md §8d):
        #   D*44 (sorted id + 40-B splat features) + H*W*40 (5 upstream-gradient channels, final_T, n_contrib, +pad
        #   as in the survey formula) + V*40 (10 accumulated floats per visible Gaussian)
        dom = "render_bwd"
        dom_ms = per_stage[dom]
        alg_bytes = D * 44 + H * W * 40 + V * 40
        # HBM traffic of that kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs, gfx950 correction applied); only valid for the workload it was taken on
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
            wl = pmc["workload"]
            if (wl["points"], wl["width"], wl["height"]) == (P, W, H):
                traffic = pmc["kernels"]["rdg_render_bwd_kernel"]["hbm_bytes_corrected"]
        except Exception:
            traffic = None
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # whole-step and per-stage algorithmic bytes (the survey's own formula, radix-sort term and
        # all, so the number is comparable across builds) and the FP32-VALU fraction of the two compositing kernels
        with torch.no_grad():
            _, n_contrib = rasterizer.last_compositing_state()
            S = int(n_contrib.sum(dtype=torch.int64).item())
        K, tiles = 16, ((W + 15) // 16) * ((H + 15) // 16)
        n_pass = (32 + max(tiles - 1, 1).bit_length() + 7) // 8
        sb = {
            "preprocess": P * (44 + 12 * K) + V * 48 + P * 8,
            "binning": D * 12 + D * 24 * n_pass + D * 8 + tiles * 8,
            "render_fwd": D * 44 + H * W * 40,
            "render_bwd": alg_bytes,
            "preprocess_bwd": P * (44 + 12 * K) * 2 + V * 48,
            "deform_fwd": P * 96, "deform_bwd": P * 96,
            # Gaussian-sharded frame-DP: a rank's optimiser state covers its slice only
            "adam": 28 * (59 + 16) * (P // world if sharded else P),
        }
        sms = dict(per_stage)
        sms["binning"] = per_stage["scan_dup"] + per_stage["sort"] + per_stage["ranges"]
        stage_roofline = {k: {"algorithmic_bytes": b, "ms": sms[k],
                              "hbm_frac": (b / (sms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if sms[k] > 0 else None}
                          for k, b in sb.items()}
        b_step = sum(sb.values())
        valu = {k: {"flops": S * f, "tflops": S * f / (per_stage[k] * 1e-3) / 1e12,
                    "frac_of_fp32_vector_peak": S * f / (per_stage[k] * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS}
                for k, f in (("render_fwd", 22), ("render_bwd", 60)) if per_stage[k] > 0}
        res = {
            "metric": "train-step fps at 1M dynamic Gaussians / 1080p (fwd+bwd+Adam, one camera per GPU per step)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{P} dynamic Gaussians + deformation MLP, {W}x{H}, SH3, {args.frames}-frame "
                                   f"synthetic video (BASELINE configs[2])", "points": P, "width": W, "height": H,
                       "frames": args.frames,
                       "parallelism": (f"frame-dp{world}, Gaussian-sharded (records/gradient rows all-to-all)" if sharded
                                       else f"frame-dp{world}" + (", replicated + bucketed all-reduce" if world > 1 else "")), "num_rendered_D": D, "visible_V": V,
                       "losses": "full (config 5 set)" if args.full_losses else "photometric",
                       # single-GPU photometric step: the per-Gaussian backward kernel applies the Adam update of the SH
                       # features itself (RDG_FUSE_SH_ADAM=0 restores the separate launch; same bits either way)
                       "sh_adam_in_backward": bool(world == 1 and not sharded and not args.full_losses
                                                   and os.environ.get("RDG_FUSE_SH_ADAM", "1") != "0")},
            "gaussians_per_s": fps * P,
            "loss": float(loss.item()),
            "stage_ms": per_stage,
            "step_roofline": {"algorithmic_bytes_per_step": b_step, "achieved_GBps": b_step * fps / world / 1e9,
                              "frac_of_8TBps": b_step * fps / world / 1e9 / HBM_PEAK_GBPS,
                              "frac_of_6.3TBps_achievable": b_step * fps / world / 1e9 / 6300.0,
                              "pixel_splat_pairs_S": S},
            "stage_roofline": stage_roofline,
            "render_valu": valu,
            "roofline": {"bound": "hbm", "kernel": "rdg_render_bwd_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "avg_ms": dom_ms,
                         "note": "compositing kernels are VALU/exp/LDS-bound (SURVEY.md §8d); the HBM fraction is "
                                 "reported because north_star asks for it"},
        }
        if not args.no_cpu_baseline and world == 1:      # contract: the CPU leg runs at N = 1 only
            try:
                res["cpu_baseline"] = cpu_baseline(scene, 3)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(),
                                       "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
