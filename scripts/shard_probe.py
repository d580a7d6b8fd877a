"""Single-GPU cost model of the Gaussian-sharded step (rodygs_amd/sharded.py): `--world` virtual ranks of the bench
workload in one process; prints GPU and host milliseconds per phase of ONE rank (rank 0).  The real step on N GPUs is
owner_fwd + camera + owner_bwd + update plus the two all-to-alls and the small all-reduce.

    python scripts/shard_probe.py --world 8 [--points 1000000 --width 1920 --height 1080]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from oracle import rasterizer_oracle as O
    from rodygs_amd.synthetic import synthetic_scene
    from rodygs_amd.sharded import ShardedDynamicScene
    from rodygs_amd.trainstep import DynamicScene
    dev = torch.device("cuda", 0)
    sc = synthetic_scene(a.points, a.width, a.height, 3, seed=777)
    tgt = synthetic_scene(max(a.points // 4, 1000), a.width, a.height, 3, seed=1234)
    ds = DynamicScene(sc, num_frames=100, device=dev)
    frames = list(range(0, 100, 6))
    ds.make_ground_truth(tgt, frames)
    W = a.world
    shards = [ShardedDynamicScene.from_replica(ds, r, W, exchange=object()) for r in range(W)]
    t_single = []
    for s_ in range(3 + a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds.train_step(s_, 0, 1, frames)
        torch.cuda.synchronize()
        t_single.append((time.perf_counter() - t0) * 1e3)
    phases = ("owner_fwd", "camera", "owner_bwd", "update")
    gpu = {k: 0.0 for k in phases}
    host = {k: 0.0 for k in phases}

    def timed(name, fn, rec):
        if not rec:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        r = fn()
        host[name] += (time.perf_counter() - t0) * 1e3
        e1.record()
        torch.cuda.synchronize()
        gpu[name] += e0.elapsed_time(e1)
        return r

    n = 0
    for step in range(3 + a.steps):
        rec = step >= 3
        n += rec
        for i, s in enumerate(shards):
            timed("owner_fwd", lambda s=s: s.phase_owner_forward(step, frames), rec and i == 0)
        for c in range(W):
            for s in range(W):
                shards[c].rec_cam.view(W, -1)[s].copy_(shards[s].rec_own.view(W, -1)[c])
        for i, s in enumerate(shards):
            timed("camera", s.phase_camera, rec and i == 0)
        for s in range(W):
            for c in range(W):
                shards[s].row_own.view(W, -1)[c].copy_(shards[c].row_cam.view(W, -1)[s])
        for i, s in enumerate(shards):
            timed("owner_bwd", lambda s=s: (s.phase_owner_backward(), s.phase_small_backward(step)), rec and i == 0)
        total = torch.stack([s.sp.flat_grad for s in shards]).sum(0)
        for i, s in enumerate(shards):
            s.sp.flat_grad.copy_(total)
            timed("update", s.phase_update, rec and i == 0)
    out = {"world": W, "points": a.points, "single_gpu_step_ms": sum(t_single[3:]) / a.steps,
           "gpu_ms": {k: v / n for k, v in gpu.items()}, "host_ms": {k: v / n for k, v in host.items()},
           "rank_gpu_ms_without_comm": sum(gpu.values()) / n, "rank_host_ms": sum(host.values()) / n,
           "exchange_bytes_per_gpu_per_step": 2 * shards[0].rows * 64 * (W - 1) // W}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
