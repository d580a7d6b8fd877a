"""Frame-DP and image quality ("batch-N optimisation changes the trajectory -- PSNR must be measured").

Optimises the SAME dynamic scene (same initial cloud, MLP, poses, ground-truth video) for the SAME number of optimiser
steps
  (a) on one GPU, one camera per step                          (trainstep.DynamicScene), and
  (b) with N-way Gaussian-sharded frame-DP, N cameras per step  (sharded.ShardedDynamicScene, N virtual ranks in this
      process: the arithmetic of the N-GPU step, exchanges as block copies),
then scores both on every frame of the video (PSNR as /root/reference/src/utils/eval_utils.py:36-39).

    python scripts/dp_psnr.py --world 8 --steps 300 --out profiles/r01_dp_psnr.json
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--points", type=int, default=60000)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=272)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from oracle import rasterizer_oracle as O                       # synthetic-scene generator only
    from rodygs_amd.synthetic import synthetic_scene
    from rodygs_amd.checkpoint import psnr
    from rodygs_amd.sharded import ShardedDynamicScene, run_virtual_step
    from rodygs_amd.trainstep import DynamicScene
    dev = torch.device("cuda", 0)
    sc = synthetic_scene(a.points, a.width, a.height, 3, seed=777)
    tgt = synthetic_scene(a.points // 3, a.width, a.height, 3, seed=1234)
    perm = list(range(a.frames))

    def fresh():
        ds = DynamicScene(sc, num_frames=a.frames, device=dev, seed=777)
        ds.make_ground_truth(tgt, perm)
        return ds

    def score(ds):
        with torch.no_grad():
            vals = [float(psnr(ds.render(f)[0][0].clamp(0, 1), ds.gt[f])) for f in perm]
        return sum(vals) / len(vals)

    single = fresh()
    p0 = score(single)
    for s in range(a.steps):
        single.train_step(s, 0, 1, perm)
    p_single = score(single)
    for s in range(a.steps, a.steps * a.world):                      # ... and on to the same number of cameras seen
        single.train_step(s, 0, 1, perm)
    p_single_views = score(single)

    rep = fresh()
    shards = [ShardedDynamicScene.from_replica(rep, r, a.world, exchange=object()) for r in range(a.world)]
    for s in range(a.steps):
        run_virtual_step(shards, s, perm)
    with torch.no_grad():                                            # slices back into the replica for scoring
        for k in rep.fp.names:
            rep.fp[k].copy_(torch.cat([sh.fp[k].detach() for sh in shards]))
        rep.sp.flat.copy_(shards[0].sp.flat)
    p_dp = score(rep)
    res = {"workload": f"{a.points} dynamic Gaussians, {a.width}x{a.height}, {a.frames}-frame synthetic video, "
                       f"{a.steps} optimiser steps, photometric loss", "psnr_initial_db": p0,
           "psnr_single_gpu_1_camera_per_step_db": p_single,
           f"psnr_sharded_dp{a.world}_{a.world}_cameras_per_step_db": p_dp, "delta_db": p_dp - p_single,
           f"psnr_single_gpu_after_{a.steps * a.world}_steps_db": p_single_views,
           "cameras_seen": {"single": a.steps, f"dp{a.world}": a.steps * a.world}}
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
