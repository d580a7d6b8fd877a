"""Timing probe: knn_points (self query, K=8) and distCUDA2 at RigidityLoss / init sizes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rodygs_amd.knn import knn_points, distCUDA2
from oracle import rasterizer_oracle as O
from rodygs_amd.synthetic import synthetic_scene
for n in (100_000, 500_000, 1_000_000):
    sc = synthetic_scene(n, 1920, 1080, 3, seed=1)
    p = sc["means3D"].cuda()
    for name, fn in (("knn_points K=8 self", lambda: knn_points(p[None], p[None], K=8)), ("distCUDA2", lambda: distCUDA2(p))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        print(f"{name:22s} N={n:8d}  {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
