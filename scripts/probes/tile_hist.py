"""Histogram of the tile-list lengths of the bench frames (how many lists exceed 512 / 1024)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_stages as HS
from rodygs_amd.synthetic import synthetic_scene
sc = synthetic_scene(1000000, 1920, 1080, 3, seed=777)
hs = HS.run_stages(sc, 3)
n = (hs["ranges"][:, 1].astype(np.int64) - hs["ranges"][:, 0])
print("tiles", len(n), "mean", n.mean(), "max", n.max(), "<=64", (n <= 64).mean(), "<=128", (n <= 128).mean(), "<=256", (n <= 256).mean(), "<=512", (n <= 512).mean(), "<=1024", (n <= 1024).mean())
print("instances in lists <=512:", n[n <= 512].sum() / n.sum())
