"""Does the rows' Adam launch overlap the MLP backward?  One-step kernel timeline (rocprofv3 --kernel-trace) for both settings."""
import sys, torch
sys.path.insert(0, ".")
import bench
from rodygs_amd.trainstep import DynamicScene
dev = torch.device("cuda:0")
scene = bench.make_scene(1_000_000, 3, seed=777) if hasattr(bench, "make_scene") else None
print("make_scene" if scene is not None else "no make_scene")
