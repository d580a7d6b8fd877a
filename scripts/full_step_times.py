"""Wall time of each step of the config-5 loss set (which steps are expensive, and how much of a step is host time)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rasterizer_oracle as O
from rodygs_amd.synthetic import synthetic_scene
from rodygs_amd.trainstep import DynamicScene
P, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 1920, 1080
sc = synthetic_scene(P, W, H, 3, seed=777)
ds = DynamicScene(sc, num_frames=100, device="cuda", full_losses=True)
fr = list(range(0, 100, 12))
ds.make_ground_truth(synthetic_scene(P // 4, W, H, 3, seed=1234), fr)
for s in range(6):
    ds.train_step(s, 0, 1, fr)
out = []
for s in range(10, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds.train_step(s, 0, 1, fr)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); out.append((s, round(th * 1e3, 2), round((time.perf_counter() - t0) * 1e3, 2)))
print(json.dumps({"step, host ms, total ms": out}))
