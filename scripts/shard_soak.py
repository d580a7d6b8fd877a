"""Long run of the Gaussian-sharded step with 8 virtual ranks and periodic densification (stability check)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rasterizer_oracle as O
from rodygs_amd.synthetic import synthetic_scene
from rodygs_amd.sharded import ShardedDynamicScene, run_virtual_densify, run_virtual_step
from rodygs_amd.trainstep import DynamicScene
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = synthetic_scene(40000, 480, 272, 3, seed=5)
ds = DynamicScene(sc, num_frames=24, device="cuda")
ds.make_ground_truth(synthetic_scene(12000, 480, 272, 3, seed=6), range(24))
sh = [ShardedDynamicScene.from_replica(ds, r, W, exchange=object()) for r in range(W)]
for s in sh:
    s.track_densification()
hist, sizes = [], []
for step in range(600):
    hist.append(float(np.mean([float(x) for x in run_virtual_step(sh, step, list(range(24)))])))
    if step in (150, 300, 450):
        info = run_virtual_densify(sh, max_grad=1e-5, min_opacity=0.02, percent_dense=0.002)
        sizes.append((step, info[0]["P"], [s.n for s in sh], sh[0].stride))
assert all(np.isfinite(hist))
print(json.dumps({"world": W, "loss_first10": float(np.mean(hist[:10])), "loss_last10": float(np.mean(hist[-10:])),
                  "densify": sizes, "mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
