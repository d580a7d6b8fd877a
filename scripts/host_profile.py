"""cProfile of the eager photometric train step at a host-bound size (100 k Gaussians): where the Python time goes."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rodygs_amd import rasterizer, synthetic  # noqa: E402
from rodygs_amd.trainstep import DynamicScene  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sc = synthetic.synthetic_scene(P, 1920, 1080, 3, seed=777)
tgt = synthetic.synthetic_scene(P // 2, 1920, 1080, 3, seed=778)
ds = DynamicScene(sc, num_frames=100, device="cuda", spatial_order=True)
perm = list(range(0, 100, 6))
ds.make_ground_truth(tgt, perm)
for s in range(10):
    ds.train_step(s, perm=perm)
rasterizer.DEFERRED_OVERFLOW_CHECK = True
for s in range(10, 40):
    ds.train_step(s, perm=perm)
torch.cuda.synchronize()
# the backward functions run on the autograd engine's thread (invisible to cProfile): wall-clock them one by one
import collections
import time as _t
import rodygs_amd.deform as _D
import rodygs_amd.losses as _Lo
import rodygs_amd.model_ops as _M
BW = collections.defaultdict(float)


def _wrap(cls):
    orig = cls.backward

    def timed(ctx, *a):
        t = _t.perf_counter()
        r = orig(ctx, *a)
        BW[cls.__name__] += _t.perf_counter() - t
        return r
    cls.backward = staticmethod(timed)


for mod in (rasterizer, _D, _Lo, _M):
    for name in dir(mod):
        obj = getattr(mod, name)
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            _wrap(obj)
pr = cProfile.Profile()
pr.enable()
import time
t0 = time.perf_counter()
for s in range(40, 340):
    ds.train_step(s, perm=perm)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print(f"{dt / 300 * 1e3:.4f} ms per step (with the profiler on)")
for k, v in sorted(BW.items(), key=lambda kv: -kv[1]):
    print(f"  backward of {k:28s} {v / 300 * 1e6:7.1f} us per step")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
