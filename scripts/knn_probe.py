"""Times knn_points(p, p, K = 8) (pytorch3d call of RigidityLoss) on a uniform random cloud and on a clustered one."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rodygs_amd import knn as KN
n = int(sys.argv[1])
g = torch.Generator().manual_seed(1)
clouds = {"uniform": torch.rand(n, 3, generator=g) * torch.tensor([8.0, 5.0, 12.0]),
          "clustered": torch.randn(n, 3, generator=g) * torch.rand(n, 1, generator=g) ** 3 * 4.0}
for name, p in clouds.items():
    p = p.cuda()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(4):
        ev[0].record(); res = KN.knn_points(p[None], p[None], K=8); ev[1].record(); torch.cuda.synchronize()
    print(f"n={n} {name}: knn_points {ev[0].elapsed_time(ev[1]):.3f} ms, checksum idx {int(res.idx.sum())} d2 {float(res.dists.double().sum()):.9e}")
