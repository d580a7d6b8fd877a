"""Times rdg_rigidity_dp_forward alone at config-5 sizes (n sampled Gaussians, nt times, K = 8)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rodygs_amd import knn as KN
from rodygs_amd.rigidity import _FusedDistancePreserving
n, nt = int(sys.argv[1]), 25
g = torch.Generator().manual_seed(1)
pts = (torch.rand(n, 3, generator=g) * torch.tensor([8.0, 5.0, 12.0])).cuda()
res = KN.knn_points(pts[None], pts[None], K=8)
pos_t = (pts[None] + 0.05 * torch.randn(nt, n, 3, generator=g).cuda()).requires_grad_(True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = _FusedDistancePreserving.apply(pos_t, res.idx[0], res.dists[0], 1e-6)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"n={n} fused distance-preserving forward incl. host glue: {(t1 - t0) * 1e3:.2f} ms, value {float(out):.6e}")
