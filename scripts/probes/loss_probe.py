"""loss forward / backward kernel times at 1080p (hipEvents around 200 launches each)."""
import sys, torch
sys.path.insert(0, ".")
from rodygs_amd import _lib
from rodygs_amd.losses import fused_photometric_loss
L = _lib.lib()
H, W = 1080, 1920
g = torch.Generator().manual_seed(1)
img = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
gt = torch.rand(3, H, W, generator=g).cuda()
for _ in range(5):
    fused_photometric_loss(img, gt, 0.2).backward()
torch.cuda.synchronize()
_lib.timing_enable(True, ["loss_fwd", "loss_bwd"])
_lib.timing_reset()
for _ in range(200):
    fused_photometric_loss(img, gt, 0.2).backward()
torch.cuda.synchronize()
t = _lib.stage_times()
print({k: round(1e3 * t[k][0] / max(t[k][1], 1), 1) for k in ("loss_fwd", "loss_bwd")}, "us")
