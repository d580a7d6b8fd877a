"""Where do the MLP gradients of trainstep.ReferenceIteration and of the framework-op flow part ways?  (debug probe)"""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rasterizer_oracle as O
from rodygs_amd.trainstep import ReferenceIteration
from rodygs_amd.deform import gaussian_deformation_packed
DEV = "cuda"
W, H, T = 208, 144, 6
ri = ReferenceIteration(O.synthetic_scene(3000, W, H, 3, seed=3), O.synthetic_scene(4000, W, H, 3, seed=4), num_frames=T, device=DEV)
net = copy.deepcopy(ri.net)
for (n1, p1), (n2, p2) in zip(ri.net.named_parameters(), net.named_parameters()):
    assert torch.equal(p1, p2), n1
g = torch.Generator().manual_seed(0)
gd = torch.randn(4000, 3, generator=g).to(DEV), torch.randn(4000, 4, generator=g).to(DEV)
coeff = ri.fp_d["motion_coeff"].detach().clone().requires_grad_(True)
c2 = coeff.detach().clone().requires_grad_(True)
keep = {}
for tag, nn_, c, hip in (("hip", ri.net, coeff, True), ("torch", net, c2, False)):
    allb = nn_.motion_basis(ri.emb_rows[1])
    allb.retain_grad()
    if hip:
        dx, dr = gaussian_deformation_packed(c, ri.time_ind, allb, 5.0)
    else:
        table, bt = allb[:-1], allb[-1]
        delta = (c.reshape(-1, 1, 16) @ (bt.unsqueeze(0) - table[ri.time_ind])).squeeze(1)
        dx, dr = delta[:, :3] * 5.0, delta[:, 3:]
    ((dx * gd[0]).sum() + (dr * gd[1]).sum()).backward()
    keep[tag] = (allb.grad.clone(), torch.cat([p.grad.flatten() for p in nn_.parameters()]), dx.detach(), c.grad.clone())
a, b = keep["hip"], keep["torch"]
for i, name in enumerate(("d_allb", "MLP grads", "dxyz fwd", "d_coeff")):
    d = (a[i] - b[i]).abs().max().item(); s = b[i].abs().max().item()
    print(f"{name}: max abs diff {d:.3e} scale {s:.3e} rel {d / s:.3e}")
# the MLP backward alone: the same upstream gradient through both nets
for nn_ in (ri.net, net):
    for p in nn_.parameters():
        p.grad = None
up = torch.randn(T + 1, 16, 7, generator=g).to(DEV)
outs = []
for nn_ in (ri.net, net):
    nn_.motion_basis(ri.emb_rows[1]).backward(up)
    outs.append(torch.cat([p.grad.flatten() for p in nn_.parameters()]))
print("same upstream through both nets: max abs diff", (outs[0] - outs[1]).abs().max().item(), "scale", outs[1].abs().max().item())
