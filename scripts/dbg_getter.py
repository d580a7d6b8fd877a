import torch, sys
sys.path.insert(0, '.')
from rodygs_amd import _lib
from rodygs_amd.deform import _birth_order, dynamic_gaussians
DEV='cuda'
L=_lib.lib()
NV=3
g = torch.Generator().manual_seed(5 + NV)
P, Tu, stride = 5003, 12, 5120
rnd = lambda *sh: torch.randn(*sh, generator=g).to(DEV)
xyz, scaling, rotation, opacity, coeff = rnd(P, 3), 0.3 * rnd(P, 3), rnd(P, 4), rnd(P, 1), 0.2 * rnd(P, 16)
ti = torch.randint(0, Tu, (P,), generator=g).to(DEV)
table, bt = 0.1 * rnd(Tu, 16, 7), 0.1 * rnd(NV, 16, 7)
bases_all = torch.cat([table.unsqueeze(0).expand(NV, -1, -1, -1), bt.unsqueeze(1)], dim=1).contiguous()
gm, gs_, gr, go = rnd(NV, stride, 3), rnd(NV, stride, 3), rnd(NV, stride, 4), rnd(NV, stride, 1)
f32 = dict(dtype=torch.float32, device=DEV)
d = {k: torch.zeros_like(t) for k, t in (("xyz", xyz), ("scaling", scaling), ("rotation", rotation), ("opacity", opacity), ("coeff", coeff))}
d_bases = torch.zeros_like(bases_all)
order, inv, _ = _birth_order(ti)
sws = torch.empty(L.rdg_deform_sorted_views_ws_bytes(P, NV), dtype=torch.uint8, device=DEV)
_lib.check(L.rdg_dyn_getter_views_backward(P, Tu, NV, stride, coeff.data_ptr(), ti.data_ptr(), bases_all.data_ptr(),
                                           5.0, scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(),
                                           gm.data_ptr(), gs_.data_ptr(), gr.data_ptr(), go.data_ptr(),
                                           d["xyz"].data_ptr(), d["scaling"].data_ptr(), d["rotation"].data_ptr(),
                                           d["opacity"].data_ptr(), d["coeff"].data_ptr(), d_bases.data_ptr(),
                                           order.data_ptr(), inv.data_ptr(), sws.data_ptr(), _lib.stream_ptr()), "views bwd")
for v in range(NV):
    leaves = [t.clone().requires_grad_(True) for t in (xyz, scaling, rotation, opacity, coeff)]
    bv = bases_all[v].clone().requires_grad_(True)
    o = dynamic_gaussians(*leaves, ti, bv, 5.0)
    torch.autograd.backward(o, [gm[v, :P], gs_[v, :P], gr[v, :P], go[v, :P]])
    gfull = torch.cat([gm[v,:P] * 5.0, gr[v,:P]], dim=1)
    outer = coeff.unsqueeze(2) * gfull.unsqueeze(1)
    want = torch.zeros(Tu + 1, 16, 7, device=DEV)
    want[:Tu].index_add_(0, ti, -outer)
    want[Tu] = outer.sum(0)
    print(v, "single err", [f"{e:.1e}" for e in (bv.grad - want).abs().amax(dim=(1, 2)).tolist()])
    print(v, "views  err", [f"{e:.1e}" for e in (d_bases[v] - want).abs().amax(dim=(1, 2)).tolist()], "scale", float(want.abs().max()))
