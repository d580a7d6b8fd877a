"""Times rdg_rigidity_dp_rows (both launches) and rdg_rigidity_pack_rows alone through the C-ABI: n sampled Gaussians, nt = 25."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rodygs_amd import knn as KN, _lib
from rodygs_amd.rigidity import _curve_order, _sort_by_key
n, nt, K = int(sys.argv[1]), 25, 8
L = _lib.lib()
g = torch.Generator().manual_seed(1)
canon = (torch.rand(n, 3, generator=g) * torch.tensor([8.0, 5.0, 12.0])).cuda()
own = (0.05 * torch.randn(n, nt, 3, generator=g)).cuda()
res = KN.knn_points(canon[None], canon[None], K=K)
ii, dd = res.idx[0].contiguous(), res.dists[0].contiguous()
order = _curve_order(own[:, 0] + canon)
rank = torch.empty_like(order); rank[order] = torch.arange(n, device="cuda")
ii = rank[ii[order]].contiguous()
P3 = torch.empty(n, nt, 3, device="cuda")
srt, rev_edge = _sort_by_key(ii.reshape(-1), max(1, (n - 1).bit_length()))
rev_off = torch.searchsorted(srt, torch.arange(n + 1, device="cuda")).contiguous()
loss = torch.empty(1, dtype=torch.float64, device="cuda")
IG = torch.empty(n * K * nt, device="cuda"); G_own = torch.empty(n, nt, 3, device="cuda")
G_canon = torch.empty(n, 3, device="cuda"); d_d2 = torch.empty_like(dd)
import os
RA = torch.empty(n * nt * 2, device="cuda")
RAp = _lib.ptr(RA) if os.environ.get("RDG_RIG_RA", "1") != "0" else None
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tp = td = 0.0
for it in range(6):
    ev[0].record()
    _lib.check(L.rdg_rigidity_pack_rows(n, nt, _lib.ptr(own), _lib.ptr(canon), _lib.ptr(order), _lib.ptr(P3), _lib.stream_ptr()), "pack")
    ev[1].record()
    _lib.check(L.rdg_rigidity_dp_rows(n, K, nt, _lib.ptr(P3), _lib.ptr(ii), _lib.ptr(dd), _lib.ptr(rev_off), _lib.ptr(rev_edge),
                                      _lib.ptr(order), _lib.ptr(rank), 1e-6, _lib.ptr(loss), _lib.ptr(IG), RAp, _lib.ptr(d_d2), _lib.ptr(G_own),
                                      _lib.ptr(G_canon), _lib.stream_ptr()), "dp_rows")
    ev[2].record(); torch.cuda.synchronize()
    if it: tp += ev[0].elapsed_time(ev[1]) / 5; td += ev[1].elapsed_time(ev[2]) / 5
print(f"n={n}: pack_rows {tp:.3f} ms, dp_rows (out + in) {td:.3f} ms, loss {float(loss):.9e} |d_d2| {float(d_d2.double().abs().sum()):.9e} sum {float(d_d2.double().sum()):.9e}")
