"""Every torch.empty() filled with NaN / 0xFF (torch.utils.deterministic.fill_uninitialized_memory): a kernel that reads a
workspace before writing it shows up as NaN.  usage: python scripts/dbg_fill.py [P] [graph]"""
import sys, torch
sys.path.insert(0, '.')
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
from rodygs_amd.synthetic import synthetic_scene
from rodygs_amd.trainstep import DynamicScene, GraphedStep
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W, H = 1920, 1080
sc = synthetic_scene(P, W, H, 3, seed=777)
tgt = synthetic_scene(P // 4, W, H, 3, seed=1234)
ds = DynamicScene(sc, num_frames=100, device="cuda", spatial_order=True)
perm = sorted(set(int(round(i * 100 / 16)) % 100 for i in range(16)))
ds.make_ground_truth(tgt, perm)

def bad():
    torch.cuda.synchronize()
    out = []
    for f in (ds.fp, ds.sp):
        for k in f.names:
            if not bool(torch.isfinite(f[k]).all()): out.append("param:" + k)
            if not bool(torch.isfinite(f[k].grad).all()): out.append("grad:" + k + f"({int((~torch.isfinite(f[k].grad)).sum())})")
    return out

losses = []
for s_ in range(8):
    losses.append(float(ds.train_step(s_, 0, 1, perm)))
    b = bad()
    if b:
        print("eager step", s_, "BAD", b[:10]); break
print("eager losses", [round(l, 5) for l in losses], "m2.grad col2 finite", bool(torch.isfinite(ds.m2.grad[:, 2]).all()))
if len(sys.argv) > 2 and sys.argv[2] == "eagerdev":
    # the *_dev entry points and staged inputs, launched eagerly (no graph)
    gs = GraphedStep.__new__(GraphedStep)
    gs.ds, gs.perm = ds, list(perm)
    gs.scal = torch.zeros(4, dtype=torch.float32, device="cuda")
    gs.ring = torch.zeros(gs.RING, 4, dtype=torch.float32).pin_memory(); gs.ring_i32 = gs.ring.view(torch.int32)
    gs.emb_in = torch.empty_like(ds.emb_rows[0]); gs.gt_in = torch.empty_like(ds.gt[perm[0]])
    gs._slot, gs._fence = 0, []
    ds._graph_inputs = (gs.emb_in, gs.gt_in, gs.scal)
    from rodygs_amd import rasterizer
    rasterizer.GRAPH_CAPTURE = True
    for s_ in range(8, 16):
        gs._stage(s_)
        l = float(ds.train_step(s_, 0, 1, perm))
        print("eager-dev step", s_, "loss", round(l, 5), "BAD", bad()[:8])
elif len(sys.argv) > 2:
    gs = GraphedStep(ds, perm, warmup=2, first_step=8)
    for i in range(3):
        l = float(gs.step())
        b = bad()
        print("replay", i, "loss", round(l, 5), "BAD", b[:6])
        if b:
            g = ds.fp["xyz"].grad
            nanrow = ~torch.isfinite(g).all(dim=1)
            print("  NaN xyz rows", int(nanrow.sum()), "first", nanrow.nonzero()[:5].flatten().tolist(), "last", nanrow.nonzero()[-3:].flatten().tolist())
            for k in ("scaling", "rotation", "opacity", "features", "motion_coeff"):
                gg = ds.fp[k].grad.reshape(P, -1)
                nr = ~torch.isfinite(gg).all(dim=1)
                print("  ", k, int(nr.sum()), "same rows as xyz:", bool((nr == nanrow).all()))
            print("   nren", gs._nren.tolist(), "m2 nan rows", int((~torch.isfinite(ds.m2.grad).all(dim=1)).sum()) if ds.m2.grad is not None else None)
            break
