"""CPU multi-process tests (gloo, world_size 2) of the frame-data-parallel path: the flat-bucket gradient
all-reduce used by bench.py / trainstep.py, with per-rank gradients produced by the oracle rasterizer for two
different cameras -- the all-reduced bucket must equal the sum of the single-rank gradients (SURVEY.md §4 tier 4)."""
import os
import socket
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_grads(rank):
    from oracle import rasterizer_oracle as O
    sc = O.synthetic_scene(300, 64, 48, 1, seed=21)
    view = sc["viewmatrix"].clone()
    view[3, 0] = 0.2 * (rank + 1)  # glm storage: row 3 is the translation; every rank = another camera/frame
    st = O.OracleSettings(48, 64, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 1)
    ins = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    out = O.rasterize(ins["means3D"], torch.zeros(300, 3), ins["opacities"], view, st, shs=ins["shs"],
                      scales=ins["scales"], rotations=ins["rotations"])
    out[0].sum().add(out[1].sum()).backward()
    return {k: v.grad for k, v in ins.items()}


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rodygs_amd.dp import FlatParams, allreduce_sum_, frame_for
    spec = {"means3D": ((300, 3), 1e-3), "shs": ((300, 4, 3), 1e-3), "opacities": ((300, 1), 1e-2),
            "scales": ((300, 3), 1e-3), "rotations": ((300, 4), 1e-3)}
    fp = FlatParams(spec, "cpu")
    grads = _rank_grads(rank)
    for k, g in grads.items():
        fp[k].grad.copy_(g)
    small = [torch.full((5,), float(rank + 1)), torch.full((2, 3), 10.0 * (rank + 1))]
    # the overlapped exchange on a second copy: "shs" goes out early (as from inside backward), the rest afterwards
    from rodygs_amd.dp import BucketedAllReduce
    fp2 = FlatParams(spec, "cpu")
    fp2.flat_grad.copy_(fp.flat_grad)
    side = torch.full((11,), float(rank + 1))
    sync = BucketedAllReduce(fp2, [side])
    sync.ready("shs")
    sync.ready("shs")                       # idempotent
    sync.finish()
    pieces = list(sync.drain())
    from rodygs_amd.densify import DensifyStats, allreduce_stats_
    stats = DensifyStats(torch.full((4, 1), float(rank + 1)), torch.full((4, 1), 1.0), torch.tensor([1.0, 5.0, 2.0, 0.0]) * (rank + 1))
    allreduce_stats_(stats)
    allreduce_sum_(fp.flat_grad, small)
    torch.save({"stats": (stats.xyz_gradient_accum, stats.denom, stats.max_radii2D), "flat": fp.flat_grad.clone(), "small": small, "offsets": fp.offsets, "flat2": fp2.flat_grad.clone(),
                "pieces": pieces, "side": side,
                "frames": [frame_for(s, rank, world, list(range(7))) for s in range(7)]},
               os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_sum_of_single_rank_grads():
    world = 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        r0 = torch.load(os.path.join(d, "r0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(d, "r1.pt"), weights_only=False)
    assert torch.equal(r0["flat"], r1["flat"])                       # every rank holds the same summed bucket
    g0, g1 = _rank_grads(0), _rank_grads(1)
    for k, (o, n) in r0["offsets"].items():
        want = (g0[k] + g1[k]).reshape(-1)
        got = r0["flat"][o:o + n]
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-7), k
    # bucketed / overlapped exchange: same sums, pieces = early segment, then the maximal contiguous remainders, then
    # the side bucket
    assert torch.equal(r0["flat2"], r0["flat"]) and torch.equal(r1["flat2"], r0["flat"])
    assert r0["pieces"] == [["shs"], ["means3D"], ["opacities", "scales", "rotations"], None]
    assert torch.equal(r0["side"], torch.full((11,), 3.0))
    acc, den, rad = r0["stats"]
    assert torch.equal(acc, torch.full((4, 1), 3.0)) and torch.equal(den, torch.full((4, 1), 2.0))
    assert torch.equal(rad, torch.tensor([2.0, 10.0, 4.0, 0.0]))
    assert float(g0["means3D"].sub(g1["means3D"]).abs().max()) > 0  # the two frames really differ
    assert torch.equal(r0["small"][0], torch.full((5,), 3.0)) and torch.equal(r1["small"][1], torch.full((2, 3), 30.0))
    # strided frame assignment: the two ranks never render the same frame in a step and cover the permutation
    assert all(a != b for a, b in zip(r0["frames"], r1["frames"]))
    assert sorted(set(r0["frames"] + r1["frames"])) == list(range(7))


def test_flat_params_views_and_single_process_noop():
    from rodygs_amd.dp import FlatParams, allreduce_sum_
    fp = FlatParams({"a": ((5, 3), 0.1), "b": ((7,), 0.2)}, "cpu")
    assert fp["a"].is_leaf and fp["a"].requires_grad and fp["a"].grad.data_ptr() == fp.flat_grad.data_ptr()
    (fp["a"].sum() * 2 + fp["b"].sum() * 3).backward()
    o, n = fp.offsets["b"]
    assert torch.equal(fp.flat_grad[o:o + n], torch.full((7,), 3.0)) and float(fp.flat_grad[:15].sum()) == 30.0
    before = fp.flat_grad.clone()
    allreduce_sum_(fp.flat_grad)            # not initialised -> no-op
    assert torch.equal(before, fp.flat_grad)
    fp.zero_grad()
    assert float(fp["a"].grad.abs().sum()) == 0.0


# ---- Gaussian-sharded frame-DP (rodygs_amd/sharded.py): the exchange pattern with the oracle doing the arithmetic --------
_SH_P, _SH_W, _SH_H = 301, 64, 48
_FCOLS = ("px", "py", "conic", "opacity", "depth", "rgb", "normal")          # float part of a record (13 columns)
_ICOLS = ("radii", "tiles_touched", "rminx", "rminy", "rmaxx", "rmaxy")       # integer part (exact as floats)


def _sh_scene():
    from oracle import rasterizer_oracle as O
    sc = O.synthetic_scene(_SH_P, _SH_W, _SH_H, 1, seed=33)
    st = O.OracleSettings(_SH_H, _SH_W, sc["tanfovx"], sc["tanfovy"], torch.zeros(3), 1.0, sc["projmatrix"], 1)
    views = []
    for c in range(2):
        v = sc["viewmatrix"].clone()
        v[3, 0] = 0.15 * (c + 1)
        views.append(v)
    return O, sc, st, views


def _sh_records(O, sc, st, view, lo, hi, leaves=None):
    """Oracle per-Gaussian stage on Gaussians [lo, hi) -> (record rows [n,19] attached to autograd, leaves)."""
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    leaves = leaves or {k: sc[k][lo:hi].clone().requires_grad_(True) for k in names}
    g = O.preprocess(leaves["means3D"], torch.zeros(hi - lo, 3), leaves["opacities"], view, st, shs=leaves["shs"],
                     scales=leaves["scales"], rotations=leaves["rotations"])
    f = [g[k].reshape(hi - lo, -1) for k in _FCOLS]
    rect = dict(zip(("rminx", "rminy", "rmaxx", "rmaxy"), g["rect"]))
    i = [(g[k] if k in g else rect[k]).reshape(hi - lo, 1).to(torch.float32) for k in _ICOLS]
    return torch.cat(f + i, dim=1), leaves, g["grid"]


def _sh_geom(rows, grid):
    """Camera side: the oracle's geom dict rebuilt from gathered record rows (a leaf requiring grad)."""
    cols, o = {}, 0
    for k, w in zip(_FCOLS, (1, 1, 3, 1, 1, 3, 3)):
        cols[k] = rows[:, o:o + w] if w > 1 else rows[:, o]
        o += w
    ints = {k: rows[:, o + j].detach().to(torch.int32) for j, k in enumerate(_ICOLS)}
    geom = dict(cols)
    geom.update(radii=ints["radii"], tiles_touched=ints["tiles_touched"], grid=grid,
                rect=tuple(ints[k].to(torch.int64) for k in ("rminx", "rminy", "rmaxx", "rmaxy")))
    return geom


def _sh_loss(O, geom, st):
    img = O.render_tiles(geom, O.bin_and_sort(geom), st.bg, _SH_H, _SH_W)
    return img["color"].sum() + 0.5 * img["depth"].sum()


def _shard_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rodygs_amd.sharded import DistExchange, shard_rows
    O, sc, st, views = _sh_scene()
    ex = DistExchange()
    per, stride = shard_rows(_SH_P, world)
    lo, hi = rank * per, min((rank + 1) * per, _SH_P)
    n, ncol = hi - lo, 13 + len(_ICOLS)
    vleaf = [v.clone().requires_grad_(True) for v in views]
    # owner forward: my slice for BOTH cameras -> one contiguous send buffer, camera c in rows [c*stride, c*stride+n)
    send = torch.zeros(world, stride, ncol)
    recs, leaves, grid = [], None, None
    for c in range(world):
        r, leaves, grid = _sh_records(O, sc, st, vleaf[c], lo, hi, leaves)
        recs.append(r)
        send[c, :n] = r.detach()
    recv = torch.empty_like(send)
    ex.all_to_all(recv.view(-1), send.view(-1))
    # camera stage: every Gaussian of the cloud (shard s in rows [s*stride, ...)), my camera
    rows = recv.view(world * stride, ncol).clone().requires_grad_(True)
    loss = _sh_loss(O, _sh_geom(rows, grid), st)
    loss.backward()
    back = torch.empty_like(send)
    ex.all_to_all(back.view(-1), rows.grad.view(-1).contiguous())
    # owner backward: gradient rows of my slice from both cameras
    torch.autograd.backward(recs, [back[c, :n] for c in range(world)])
    dview = torch.stack([v.grad if v.grad is not None else torch.zeros(4, 4) for v in vleaf])
    ex.all_reduce(dview)
    torch.save({"loss": loss.detach(), "grads": {k: v.grad for k, v in leaves.items()}, "dview": dview, "lo": lo, "hi": hi,
                "stride": stride}, os.path.join(outdir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_exchange_reproduces_replicated_gradients():
    """Two gloo ranks, each owning half of the Gaussians and rendering one camera, with the oracle as the arithmetic:
    records out (all_to_all_single), gradient rows back, per-Gaussian backward at the owner.  The parameter gradients
    must equal the SUM over both cameras of the unsharded oracle's gradients, the pose gradients (all-reduced partial
    sums) the unsharded ones, and each rank's loss its camera's unsharded loss."""
    world = 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_shard_worker, args=(world, port, d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f"s{r}.pt"), weights_only=False) for r in range(world)]
    O, sc, st, views = _sh_scene()
    want, want_view, want_loss = None, [], []
    for c in range(world):
        v = views[c].clone().requires_grad_(True)
        rows, leaves, grid = _sh_records(O, sc, st, v, 0, _SH_P)
        full = rows.detach().clone().requires_grad_(True)
        loss = _sh_loss(O, _sh_geom(full, grid), st)
        loss.backward()
        rows.backward(full.grad)
        want_loss.append(loss.detach())
        want_view.append(v.grad)
        g = {k: t.grad for k, t in leaves.items()}
        want = g if want is None else {k: want[k] + g[k] for k in g}
    assert res[0]["stride"] == 256 and (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 151, 151, 301)
    for r in range(world):
        assert torch.allclose(res[r]["loss"], want_loss[r], rtol=1e-6, atol=0)
        for k, gr in res[r]["grads"].items():
            ref = want[k][res[r]["lo"]:res[r]["hi"]]
            assert float(ref.abs().max()) > 0
            assert torch.allclose(gr, ref, rtol=1e-4, atol=1e-6 * float(ref.abs().max())), (r, k)
        assert torch.allclose(res[r]["dview"], torch.stack(want_view), rtol=1e-4, atol=1e-5)


def _rows_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rodygs_amd.sharded import DistExchange
    ex = DistExchange()
    stride, cols = 12, 16
    send = (torch.arange(world * stride * cols, dtype=torch.float32) + 1000.0 * rank).view(world, stride, cols)
    whole = torch.empty_like(send)
    ex.all_to_all(whole.view(-1), send.view(-1).contiguous())
    pieces = torch.zeros_like(send)
    for r0, r1 in ((0, 5), (5, 6), (6, 12)):
        ex.all_to_all_rows(pieces.view(-1), send.view(-1), stride, r0, r1)
    torch.save({"whole": whole, "pieces": pieces}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_row_range_exchange_equals_the_whole_all_to_all():
    """all-to-all #1 of the sharded step cut into row ranges (DistExchange.all_to_all_rows, grouped point-to-point) fills
    the receive buffer exactly as the one equal-split all_to_all_single does (world 2 and 3, gloo)."""
    for world in (2, 3):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_rows_worker, args=(world, _free_port(), d), nprocs=world, join=True)
            for r in range(world):
                g = torch.load(os.path.join(d, f"r{r}.pt"))
                assert torch.equal(g["whole"], g["pieces"])


def test_chunk_ranges_cover_the_shard_in_aligned_pieces():
    """The row ranges of the pipelined all-to-all #1: multiples of 256, disjoint, in order, covering [0, stride), at most
    the requested number of pieces -- the same on every rank because they depend on the stride alone."""
    from rodygs_amd.sharded import chunk_ranges, shard_rows
    for P, world in ((1_000_000, 8), (6001, 2), (5003, 3), (37, 8), (4_000_000, 8)):
        _, stride = shard_rows(P, world)
        for chunks in (1, 2, 3, 4, 7, 64, 10_000):
            r = chunk_ranges(stride, chunks)
            assert r[0][0] == 0 and r[-1][1] == stride and len(r) <= max(1, min(chunks, stride // 256))
            assert all(a % 256 == 0 and b % 256 == 0 and a < b for a, b in r)
            assert all(r[i][1] == r[i + 1][0] for i in range(len(r) - 1))
    # an empty cloud: one empty range on every rank (the collective sequence must not depend on the data)
    for chunks in (1, 3, 64):
        assert chunk_ranges(shard_rows(0, 8)[1], chunks) == [(0, 0)]
