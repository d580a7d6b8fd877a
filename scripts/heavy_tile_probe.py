"""Binning time of a frame with ONE tile holding `--heavy` instances (default 200 k), on the GPU box.

    python scripts/heavy_tile_probe.py [--heavy 200000] > gpurun_out/heavy_tile.json

Real RoDyGS scenes densify every 100 iterations (/root/reference/configs/train/train_kubric_mrig.yaml:168-173) and
concentrate Gaussians; the per-tile sort must not fall off a cliff there.  Prints one JSON object: the stage times of
the binning kernels (hipEvents around the launches, mean of --reps forwards), the tile-occupancy profile, and whether
the sorted (key, value) stream equals numpy's stable sort of the emitted stream (bit-exact)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heavy", type=int, default=200000)
    ap.add_argument("--background", type=int, default=300000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import hip_stages as HS
    from rodygs_amd import GaussianRasterizer, _lib
    from rodygs_amd.synthetic import skewed_scene
    W, H = 1920, 1080
    sc = skewed_scene(W, H, [(60, 34, args.heavy), (20, 10, 30000), (90, 50, 7000), (100, 20, 3000)],
                      background=args.background, sh_degree_max=3, seed=5, equal_depth_every=9)
    P = sc["means3D"].shape[0]
    hs = HS.run_stages(sc, 3)
    order = np.argsort(hs["keys_unsorted"], kind="stable")
    exact = bool(np.array_equal(hs["keys_sorted"], hs["keys_unsorted"][order]) and
                 np.array_equal(hs["vals_sorted"], hs["vals_unsorted"][order]))
    r = hs["ranges"].astype(np.int64)
    n = np.sort(r[:, 1] - r[:, 0])[::-1]
    dev = "cuda"
    ins = {k: sc[k].to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix")}
    rs = HS.make_settings(sc, 3)
    m2 = torch.zeros(P, 3, device=dev)

    def fwd():
        with torch.no_grad():
            return GaussianRasterizer(rs)(means3D=ins["means3D"], means2D=m2, shs=ins["shs"], opacities=ins["opacities"],
                                          scales=ins["scales"], rotations=ins["rotations"], viewmatrix=ins["viewmatrix"])
    from rodygs_amd import rasterizer

    def timed(radix_above):
        rasterizer.BIN_RADIX_ABOVE = radix_above
        rasterizer._BIN_HINT.clear()
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        _lib.timing_enable(True)
        _lib.timing_reset()
        for _ in range(args.reps):
            fwd()
        torch.cuda.synchronize()
        t = {k: (ms / c if c else 0.0) for k, (ms, c) in _lib.stage_times().items()}
        _lib.timing_enable(False)
        return t, bool(rasterizer._BIN_HINT)

    def timed_fwd_bwd(split: bool):
        """forward + backward (photometric-style upstream gradient) with the long lists composited by one workgroup each
        (split off) or by the split path (per-segment partial composites, several workgroups per list)"""
        keep = rasterizer.SPLIT_ABOVE
        rasterizer.SPLIT_ABOVE = keep if split else 10 ** 12
        rasterizer._SPLIT_HINT.clear()
        leaves = {k: ins[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        w = torch.rand(3, H, W, device=dev)

        def step():
            out = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                                         shs=leaves["shs"], opacities=leaves["opacities"], scales=leaves["scales"],
                                         rotations=leaves["rotations"], viewmatrix=ins["viewmatrix"])
            (out[0] * w).sum().backward()
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            _lib.timing_enable(True)
            _lib.timing_reset()
            for _ in range(args.reps):
                step()
            torch.cuda.synchronize()
            t = {k: (ms / c if c else 0.0) for k, (ms, c) in _lib.stage_times().items()}
            _lib.timing_enable(False)
        finally:
            rasterizer.SPLIT_ABOVE = keep
            rasterizer._SPLIT_HINT.clear()
        return {k: t[k] for k in ("render_fwd", "render_bwd")}

    default_above = rasterizer.BIN_RADIX_ABOVE
    st_b, _ = timed(1 << 40)              # hint disabled: bucket binning + merge tree on every frame
    st, radix_hint = timed(default_above)  # as shipped: the first frame's largest tile switches the next ones to radix
    comp_split, comp_one = timed_fwd_bwd(True), timed_fwd_bwd(False)
    print(json.dumps({"workload": f"{P} Gaussians, {W}x{H}: one tile with {int(n[0])} instances, next {n[1:6].tolist()}",
                      "num_rendered_D": int(hs["D"]), "largest_tiles": n[:8].tolist(), "sorted_stream_bit_exact": exact,
                      "binning_ms": st["scan_dup"] + st["sort"] + st["ranges"],
                      "radix_by_hint": radix_hint,
                      "stage_ms": {k: st[k] for k in ("preprocess", "scan_dup", "sort", "ranges", "render_fwd")},
                      "bucket_only_binning_ms": st_b["scan_dup"] + st_b["sort"] + st_b["ranges"],
                      "bucket_only_stage_ms": {k: st_b[k] for k in ("scan_dup", "sort")},
                      "compositing_ms_split_path": comp_split, "compositing_ms_one_workgroup_per_tile": comp_one,
                      "reps": args.reps, "bin_mode": os.environ.get("RDG_BIN_MODE", "bucket")}))


if __name__ == "__main__":
    main()
