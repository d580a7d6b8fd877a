"""PSNR delta (BASELINE.json metric: "PSNR delta vs ref", north_star: within 0.05 dB) THROUGH THE REAL TRAIN STEP.

    python scripts/psnr_delta.py [--points 20000 --width 320 --height 240 --steps 500] --out profiles/r02_psnr_delta.json

The reference CUDA rasterizer cannot run here (no source, no NVIDIA GPU); the oracle is the normative restatement of
it.  The same dynamic scene is trained three times from the same initial state, same frame order, same
ground-truth images, same densification samples:

  hip_fused    rodygs_amd.trainstep.DynamicScene.train_step as bench.py times it: MFMA MLP, fused deformation +
               activations, HIP rasterizer, fused 0.8 L1 + 0.2 D-SSIM, fused Adam with the SH features stepped inside
               the per-Gaussian backward kernel;
  hip_unfused  the same with the SH Adam in the separate launch (RDG_FUSE_SH_ADAM=0);
  oracle       a CPU loop built from the oracle (oracle/deform_oracle.py + oracle/rasterizer_oracle.py +
               oracle/densify_oracle.py), the torch loss expression and torch.optim.Adam(eps=1e-15) with the reference's
               parameter groups (/root/reference/src/trainer/rodygs_static.py:106-141).

One densification (densify_and_prune, /root/reference/src/trainer/rodygs_static.py:280-301) happens half-way in all
three; the clone / split / prune masks of the first run are replayed in the other two, so that all runs train the SAME
set of Gaussians (with every run thresholding its own statistics, a handful of borderline Gaussians flip -- float
atomics make even two identical HIP runs differ in the last bits -- P differs by a few, and the trajectories separate
by ~0.1 dB: `hip_fused_free_rerun` records that spread).  PSNR as the reference's evaluator computes it (/root/reference/src/utils/eval_utils.py:36-39), mean over the
training frames.  Two identical HIP runs already differ by a few hundredths of a dB after 500 steps (float atomics change
the last bits of every gradient, Adam with eps 1e-15 amplifies them), so the HIP side is run four times and the gates
of the -m gpu test are: every run within 0.01 dB of the oracle early on (step 100, before the trajectories have had
time to separate), and |mean(HIP runs) - oracle| <= 0.05 dB + 2 sigma(HIP runs) at the end."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GROUPS = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "motion_coeff")


def _hip_run(scene, target, frames, steps, densify_at, fused, z, dev, sh_degree=3, decisions=None, checkpoints=()):
    from rodygs_amd import trainstep
    from rodygs_amd.checkpoint import psnr
    from rodygs_amd.trainstep import DynamicScene
    trainstep._FUSE_SH_ADAM = bool(fused)
    ds = DynamicScene(scene, num_frames=frames, sh_degree=sh_degree, device=dev, seed=777)
    perm = list(range(frames))
    ds.make_ground_truth(target, perm)
    init = {"fp": {k: ds.fp[k].detach().cpu().clone() for k in ds.fp.names},
            "sd": {k: v.detach().cpu().clone() for k, v in ds.net.state_dict().items()},
            "cam_q": ds.cam_q.detach().cpu().clone(), "cam_t": ds.cam_t.detach().cpu().clone(),
            "time_ind": ds.time_ind.cpu().clone(), "emb_rows": ds.emb_rows.detach().cpu().clone(),
            "gt": {f: ds.gt[f].cpu().clone() for f in perm}, "lr": dict(ds.fp.lr), "row_lr": dict(ds.row_lr),
            "spatial_lr_scale": ds.spatial_lr_scale, "proj_t": ds.proj_t.cpu().clone(),
            "tanfovx": ds.tanfovx, "tanfovy": ds.tanfovy}
    ds.track_densification()

    def mean_psnr():
        with torch.no_grad():
            return float(torch.stack([psnr(ds.gt[f], ds.render(f)[0][0]) for f in perm]).mean())

    first = mean_psnr()
    t0 = time.perf_counter()
    info, curve = None, {}
    for step in range(steps):
        if step in checkpoints:
            curve[step] = mean_psnr()
        if step == densify_at:
            info = ds.densify(z=z, decisions=decisions)
            init["decisions"] = {k: v.cpu() for k, v in info.pop("decisions").items()}
        ds.train_step(step, 0, 1, perm)
    torch.cuda.synchronize()
    out = {"psnr_start_db": first, "psnr_end_db": mean_psnr(), "psnr_at_step": curve,
           "seconds": time.perf_counter() - t0, "P_end": ds.P, "densify": info}
    trainstep._FUSE_SH_ADAM = os.environ.get("RDG_FUSE_SH_ADAM", "1") != "0"
    return out, init


def _oracle_run(init, frames, steps, densify_at, z, W, H, sh_degree=3, threads=16, decisions=None, checkpoints=()):
    from oracle import deform_oracle as DO
    from oracle import densify_oracle as DN
    from oracle import rasterizer_oracle as O
    from rodygs_amd.checkpoint import psnr
    from rodygs_amd.losses import photometric_loss            # the torch expression (host mirror pinned by golden G6)
    from rodygs_amd.trainstep import world_view_transform
    torch.set_num_threads(max(1, min(torch.get_num_threads(), threads)))
    fp0, lr = init["fp"], init["lr"]
    K = fp0["features"].shape[1]
    params = {"xyz": fp0["xyz"], "f_dc": fp0["features"][:, :1], "f_rest": fp0["features"][:, 1:],
              "scaling": fp0["scaling"], "rotation": fp0["rotation"], "opacity": fp0["opacity"],
              "motion_coeff": fp0["motion_coeff"]}
    params = {k: v.clone().contiguous().requires_grad_(True) for k, v in params.items()}
    glr = {"xyz": lr["xyz"], "f_dc": lr["features"], "f_rest": init["row_lr"]["features"][2], "scaling": lr["scaling"],
           "rotation": lr["rotation"], "opacity": lr["opacity"], "motion_coeff": lr["motion_coeff"]}
    sd = {k: v.clone().requires_grad_(True) for k, v in init["sd"].items() if v.is_floating_point()}
    cam_q = init["cam_q"].clone().requires_grad_(True)
    cam_t = init["cam_t"].clone().requires_grad_(True)
    small = [{"params": list(sd.values()), "lr": 0.0016}, {"params": [cam_q], "lr": 1e-5}, {"params": [cam_t], "lr": 1e-6}]

    def make_opt(p, state=None):
        opt = torch.optim.Adam([{"params": [p[k]], "lr": glr[k], "name": k} for k in GROUPS] + small, lr=0.0, eps=1e-15)
        if state is not None:
            for k, st_ in state.items():
                opt.state[p[k]] = st_
        return opt

    opt = make_opt(params)
    time_ind = init["time_ind"].clone()
    T = frames
    stg = O.OracleSettings(H, W, init["tanfovx"], init["tanfovy"], torch.zeros(3), 1.0, init["proj_t"], sh_degree)
    accum = torch.zeros(params["xyz"].shape[0], 1)
    denom = torch.zeros(params["xyz"].shape[0], 1)
    max_radii = torch.zeros(params["xyz"].shape[0])

    def render(frame, p, want_m2=False):
        P = p["xyz"].shape[0]
        allb = DO.motion_basis(sd, init["emb_rows"][frame])                     # [T+1,16,7]
        dxyz, drot = DO.gaussian_deformation(p["motion_coeff"], time_ind, allb[T], allb[:T], init["spatial_lr_scale"])
        m2 = torch.zeros(P, 3, requires_grad=want_m2)
        vm = world_view_transform(cam_q[frame], cam_t[frame]).t().contiguous()
        out = O.rasterize(p["xyz"] + dxyz, m2, torch.sigmoid(p["opacity"]), vm, stg,
                          shs=torch.cat([p["f_dc"], p["f_rest"]], dim=1), scales=torch.exp(p["scaling"]),
                          rotations=torch.nn.functional.normalize(p["rotation"], dim=1) + drot)
        return out, m2

    def mean_psnr():
        with torch.no_grad():
            return float(torch.stack([psnr(init["gt"][f], render(f, params)[0][0]) for f in range(frames)]).mean())

    first = mean_psnr()
    t0 = time.perf_counter()
    info, curve = None, {}
    for step in range(steps):
        if step in checkpoints:
            curve[step] = mean_psnr()
        if step == densify_at:
            st = DN.State({k: v.detach().clone() for k, v in params.items()},
                          {k: opt.state[params[k]]["exp_avg"].clone() for k in GROUPS},
                          {k: opt.state[params[k]]["exp_avg_sq"].clone() for k in GROUPS},
                          accum, denom, max_radii, {"time_ind": time_ind})
            n_step = opt.state[params["xyz"]]["step"]
            small_state = {id(q): opt.state[q] for g_ in small for q in g_["params"] if q in opt.state}
            n_clone, n_split = DN.densify_and_prune(st, 0.0002, 0.005, init["spatial_lr_scale"], None, 0.01, 2, z,
                                                    decisions=decisions)
            params = {k: v.clone().contiguous().requires_grad_(True) for k, v in st.params.items()}
            opt = make_opt(params, {k: {"step": n_step.clone(), "exp_avg": st.exp_avg[k].clone(),
                                        "exp_avg_sq": st.exp_avg_sq[k].clone()} for k in GROUPS})
            for g_ in small:
                for q in g_["params"]:
                    if id(q) in small_state:
                        opt.state[q] = small_state[id(q)]
            time_ind = st.per_point["time_ind"]
            accum, denom, max_radii = st.accum, st.denom, st.max_radii
            info = {"P": st.P, "cloned": n_clone, "split": n_split}
        frame = step % frames
        opt.zero_grad(set_to_none=True)
        out, m2 = render(frame, params, want_m2=True)
        photometric_loss(out[0], init["gt"][frame], 0.2).backward()
        with torch.no_grad():                                                   # add_densification_stats
            vis = out[4] > 0
            g = torch.norm(m2.grad[:, :2], dim=-1, keepdim=True)
            accum[vis] += g[vis]
            denom[vis] += 1
            max_radii[vis] = torch.max(max_radii[vis], out[4][vis].to(max_radii.dtype))
        opt.step()
    return {"psnr_start_db": first, "psnr_end_db": mean_psnr(), "psnr_at_step": curve,
            "seconds": time.perf_counter() - t0, "P_end": int(params["xyz"].shape[0]), "densify": info,
            "threads": torch.get_num_threads()}


def run(points=20000, width=320, height=240, steps=500, frames=8, densify_at=None, dev="cuda", verbose=False,
        free_rerun=False):
    from rodygs_amd.synthetic import synthetic_scene
    densify_at = steps // 2 if densify_at is None else densify_at
    scene = synthetic_scene(points, width, height, 3, seed=3)
    target = synthetic_scene(points, width, height, 3, seed=4)
    z = torch.randn(2 * 3 * points, 3, generator=torch.Generator().manual_seed(99))   # split samples, shared by all runs
    ck = tuple(sorted({min(100, steps // 5), densify_at}))          # early, and just before the densification
    res = {}
    res["hip_fused"], init = _hip_run(scene, target, frames, steps, densify_at, True, z, dev, checkpoints=ck)
    dec = init["decisions"]
    if verbose:
        print("hip_fused", res["hip_fused"], flush=True)
    # the clone / split / prune masks of the first run are REPLAYED in the others: thresholding each run's own
    # statistics lets a borderline Gaussian flip, which changes P and from there the whole trajectory
    for name, fused in (("hip_unfused", False), ("hip_fused_2", True), ("hip_unfused_2", False)):
        res[name], _ = _hip_run(scene, target, frames, steps, densify_at, fused, z, dev, decisions=dec, checkpoints=ck)
        if verbose:
            print(name, res[name], flush=True)
    if free_rerun:
        res["hip_fused_free_rerun"], _ = _hip_run(scene, target, frames, steps, densify_at, True, z, dev, checkpoints=ck)
        if verbose:
            print("hip_fused_free_rerun", res["hip_fused_free_rerun"], flush=True)
    res["oracle"] = _oracle_run(init, frames, steps, densify_at, z, width, height, decisions=dec, checkpoints=ck)
    if verbose:
        print("oracle", res["oracle"], flush=True)
    hips = [k for k in res if k.startswith("hip_") and "free" not in k]
    ends = torch.tensor([res[k]["psnr_end_db"] for k in hips], dtype=torch.float64)
    res["delta_db"] = {k: res[k]["psnr_end_db"] - res["oracle"]["psnr_end_db"] for k in res if k != "oracle"}
    res["delta_db_at_step"] = {str(c): {k: res[k]["psnr_at_step"][c] - res["oracle"]["psnr_at_step"][c] for k in hips}
                               for c in ck}
    # Float atomics make two runs of the SAME HIP program differ in the last bits of every gradient; Adam (eps 1e-15)
    # turns that into a run-to-run spread of the final PSNR.  The oracle is one more sample of the same process: the
    # figure of merit is the distance of the oracle from the MEAN of the HIP runs, next to that spread.
    res["summary"] = {"hip_runs": hips, "hip_mean_end_db": float(ends.mean()), "hip_std_end_db": float(ends.std()),
                      "oracle_end_db": res["oracle"]["psnr_end_db"],
                      "mean_delta_db": float(ends.mean()) - res["oracle"]["psnr_end_db"]}
    res["config"] = {"points": points, "width": width, "height": height, "steps": steps, "frames": frames,
                     "densify_at": densify_at, "sh_degree": 3, "loss": "0.8 L1 + 0.2 D-SSIM",
                     "optimizer": "Adam eps 1e-15, reference parameter groups (xyz, f_dc, f_rest, scaling, rotation, "
                                  "opacity, motion coefficients, deformation MLP, camera poses)",
                     "psnr": "10 log10(1 / MSE) on images clipped to [0,1] (eval_utils.py:36-39), mean over frames",
                     "gate_db": 0.05}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = run(a.points, a.width, a.height, a.steps, a.frames, verbose=True, free_rerun=True)
    print(json.dumps(res))
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
