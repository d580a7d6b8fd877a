"""PSNR delta (BASELINE.json metric: "PSNR delta vs ref", north_star: within 0.05 dB) THROUGH THE REAL TRAIN STEP.

    python scripts/psnr_delta.py [--points 20000 --width 320 --height 240 --steps 500] --out profiles/r03_psnr_delta.json
    python scripts/psnr_delta.py --teacher-forced [--bias 0.01] ...      # the systematic part alone / its power check

The reference CUDA rasterizer cannot run here (no source, no NVIDIA GPU); the oracle is the normative restatement of
it.  One dynamic scene, one initial state, one frame order, one set of ground-truth images and
split samples; the oracle trains it on the CPU (oracle/deform_oracle.py + oracle/rasterizer_oracle.py +
oracle/densify_oracle.py, the torch loss expression, torch.optim.Adam(eps=1e-15) with the reference's parameter groups,
/root/reference/src/trainer/rodygs_static.py:106-141; one densify_and_prune half-way, :280-301).  Against it:

  teacher_forced  at EVERY state of the oracle's training the HIP gradient is computed too, and the next states the two
                  gradients lead to are scored by the same renderer: drift_db = the accumulated systematic difference,
                  free of chaos, deterministic, gated at 0.05 dB with no sigma (see teacher_forced());
  det_fused       rodygs_amd.trainstep.DynamicScene.train_step as bench.py times it (MFMA MLP, fused deformation +
                  activations, HIP rasterizer, fused 0.8 L1 + 0.2 D-SSIM, fused Adam with the SH features stepped
                  inside the per-Gaussian backward kernel), free-running, in DETERMINISTIC mode (no float atomics: one
                  reproducible trajectory); det_unfused = the SH Adam in the separate launch -- the same bits;
  hip_*           free-running runs of the default float-atomic mode: their run-to-run spread is the chaotic part (Adam
                  with eps 1e-15 amplifies last-bit differences: two runs of ONE binary end 0.1-0.2 dB apart), and the
                  oracle's end value is one more draw of that process.

The densification masks (decided by the HIP densify on the oracle's state and statistics) are replayed in every run, so
that all runs train the SAME set of Gaussians; `hip_fused_free_rerun` thresholds its own statistics instead.  PSNR as the
reference's evaluator computes it (/root/reference/src/utils/eval_utils.py:36-39), mean over the training frames.
Gates of the -m gpu test: |teacher-forced drift| <= 0.05 dB; the deterministic run within 0.01 dB of the oracle at step
100, every run within 0.1 dB at the densification and the HIP runs within 0.12 dB of each other there (the oracle's CPU
trajectory is host-dependent by then); |mean(atomic runs) - oracle| <= 0.05 dB +
2 standard errors at the end, the standard error counting the oracle as one draw."""
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


def _hip_run(scene, target, frames, steps, densify_at, fused, z, dev, sh_degree=3, decisions=None, checkpoints=(),
             deterministic=False):
    from rodygs_amd import rasterizer, trainstep
    from rodygs_amd.checkpoint import psnr
    from rodygs_amd.trainstep import DynamicScene
    trainstep._FUSE_SH_ADAM = bool(fused)
    rasterizer.DETERMINISTIC = bool(deterministic)
    ds = DynamicScene(scene, num_frames=frames, sh_degree=sh_degree, device=dev, seed=777)
    perm = list(range(frames))
    ds.make_ground_truth(target, perm)
    init = {}
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
    rasterizer.DETERMINISTIC = os.environ.get("RDG_DETERMINISTIC", "0") == "1"
    return out, init


def teacher_forced(points=20000, width=320, height=240, steps=500, frames=8, densify_at=None, dev="cuda", seed=3,
                   sh_degree=3, threads=16, verbose=False, checkpoints=(), bias=0.0):
    """The SYSTEMATIC part of the PSNR delta, separated from the chaotic part.

    Two free-running trainings -- even two runs of one program with float atomics -- end 0.1-0.2 dB apart after 500
    steps: Adam with eps 1e-15 amplifies last-bit differences, so a free-running comparison measures chaos, not the
    kernels.  Here the oracle's training IS the trajectory: at every step k both sides start from the oracle's state
    S_k (parameters, MLP, poses), the HIP path (rodygs_amd.trainstep.DynamicScene.render + fused loss + backward, i.e.
    the kernels bench.py times) and the oracle each compute the gradient of the step's frame, the SAME Adam arithmetic
    (torch, from the oracle's moments) turns each gradient into a next state, and the PSNR of the two next states over
    all training frames is compared (both through the HIP renderer, whose forward parity is 1e-4-gated elsewhere):
        delta_k = PSNR(step(S_k, g_hip)) - PSNR(step(S_k, g_oracle)).
    A biased gradient (an approximation in the compositing backward, a wrong term) has delta_k of one sign and adds up;
    rounding noise does not.  `drift_db` = sum_k delta_k is what a HIP-gradient training would lose or gain against the
    oracle to first order; it is deterministic (RDG_DETERMINISTIC backward) and is gated at 0.05 dB with no sigma.
    `bias`: power check -- the HIP dL/dopacity is scaled by (1 + bias) before it is used (a 1 % error in ONE gradient).
    The oracle's optimiser never sees a HIP gradient, so its trajectory is the free-running oracle training: its PSNR
    (through the ORACLE renderer) at `checkpoints` and at the end, and the densification masks, are returned for the
    free-running comparison of run()."""
    from oracle import deform_oracle as DO
    from oracle import densify_oracle as DN
    from oracle import rasterizer_oracle as O
    from rodygs_amd import rasterizer
    from rodygs_amd.checkpoint import psnr
    from rodygs_amd.losses import fused_photometric_loss, photometric_loss
    from rodygs_amd.synthetic import synthetic_scene
    from rodygs_amd.trainstep import DynamicScene, world_view_transform
    torch.set_num_threads(max(1, min(torch.get_num_threads(), threads)))
    densify_at = steps // 2 if densify_at is None else densify_at
    scene = synthetic_scene(points, width, height, 3, seed=seed)
    target = synthetic_scene(points, width, height, 3, seed=seed + 1)
    z = torch.randn(2 * 3 * points, 3, generator=torch.Generator().manual_seed(96 + seed))
    old_det = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    ds = DynamicScene(scene, num_frames=frames, sh_degree=sh_degree, device=dev, seed=777)
    perm = list(range(frames))
    ds.make_ground_truth(target, perm)
    ds.track_densification()
    gt = {f: ds.gt[f].cpu().clone() for f in perm}
    lr, row_lr = dict(ds.fp.lr), dict(ds.row_lr)
    fp0 = {k: ds.fp[k].detach().cpu().clone() for k in ds.fp.names}
    params = {"xyz": fp0["xyz"], "f_dc": fp0["features"][:, :1], "f_rest": fp0["features"][:, 1:],
              "scaling": fp0["scaling"], "rotation": fp0["rotation"], "opacity": fp0["opacity"],
              "motion_coeff": fp0["motion_coeff"]}
    params = {k: v.clone().contiguous().requires_grad_(True) for k, v in params.items()}
    glr = {"xyz": lr["xyz"], "f_dc": lr["features"], "f_rest": row_lr["features"][2], "scaling": lr["scaling"],
           "rotation": lr["rotation"], "opacity": lr["opacity"], "motion_coeff": lr["motion_coeff"]}
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ds.net.state_dict().items()
          if v.is_floating_point()}
    cam_q = ds.cam_q.detach().cpu().clone().requires_grad_(True)
    cam_t = ds.cam_t.detach().cpu().clone().requires_grad_(True)
    small = [{"params": list(sd.values()), "lr": 0.0016}, {"params": [cam_q], "lr": 1e-5}, {"params": [cam_t], "lr": 1e-6}]
    emb_rows, proj_t = ds.emb_rows.detach().cpu().clone(), ds.proj_t.cpu().clone()
    time_ind = ds.time_ind.cpu().clone()
    sls = ds.spatial_lr_scale

    def make_opt(p, state=None):
        opt = torch.optim.Adam([{"params": [p[k]], "lr": glr[k], "name": k} for k in GROUPS] + small, lr=0.0, eps=1e-15)
        if state is not None:
            for k, st_ in state.items():
                opt.state[p[k]] = st_
        return opt

    opt = make_opt(params)
    stg = O.OracleSettings(height, width, ds.tanfovx, ds.tanfovy, torch.zeros(3), 1.0, proj_t, sh_degree)
    accum = torch.zeros(points, 1)
    denom = torch.zeros(points, 1)
    max_radii = torch.zeros(points)

    def load_into_hip(p, sd_, cq, ct):
        """parameters (not moments) -> the HIP scene's flat buckets"""
        with torch.no_grad():
            for k in ("xyz", "scaling", "rotation", "opacity", "motion_coeff"):
                ds.fp[k].copy_(p[k].detach().to(dev).reshape(ds.fp[k].shape))
            ds.fp["features"].copy_(torch.cat([p["f_dc"].detach(), p["f_rest"].detach()], dim=1).to(dev))
            ds.net.load_state_dict({k: v.detach() for k, v in sd_.items()}, strict=False)
            ds.cam_q.copy_(cq.detach().to(dev))
            ds.cam_t.copy_(ct.detach().to(dev))

    def hip_psnr():
        # eval_utils.py:36-39 (10 log10(1 / MSE) on images clipped to [0, 1]) with the MSE summed in float64: the
        # per-step differences this protocol adds up are far below float32 resolution of a 20 dB figure
        with torch.no_grad():
            vals = []
            for f in perm:
                d = ds.render(f)[0][0].clamp(0, 1).double() - ds.gt[f].clamp(0, 1).double()
                vals.append(-10.0 * torch.log10((d * d).mean()))
            return float(torch.stack(vals).mean())

    def hip_grads(frame):
        out, _ = ds.render(frame)                      # fuse_sh_adam is off outside train_step: plain gradient sinks
        fused_photometric_loss(out[0], ds.gt[frame], 0.2).backward()
        g = {k: ds.fp[k].grad.detach().cpu().reshape(params[k].shape).clone()
             for k in ("xyz", "scaling", "rotation", "opacity", "motion_coeff")}
        fg = ds.fp["features"].grad.detach().cpu()
        g["f_dc"], g["f_rest"] = fg[:, :1].clone(), fg[:, 1:].clone()
        if bias:
            g["opacity"] = g["opacity"] * (1.0 + bias)
        stacked = {n: q.grad.detach().cpu().clone() for n, q in ds.net.named_parameters()}
        for (layer, kind), name in (((0, "weight"), "head_w1"), ((0, "bias"), "head_b1"), ((2, "weight"), "head_w2"),
                                    ((2, "bias"), "head_b2")):
            t = stacked.pop(name)
            for b in range(t.shape[0]):
                stacked[f"basis_xyz.{b}.basis.{layer}.{kind}"] = t[b]
        return g, stacked, ds.sp["cam_q"].grad.detach().cpu().clone(), ds.sp["cam_t"].grad.detach().cpu().clone()

    def adam_next(p, g, st, lr_, b1=0.9, b2=0.999, eps=1e-15):
        """torch.optim.Adam's single-tensor arithmetic from the pre-step moments `st` (not modified)."""
        step = float(st["step"]) + 1 if st else 1.0
        m = (st["exp_avg"] if st else torch.zeros_like(p)).lerp(g, 1 - b1)
        v = (st["exp_avg_sq"] if st else torch.zeros_like(p)) * b2 + (1 - b2) * g * g
        den = v.sqrt() / (1 - b2 ** step) ** 0.5 + eps
        return p.detach() - (lr_ / (1 - b1 ** step)) * m / den

    def next_state(gp, gsd, gq, gt_):
        p2 = {k: adam_next(params[k], gp[k], opt.state.get(params[k]), glr[k]) for k in GROUPS}
        sd2 = {k: adam_next(v, gsd[k], opt.state.get(v), 0.0016) for k, v in sd.items()}
        return p2, sd2, adam_next(cam_q, gq, opt.state.get(cam_q), 1e-5), adam_next(cam_t, gt_, opt.state.get(cam_t), 1e-6)

    def render(frame, p, want_m2=False):
        P = p["xyz"].shape[0]
        allb = DO.motion_basis(sd, emb_rows[frame])
        dxyz, drot = DO.gaussian_deformation(p["motion_coeff"], time_ind, allb[frames], allb[:frames], sls)
        m2 = torch.zeros(P, 3, requires_grad=want_m2)
        vm = world_view_transform(cam_q[frame], cam_t[frame]).t().contiguous()
        out = O.rasterize(p["xyz"] + dxyz, m2, torch.sigmoid(p["opacity"]), vm, stg,
                          shs=torch.cat([p["f_dc"], p["f_rest"]], dim=1), scales=torch.exp(p["scaling"]),
                          rotations=torch.nn.functional.normalize(p["rotation"], dim=1) + drot)
        return out, m2

    def oracle_psnr():
        with torch.no_grad():
            return float(torch.stack([psnr(gt[f], render(f, params)[0][0]) for f in range(frames)]).mean())

    load_into_hip(params, sd, cam_q, cam_t)
    first, first_oracle = hip_psnr(), oracle_psnr()
    t0 = time.perf_counter()
    deltas, info, dec, curve = [], None, None, {}
    for step in range(steps):
        if step in checkpoints:
            curve[step] = oracle_psnr()
        if step == densify_at:
            # the HIP densification runs on the oracle's state and statistics; the oracle replays its decisions
            load_into_hip(params, sd, cam_q, cam_t)
            with torch.no_grad():
                ds.stats.xyz_gradient_accum.copy_(accum.to(dev))
                ds.stats.denom.copy_(denom.to(dev))
                ds.stats.max_radii2D.copy_(max_radii.to(dev).reshape(ds.stats.max_radii2D.shape))
            hinfo = ds.densify(z=z)
            dec = {k: v.cpu() for k, v in hinfo.pop("decisions").items()}
            st = DN.State({k: v.detach().clone() for k, v in params.items()},
                          {k: opt.state[params[k]]["exp_avg"].clone() for k in GROUPS},
                          {k: opt.state[params[k]]["exp_avg_sq"].clone() for k in GROUPS},
                          accum, denom, max_radii, {"time_ind": time_ind})
            n_step = opt.state[params["xyz"]]["step"]
            small_state = {id(q): opt.state[q] for g_ in small for q in g_["params"] if q in opt.state}
            n_clone, n_split = DN.densify_and_prune(st, 0.0002, 0.005, sls, None, 0.01, 2, z, decisions=dec)
            params = {k: v.clone().contiguous().requires_grad_(True) for k, v in st.params.items()}
            opt = make_opt(params, {k: {"step": n_step.clone(), "exp_avg": st.exp_avg[k].clone(),
                                        "exp_avg_sq": st.exp_avg_sq[k].clone()} for k in GROUPS})
            for g_ in small:
                for q in g_["params"]:
                    if id(q) in small_state:
                        opt.state[q] = small_state[id(q)]
            time_ind = st.per_point["time_ind"]
            accum, denom, max_radii = st.accum, st.denom, st.max_radii
            info = {"P": st.P, "cloned": n_clone, "split": n_split, "hip_P": hinfo["P"]}
            assert st.P == hinfo["P"] and torch.equal(ds.time_ind.cpu(), time_ind)
        frame = step % frames
        # ---- both gradients at S_k ----
        load_into_hip(params, sd, cam_q, cam_t)
        g_hip = hip_grads(frame)
        opt.zero_grad(set_to_none=True)
        out, m2 = render(frame, params, want_m2=True)
        photometric_loss(out[0], gt[frame], 0.2).backward()
        g_or = ({k: params[k].grad for k in GROUPS}, {k: v.grad for k, v in sd.items()}, cam_q.grad, cam_t.grad)
        # ---- the two next states, scored by the same renderer ----
        load_into_hip(*next_state(*g_hip))
        p_hip = hip_psnr()
        load_into_hip(*next_state(*g_or))
        p_or = hip_psnr()
        deltas.append(p_hip - p_or)
        with torch.no_grad():                                                   # add_densification_stats (oracle's)
            vis = out[4] > 0
            g = torch.norm(m2.grad[:, :2], dim=-1, keepdim=True)
            accum[vis] += g[vis]
            denom[vis] += 1
            max_radii[vis] = torch.max(max_radii[vis], out[4][vis].to(max_radii.dtype))
        opt.step()
        if verbose and (step % 50 == 0 or step == steps - 1):
            print(f"teacher-forced step {step}: psnr {p_or:.4f} dB, delta {deltas[-1]:+.2e}, drift {sum(deltas):+.5f} dB",
                  flush=True)
    load_into_hip(params, sd, cam_q, cam_t)
    last = hip_psnr()
    rasterizer.DETERMINISTIC = old_det
    d = torch.tensor(deltas, dtype=torch.float64)
    return {"psnr_start_db": first, "psnr_end_db": last, "drift_db": float(d.sum()), "abs_sum_db": float(d.abs().sum()),
            "max_abs_step_delta_db": float(d.abs().max()), "mean_step_delta_db": float(d.mean()),
            "std_step_delta_db": float(d.std()), "steps": steps, "densify": info, "bias": bias,
            "seconds": time.perf_counter() - t0,
            "config": {"points": points, "width": width, "height": height, "frames": frames, "seed": seed},
            "oracle": {"psnr_start_db": first_oracle, "psnr_end_db": oracle_psnr(), "psnr_at_step": curve,
                       "P_end": int(params["xyz"].shape[0]), "densify": info, "threads": torch.get_num_threads(),
                       "seconds": time.perf_counter() - t0},
            "_decisions": dec}


def run(points=20000, width=320, height=240, steps=500, frames=8, densify_at=None, dev="cuda", verbose=False,
        free_rerun=False, atomic_runs=4, seed=3):
    """det_fused / det_unfused: the train step in DETERMINISTIC mode (no float atomics: one reproducible trajectory;
    the two must agree bit for bit, the SH Adam being the same arithmetic in either place); hip_*: `atomic_runs` runs of
    the default float-atomic mode, whose run-to-run spread is reported next to the deterministic delta."""
    from rodygs_amd.synthetic import synthetic_scene
    densify_at = steps // 2 if densify_at is None else densify_at
    scene = synthetic_scene(points, width, height, 3, seed=seed)
    target = synthetic_scene(points, width, height, 3, seed=seed + 1)
    z = torch.randn(2 * 3 * points, 3, generator=torch.Generator().manual_seed(96 + seed))   # split samples, shared by all runs
    ck = tuple(sorted({min(100, steps // 5), densify_at}))          # early, and just before the densification
    res = {}
    # the oracle's training, with the HIP gradient evaluated at every one of its states (teacher_forced): the systematic
    # part of the delta, the oracle's own PSNR curve, and the densification masks every other run replays
    tf = teacher_forced(points, width, height, steps, frames, densify_at, dev, seed, verbose=verbose, checkpoints=ck)
    dec = tf.pop("_decisions")
    res["oracle"] = tf.pop("oracle")
    res["teacher_forced"] = tf
    if verbose:
        print("teacher_forced", tf, flush=True)
        print("oracle", res["oracle"], flush=True)
    names = (("det_fused", True, True), ("det_unfused", False, True), ("hip_fused", True, False),
             ("hip_unfused", False, False), ("hip_fused_2", True, False), ("hip_unfused_2", False, False),
             ("hip_fused_3", True, False), ("hip_unfused_3", False, False), ("hip_fused_4", True, False),
             ("hip_unfused_4", False, False))
    for name, fused, det in names[:2 + atomic_runs]:
        res[name], _ = _hip_run(scene, target, frames, steps, densify_at, fused, z, dev, decisions=dec, checkpoints=ck,
                                deterministic=det)
        if verbose:
            print(name, res[name], flush=True)
    if free_rerun:
        res["hip_fused_free_rerun"], _ = _hip_run(scene, target, frames, steps, densify_at, True, z, dev, checkpoints=ck)
        if verbose:
            print("hip_fused_free_rerun", res["hip_fused_free_rerun"], flush=True)
    hips = [k for k in res if k.startswith("hip_") and "free" not in k]
    dets = [k for k in res if k.startswith("det_")]
    ends = torch.tensor([res[k]["psnr_end_db"] for k in hips], dtype=torch.float64)
    res["delta_db"] = {k: res[k]["psnr_end_db"] - res["oracle"]["psnr_end_db"] for k in res
                       if k.startswith(("hip_", "det_"))}
    res["delta_db_at_step"] = {str(c): {k: res[k]["psnr_at_step"][c] - res["oracle"]["psnr_at_step"][c]
                                        for k in dets + hips} for c in ck}
    # Float atomics make two runs of the SAME HIP program differ in the last bits of every gradient; Adam (eps 1e-15)
    # turns that into a run-to-run spread of the final PSNR.  The oracle is one more sample of the same process: the
    # figure of merit is the distance of the oracle from the MEAN of the HIP runs, next to that spread.
    res["summary"] = {"hip_runs": hips, "oracle_end_db": res["oracle"]["psnr_end_db"],
                      "det_end_db": res["det_fused"]["psnr_end_db"],
                      "det_delta_db": res["det_fused"]["psnr_end_db"] - res["oracle"]["psnr_end_db"],
                      "det_fused_equals_unfused": res["det_fused"]["psnr_end_db"] == res["det_unfused"]["psnr_end_db"],
                      "teacher_forced_drift_db": res["teacher_forced"]["drift_db"]}
    if hips:
        n = len(hips)
        sd_ = float(ends.std()) if n > 1 else 0.0
        res["summary"].update({"hip_mean_end_db": float(ends.mean()), "hip_std_end_db": sd_, "n_atomic_runs": n,
                               "mean_delta_db": float(ends.mean()) - res["oracle"]["psnr_end_db"],
                               # the oracle is ONE draw of the same chaotic process: s.e. of (mean of n) - (one draw)
                               "mean_delta_se_db": sd_ * (1.0 + 1.0 / n) ** 0.5})
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
    ap.add_argument("--atomic-runs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--no-free-rerun", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--teacher-forced", action="store_true", help="the systematic part only (see teacher_forced)")
    ap.add_argument("--bias", type=float, default=0.0, help="teacher-forced power check: scale the HIP dL/dopacity")
    a = ap.parse_args()
    if a.teacher_forced:
        res = teacher_forced(a.points, a.width, a.height, a.steps, a.frames, seed=a.seed, verbose=True, bias=a.bias)
        res.pop("_decisions", None)
        print(json.dumps(res))
        if a.out:
            with open(os.path.join(ROOT, a.out), "w") as f:
                json.dump(res, f, indent=1)
        return
    res = run(a.points, a.width, a.height, a.steps, a.frames, verbose=True, free_rerun=not a.no_free_rerun,
              atomic_runs=a.atomic_runs, seed=a.seed)
    print(json.dumps(res))
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
