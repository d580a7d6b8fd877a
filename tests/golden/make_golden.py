"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own Python (only possible in the
build container where /root/reference is mounted; the GPU box and CI read the committed .npz files).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

What is pinned: the pieces of the hot path that exist as importable reference code --
  G1 deformation : MLPBasisNetwork / TimestepEmbedder (src/model/rodygs_dynamic.py:190-327) + the inverse-motion
                   arithmetic of DynRoDyGS.get_gaussian_deformation (:122-138), forward and autograd grads
  G2 camera      : FixedCameraTorch.world_view_transform, getProjectionMatrix (src/data/utils.py:105-170,
                   src/utils/graphic_utils.py:43-63)
  G3 SH          : eval_sh degrees 0..3 (src/utils/sh_utils.py:44-118)
  G4 covariance  : build_covariance_from_scaling_rotation (src/model/rodygs_static.py:26-30 +
                   src/utils/general_utils.py:77-127; its hard-coded device="cuda" is neutralised by a
                   torch.zeros wrapper for the duration of the call)
  G6 losses      : l1_loss, ssim (src/utils/loss_utils.py) used by the bench train step
  G7 rigidity    : RigidityLoss (src/trainer/losses.py:185-360), all three modes, value and gradients, with
                   pytorch3d's knn_points / knn_gather supplied by oracle/knn_oracle.py
                   (``python -B tests/golden/make_golden.py rigidity`` regenerates only this file)
  G8 depth loss  : pearson_depth_loss (src/utils/loss_utils.py:100-117), GlobalPearsonDepthLoss and
                   LocalPearsonDepthLoss (src/trainer/losses.py:108-182; its device="cuda" randint is redirected to
                   the CPU generator for the call and the drawn corners are stored), value and d/dpred
                   (``... make_golden.py depth``)
  G9 motion reg  : MotionL1Loss, MotionSparsityLoss, MotionBasisRegularizaiton (src/trainer/losses.py:363-525; the
                   constructor's ``.cuda()`` is neutralised for the call), value and gradients (``... motion``)
  G10 eval pose : matrix_to_quaternion (src/utils/graphic_utils.py:116-159), search_nearest_two
                   (src/evaluator/utils.py:15-26), l2_loss (src/utils/loss_utils.py:23-24) and the world-view matrix a
                   LearnableCamera builds from its (quaternion, translation) parameters (src/data/utils.py:173-232)
                   -- the importable pieces of the evaluator's test-time pose optimisation (``... make_golden.py pose``)
  G11 optimizer : get_expon_lr_func (src/utils/general_utils.py:40-73) and reset_opacity + replace_tensor_to_optimizer
                   (src/trainer/rodygs_static.py:151-160, src/trainer/utils.py:15-32) on a real torch.optim.Adam state,
                   plus the Adam step that follows (``... make_golden.py optimizer``)
  G12 checkpoint: a checkpoint dictionary written by rodygs_amd.checkpoint.export_state_dict goes through the file format
                   and is read back BY THE REFERENCE: DynRoDyGS.create_from_state_dict (src/model/rodygs_dynamic.py:
                   106-120, rodygs_static.py:172-182; ``.cuda()`` neutralised), sync_gaussian_to_time_ind,
                   get_total_motion_table, get_gaussian_deformation and the activated getters; its optimizer state is
                   loaded into the optimizer the reference's own ThreeDGSTrainer.optim_setup + DynTrainer.
                   append_motion_optim build (src/trainer/rodygs_static.py:106-141, rodygs_dynamic.py:93-116), one Adam
                   step is taken, and DynTrainer.state_dict (rodygs_dynamic.py:217-222) writes the checkpoint the
                   import direction is tested on (``... make_golden.py checkpoint``)
  G13 densify   : DynTrainer.densify_and_prune -- densify_and_clone / densify_and_split / prune_points /
                   densification_postfix and the optimizer surgery under them (src/trainer/rodygs_static.py:170-315,
                   rodygs_dynamic.py:150-197, src/trainer/utils.py:36-95) -- RUN on the model and optimizer the
                   reference builds from a checkpoint (as G12), with statistics set by hand; every ``device="cuda"``
                   factory call is redirected to the CPU by a TorchFunctionMode, and torch.normal(mean, std) is
                   served as mean + std * z with the standard-normal z recorded (the split samples); two cases
                   (no screen-size limit / max_screen_size 20 with the world-size prune firing)
                   (``... make_golden.py densify``)
Nothing from the reference is copied: only inputs and the outputs it produced are stored.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("simple_knn")
    mod("simple_knn._C", distCUDA2=None)
    mod("diff_gauss_pose", GaussianRasterizationSettings=None, GaussianRasterizer=None)
    mod("omegaconf", DictConfig=dict, OmegaConf=None)
    mod("plyfile", PlyData=None, PlyElement=None)
    # pytorch3d (un-vendored) is replaced by the repository's brute-force CPU restatement of its two ops, so that
    # the reference's own RigidityLoss code can run here and pin the loss arithmetic built on top of them
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import knn_oracle as KO
    mod("pytorch3d")
    mod("pytorch3d.ops", knn_points=KO.knn_points_batched, knn_gather=KO.knn_gather)


def main():
    sys.dont_write_bytecode = True
    _stub_modules()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    np.random.seed(0)

    # ---- G1 deformation --------------------------------------------------------------------------------------
    from src.model.rodygs_dynamic import MLPBasisNetwork
    net = MLPBasisNetwork(128, 16, 26, False, activation="gelu")
    # non-trivial weights (reference init is N(0,1e-2) with zero bias: outputs ~1e-9; scale up for a useful test)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() > 1 else 0.1))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    P, T = 300, 9
    coeff = (0.5 * torch.randn(P, 1, 16)).requires_grad_(True)
    times = torch.sort(torch.rand(T))[0]
    tind = torch.randint(0, T, (P,))
    t_now = torch.tensor(0.37)
    spatial = 2.3
    emb_now = net.t_embedder(t_now)
    embs = torch.stack([net.t_embedder(t) for t in times]).squeeze()
    trans, rot = net(coeff, t_now)
    table = net.batch_inference(embs).squeeze()
    delta = (coeff @ table[tind]).squeeze()
    trans2 = (trans - delta[..., :3]) * spatial
    rot2 = rot - delta[..., 3:]
    wx, wr = torch.randn(P, 3), torch.randn(P, 4)
    loss = (trans2 * wx).sum() + (rot2 * wr).sum()
    grads = torch.autograd.grad(loss, [coeff] + list(net.parameters()))
    np.savez_compressed(
        os.path.join(OUT, "deform_golden.npz"),
        **{"sd." + k: v.numpy() for k, v in sd.items()},
        coeff=coeff.detach().numpy(), times=times.numpy(), time_ind=tind.numpy(), t_now=t_now.numpy(),
        spatial=np.float32(spatial), emb_now=emb_now.detach().numpy(), embs=embs.detach().numpy(),
        trans_fwd=trans.detach().numpy(), rot_fwd=rot.detach().numpy(), table=table.detach().numpy(),
        trans=trans2.detach().numpy(), rot=rot2.detach().numpy(), wx=wx.numpy(), wr=wr.numpy(),
        d_coeff=grads[0].numpy(),
        **{"dsd." + n: g.numpy() for (n, _), g in zip(net.named_parameters(), grads[1:])})

    # ---- G2 camera -------------------------------------------------------------------------------------------
    from src.data.utils import FixedCameraTorch
    from src.utils.graphic_utils import getProjectionMatrix, focal2fov, fov2focal
    cams = []
    for i in range(4):
        q = torch.randn(4)
        t = torch.randn(3)
        fovx = float(np.deg2rad(40 + 10 * i))
        Wd, Hd = 320 + 64 * i, 200 + 40 * i
        fovy = focal2fov(fov2focal(fovx, Wd), Hd)
        cam = FixedCameraTorch(q, t, fovx, fovy, torch.zeros(3, Hd, Wd), "x", 0.1 * i, None, None, None, None, i)
        cams.append(dict(q=q.numpy(), t=t.numpy(), fovx=fovx, fovy=fovy, W=Wd, H=Hd,
                         w2c=cam.world_view_transform.numpy(), proj=cam.projection_matrix.numpy(),
                         proj_fn=getProjectionMatrix(0.01, 100.0, fovx, fovy).numpy()))
    np.savez_compressed(os.path.join(OUT, "camera_golden.npz"),
                        **{f"{k}{i}": np.asarray(v) for i, c in enumerate(cams) for k, v in c.items()})

    # ---- G3 SH -----------------------------------------------------------------------------------------------
    from src.utils.sh_utils import eval_sh, RGB2SH, SH2RGB
    n = 257
    sh = torch.randn(n, 3, 16)          # reference layout [..., C, K]
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=1)
    sh_out = {f"deg{d}": eval_sh(d, sh, dirs).numpy() for d in range(4)}
    rgb = torch.rand(11, 3)
    np.savez_compressed(os.path.join(OUT, "sh_golden.npz"), sh=sh.numpy(), dirs=dirs.numpy(), rgb=rgb.numpy(),
                        rgb2sh=RGB2SH(rgb).numpy(), sh2rgb=SH2RGB(rgb).numpy(), **sh_out)

    # ---- G4 covariance ---------------------------------------------------------------------------------------
    import src.utils.general_utils as GU
    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    torch.zeros = cpu_zeros
    try:
        from src.model.rodygs_static import build_covariance_from_scaling_rotation
        s = torch.exp(torch.randn(64, 3) * 0.5)
        r = torch.randn(64, 4)
        cov_norm = build_covariance_from_scaling_rotation(s, 1.3, r)                     # normalises r inside
        rn = torch.nn.functional.normalize(r)
        R = GU.build_rotation(r)
    finally:
        torch.zeros = real_zeros
    np.savez_compressed(os.path.join(OUT, "cov_golden.npz"), scales=s.numpy(), rot_raw=r.numpy(),
                        rot_unit=rn.numpy(), scale_modifier=np.float32(1.3), cov6=cov_norm.numpy(), R=R.numpy())

    # ---- G6 losses -------------------------------------------------------------------------------------------
    from src.utils.loss_utils import l1_loss, ssim
    a = torch.rand(3, 48, 64)
    b = (a + 0.1 * torch.randn(3, 48, 64)).clamp(0, 1)
    np.savez_compressed(os.path.join(OUT, "loss_golden.npz"), a=a.numpy(), b=b.numpy(),
                        l1=l1_loss(a, b).numpy(), ssim=ssim(a, b).numpy())
    rigidity_golden()
    depth_loss_golden()
    motion_reg_golden()
    print("golden vectors written to", OUT)


class _FakeDynModel:
    """The attributes RigidityLoss reads from DynRoDyGS (rodygs_dynamic.py:47,140-161)."""

    def __init__(self, xyz, coeff, fdc, table):
        self._xyz, self._motion_coeff, self._features_dc = xyz, coeff, fdc
        self.temporal_motion_table = table
        self.unique_times = list(range(table.shape[0]))

    def get_motion_for_times(self, timesteps, time_indices=None):
        return self.temporal_motion_table[time_indices]


def rigidity_golden():
    import random
    from src.trainer.losses import RigidityLoss
    g = torch.Generator().manual_seed(4242)
    P, Tu, B = 600, 12, 16
    xyz = (torch.rand(P, 3, generator=g) * 4 - 2).requires_grad_(True)
    transl = (0.05 * torch.randn(P, 3, generator=g)).requires_grad_(True)
    coeff = (0.3 * torch.randn(P, 1, B, generator=g)).requires_grad_(True)
    fdc = torch.rand(P, 1, 3, generator=g).requires_grad_(True)
    table = (0.2 * torch.randn(Tu, B, 7, generator=g)).requires_grad_(True)
    out = dict(xyz=xyz.detach().numpy(), transl=transl.detach().numpy(), coeff=coeff.detach().numpy(),
               fdc=fdc.detach().numpy(), table=table.detach().numpy())
    cases = {"coeff": dict(mode=["coeff"]),
             "coeff_l1_nocolor": dict(mode=["coeff"], sim_metric="l1", color_sim=False),
             "all": dict(mode=["coeff", "surface", "distance_preserving"], K=8, scale=2)}
    for name, kw in cases.items():
        random.seed(99)
        torch.manual_seed(7)
        model = _FakeDynModel(xyz, coeff, fdc, table)
        loss = RigidityLoss(**kw)(model, transl)
        grads = torch.autograd.grad(loss, [xyz, transl, coeff, fdc, table], allow_unused=True)
        out[name + ".loss"] = loss.detach().numpy()
        for k, gr in zip(("xyz", "transl", "coeff", "fdc", "table"), grads):
            out[f"{name}.d_{k}"] = (torch.zeros(1) if gr is None else gr).numpy()
    np.savez_compressed(os.path.join(OUT, "rigidity_golden.npz"), **out)


def depth_loss_golden():
    from src.trainer.losses import GlobalPearsonDepthLoss, LocalPearsonDepthLoss
    g = torch.Generator().manual_seed(515)
    H, W, box_p, p_corr = 150, 260, 32, 0.5
    gt = torch.rand(1, H, W, generator=g) * 8 + 1
    pred0 = gt * 0.7 + 0.5 + 0.6 * torch.randn(1, H, W, generator=g)
    motion = torch.rand(1, H, W, generator=g) > 0.6
    motion[:, :70, :90] = False          # some boxes have an empty "dynamic" mask -> the reference skips them
    out = dict(pred=pred0.numpy(), gt=gt.numpy(), motion=motion.numpy(), box_p=np.int64(box_p), p_corr=np.float64(p_corr))
    drawn = []
    real_randint = torch.randint

    def cpu_randint(*a, **k):
        k.pop("device", None)
        r = real_randint(*a, **k)
        drawn.append(r.clone())
        return r

    for mode in (None, "static", "dynamic"):
        tag = str(mode)
        mm = None if mode is None else motion
        pred = pred0.clone().requires_grad_(True)
        lg = GlobalPearsonDepthLoss(mode)(pred, gt, mm)
        (dg,) = torch.autograd.grad(lg, pred)
        out[f"global.{tag}.loss"], out[f"global.{tag}.d_pred"] = lg.detach().numpy(), dg.numpy()
        pred = pred0.clone().requires_grad_(True)
        torch.manual_seed(31)
        del drawn[:]
        torch.randint = cpu_randint
        try:
            ll = LocalPearsonDepthLoss(box_p, p_corr, mode)(pred, gt, mm)
        finally:
            torch.randint = real_randint
        (dl,) = torch.autograd.grad(ll, pred)
        out[f"local.{tag}.loss"], out[f"local.{tag}.d_pred"] = ll.detach().numpy(), dl.numpy()
        out[f"local.{tag}.rows"], out[f"local.{tag}.cols"] = drawn[0].numpy(), drawn[1].numpy()
    np.savez_compressed(os.path.join(OUT, "depth_loss_golden.npz"), **out)


def motion_reg_golden():
    from src.trainer import losses as RL
    g = torch.Generator().manual_seed(909)
    P, Tu, B = 257, 9, 16
    coeff = (0.3 * torch.randn(P, 1, B, generator=g)).requires_grad_(True)
    table = (0.4 * torch.randn(Tu, B, 7, generator=g)).requires_grad_(True)

    class M:
        _motion_coeff = coeff

        @staticmethod
        def get_total_motion_table():
            return table

    out = dict(coeff=coeff.detach().numpy(), table=table.detach().numpy())
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        cases = {"l1": RL.MotionL1Loss(), "sparsity": RL.MotionSparsityLoss(),
                 "basis_d0": RL.MotionBasisRegularizaiton(transl_degree=0),
                 "basis_d1_gauss": RL.MotionBasisRegularizaiton(transl_degree=1, rot_degree=1, freq_div_mode="gaussian")}
        for name, mod in cases.items():
            v = mod(M)
            gc, gt = torch.autograd.grad(v, [coeff, table], allow_unused=True)
            out[name + ".loss"] = v.detach().numpy()
            out[name + ".d_coeff"] = (torch.zeros(1) if gc is None else gc).numpy()
            out[name + ".d_table"] = (torch.zeros(1) if gt is None else gt).numpy()
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(OUT, "motion_reg_golden.npz"), **out)


def eval_pose_golden():
    from src.utils import graphic_utils as GU
    from src.utils.loss_utils import l2_loss
    from src.evaluator.utils import search_nearest_two
    from src.data.utils import LearnableCamera
    g = torch.Generator().manual_seed(1010)
    q = torch.randn(64, 4, generator=g)
    q[5] = torch.tensor([1e-3, 1.0, 0.0, 0.0]); q[6] = torch.tensor([0.0, 0.0, 1.0, 1e-4]); q[7] = torch.tensor([0.0, 0.0, 0.0, -1.0])
    R = GU.quaternion_to_matrix(q)
    out = dict(R=R.numpy(), quat=GU.matrix_to_quaternion(R).numpy())
    poses = torch.eye(4).repeat(12, 1, 1)
    poses[:, :3, 3] = 3.0 * torch.randn(12, 3, generator=g)
    query = torch.eye(4)
    query[:3, 3] = torch.randn(3, generator=g)
    out.update(db_poses=poses.numpy(), query_pose=query.numpy(), nearest=search_nearest_two(query, poses).numpy())
    a, b = torch.rand(3, 7, 9, generator=g), torch.rand(3, 7, 9, generator=g)
    out.update(l2_a=a.numpy(), l2_b=b.numpy(), l2=l2_loss(a, b).numpy())
    # LearnableCamera: W2C rotation/translation in -> (R_c2w_quat, T_c2w) parameters -> world_view_transform
    Rw2c = GU.quaternion_to_matrix(torch.nn.functional.normalize(torch.randn(4, generator=g), dim=0)).numpy().astype(np.float64)
    Tw2c = torch.randn(3, generator=g).numpy().astype(np.float64)
    cam = LearnableCamera(Rw2c, Tw2c, 0.9, 0.6, torch.zeros(3, 8, 10), "x", 0.0, None, None, None, 0)
    out.update(cam_R_w2c=Rw2c, cam_T_w2c=Tw2c, cam_quat=cam.R_c2w_quat.detach().numpy(),
               cam_t=cam.T_c2w.detach().numpy(), cam_w2c=cam.world_view_transform.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "eval_pose_golden.npz"), **out)


def optimizer_golden():
    """G11: the two optimiser-side pieces of the training loop that touch the flat bucket between steps --
    get_expon_lr_func (src/utils/general_utils.py:40-73; xyz schedule of rodygs_static.py:143-149) and the
    reset_opacity arithmetic (rodygs_static.py:151-160: inverse_sigmoid(min(sigmoid(logit), 0.01)), with
    replace_tensor_to_optimizer zeroing both Adam moments, src/trainer/utils.py:15-32, run here on a real
    torch.optim.Adam so the stored state is the reference's own)."""
    from src.utils.general_utils import get_expon_lr_func, inverse_sigmoid
    from src.trainer.utils import replace_tensor_to_optimizer
    steps = np.array([0, 1, 10, 100, 999, 1000, 15000, 29999, 30000, 40000], dtype=np.int64)
    out = dict(lr_steps=steps)
    for tag, kw in (("a", dict(lr_init=0.00016 * 5.0, lr_final=0.0000016 * 5.0, lr_delay_mult=0.01, max_steps=30000)),
                    ("b", dict(lr_init=1e-3, lr_final=1e-5, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000))):
        fn = get_expon_lr_func(**kw)
        out["lr_" + tag] = np.array([fn(int(s)) for s in steps], dtype=np.float64)
        out["lr_kw_" + tag] = np.array([kw["lr_init"], kw["lr_final"], kw.get("lr_delay_steps", 0), kw["lr_delay_mult"],
                                        kw["max_steps"]], dtype=np.float64)
    g = torch.Generator().manual_seed(1111)
    logit = torch.nn.Parameter(4.0 * torch.randn(4097, 1, generator=g))
    opt = torch.optim.Adam([{"params": [logit], "lr": 0.05, "name": "opacity"}], lr=0.0, eps=1e-15)
    for _ in range(3):                       # give the state non-zero moments and a step count
        opt.zero_grad()
        (torch.sigmoid(logit) * torch.linspace(-1, 1, 4097).unsqueeze(1)).sum().backward()
        opt.step()
    before = logit.detach().clone()
    st = opt.state[logit]
    out.update(reset_logit_in=before.numpy(), reset_m_in=st["exp_avg"].numpy().copy(),
               reset_v_in=st["exp_avg_sq"].numpy().copy(), reset_step=np.int64(int(st["step"])))
    get_opacity = torch.sigmoid(before)
    new = inverse_sigmoid(torch.min(get_opacity, torch.ones_like(get_opacity) * 0.01))
    res = replace_tensor_to_optimizer(opt, new, "opacity")["opacity"]
    st2 = opt.state[res]
    # .copy(): the arrays would otherwise alias tensors that the optimiser step below updates in place
    out.update(reset_logit_out=res.detach().numpy().copy(), reset_m_out=st2["exp_avg"].numpy().copy(),
               reset_v_out=st2["exp_avg_sq"].numpy().copy(), reset_step_out=np.int64(int(st2["step"])))
    # one more Adam step from the reset state: what the bucket must look like after the next optimiser step
    res.grad = torch.linspace(1, -1, 4097).unsqueeze(1) * 0.3
    opt.step()
    out.update(next_grad=res.grad.numpy().copy(), next_logit=res.detach().numpy().copy(),
               next_m=st2["exp_avg"].numpy().copy(), next_v=st2["exp_avg_sq"].numpy().copy())
    np.savez_compressed(os.path.join(OUT, "optimizer_golden.npz"), **out)

CKPT_P, CKPT_K, CKPT_T = 257, 16, 7
CKPT_SCALE = 4.2
# the shipped training configuration (/root/reference/configs/train/train_kubric_mrig.yaml, dynamic model)
CKPT_LR = dict(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
               position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)
CKPT_DEFORM = dict(deform_lr_init=0.0016, deform_lr_final=0.00016, deform_lr_delay_mult=0.01, deform_lr_max_steps=30000,
                   motion_coeff_lr=0.00016)


def checkpoint_inputs(seed=1212):
    """The build's side of G12 (shared with tests/test_abi_and_host.py, which imports this function -- it touches no
    reference code): flat buckets with random values, moments and a step count, the MLP in its small bucket."""
    from rodygs_amd.deform import MLPBasisNetwork as Net
    from rodygs_amd.dp import FlatParams
    from rodygs_amd.trainstep import bind_module_to_flat
    P, K, T = CKPT_P, CKPT_K, CKPT_T
    g = torch.Generator().manual_seed(seed)
    spec = {"xyz": ((P, 3), CKPT_LR["position_lr_init"] * CKPT_SCALE), "features": ((P, K, 3), CKPT_LR["feature_lr"]),
            "scaling": ((P, 3), CKPT_LR["scaling_lr"]), "rotation": ((P, 4), CKPT_LR["rotation_lr"]),
            "opacity": ((P, 1), CKPT_LR["opacity_lr"]), "motion_coeff": ((P, 1, 16), CKPT_DEFORM["motion_coeff_lr"])}
    fp = FlatParams(spec, "cpu")
    with torch.no_grad():
        fp.flat.copy_(0.5 * torch.randn(fp.numel, generator=g))
        fp.exp_avg.copy_(1e-3 * torch.randn(fp.numel, generator=g))
        fp.exp_avg_sq.copy_(1e-6 * torch.rand(fp.numel, generator=g))
    fp.step_count = 5
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed + 1)
        net = Net(128, 16, 26, False)
        with torch.no_grad():
            for p_ in net.parameters():
                p_.copy_(torch.randn_like(p_) * (0.3 if p_.dim() > 1 else 0.1))
    sp = bind_module_to_flat(net, CKPT_DEFORM["deform_lr_init"], "cpu")
    with torch.no_grad():
        sp.exp_avg.copy_(1e-3 * torch.randn(sp.numel, generator=g))
        sp.exp_avg_sq.copy_(1e-6 * torch.rand(sp.numel, generator=g))
    sp.step_count = 5
    times = torch.arange(T, dtype=torch.float32) / T
    g2t = times[torch.randint(0, T, (P,), generator=g)]
    cams = (torch.randn(T, 4, generator=g), torch.randn(T, 3, generator=g))
    return fp, net, sp, g2t, cams, g


def checkpoint_golden():
    import tempfile
    from src.model.rodygs_dynamic import DynRoDyGS
    from src.trainer.rodygs_dynamic import DynTrainer
    from rodygs_amd import checkpoint as CK
    fp, net, sp, g2t, cams, g = checkpoint_inputs()
    sd = CK.export_state_dict(fp, 5, 3, CKPT_SCALE, net, g2t, cams, feature_lr_rest=CKPT_LR["feature_lr"] / 20.0,
                              deform_state=sp, deform_lr=CKPT_DEFORM["deform_lr_init"])
    with tempfile.TemporaryDirectory() as td:
        CK.save_checkpoint(os.path.join(td, "dynamic_last.ckpt"), sd)
        loaded = torch.load(os.path.join(td, "dynamic_last.ckpt"), weights_only=False)     # as evaluator/eval.py:58
    assert isinstance(loaded, tuple) and loaded[1] == 5
    loaded = loaded[0]
    # ---- the reference's loader, with its hard-coded .cuda() calls neutralised for the duration ----
    t_cuda, m_cuda = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        model = DynRoDyGS(3, 128, 26, False, 16, inverse_motion=True)
        model.create_from_state_dict(loaded, CKPT_SCALE)
        out = dict(time_ind=model.gaussian_to_time_ind.numpy().astype(np.int64),
                   real_times=model.real_times.numpy(), unique_keys=np.array(model.unique_times, dtype=np.int64),
                   table=model.get_total_motion_table().detach().numpy().copy())
        for i, t in enumerate((0.0, 0.37)):
            tr, ro = model.get_gaussian_deformation(torch.tensor(t))
            out[f"deform_xyz_{i}"] = tr.detach().numpy().copy()
            out[f"deform_rot_{i}"] = ro.detach().numpy().copy()
        # .copy(): get_xyz IS the parameter -- the array would alias storage the optimizer step below updates in place
        out.update(get_xyz=model.get_xyz.detach().numpy().copy(), get_features=model.get_features.detach().numpy().copy(),
                   get_opacity=model.get_opacity.detach().numpy().copy(),
                   get_scaling=model.get_scaling.detach().numpy().copy(),
                   get_rotation=model.get_rotation.detach().numpy().copy())
        # ---- the reference's optimizer: its own group construction, our state loaded into it, one step ----
        tr_ = object.__new__(DynTrainer)
        tr_.model, tr_.spatial_lr_scale, tr_.is_optimizable_cam = model, CKPT_SCALE, False
        tr_.optim_setup(**CKPT_LR)
        tr_.append_motion_optim(**CKPT_DEFORM)
        names = [g_["name"] for g_ in tr_.optimizer.param_groups]
        assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "deform_network", "motion_coeff"], names
        assert [len(g_["params"]) for g_ in tr_.optimizer.param_groups] == [1, 1, 1, 1, 1, 1, 70, 1]
        tr_.optimizer.load_state_dict(loaded["optim"]["optimizer"])
        out["group_lr"] = np.array([g_["lr"] for g_ in tr_.optimizer.param_groups], dtype=np.float64)
        flat_params = [q for g_ in tr_.optimizer.param_groups for q in g_["params"]]
        grads = []
        for q in flat_params:
            q.grad = 1e-2 * torch.randn(q.shape, generator=g)
            grads.append(q.grad.numpy().copy())
        tr_.optimizer.step()
        tr_.max_radii2D = torch.arange(CKPT_P, dtype=torch.float32)
        tr_.xyz_gradient_accum = torch.full((CKPT_P, 1), 0.25)
        tr_.denom = torch.full((CKPT_P, 1), 2.0)
        ref_sd = tr_.state_dict(6)                         # DynTrainer.state_dict (rodygs_dynamic.py:217-222)
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = t_cuda, m_cuda
    assert set(ref_sd) == {"iteration", "active_sh_degree", "model", "optim", "spatial_lr_scale"}
    for i, a in enumerate(grads):
        out[f"grad_{i}"] = a
    # the reference-written checkpoint, as plain arrays (import direction): model tensors, MLP state_dict, optimizer state
    for k, v in ref_sd["model"].items():
        if k == "_deform_network":
            for kk, vv in v.items():
                out["ref_mlp." + kk] = vv.detach().numpy().copy()
        else:
            out["ref_model." + k] = v.detach().numpy().copy()
    ro = ref_sd["optim"]["optimizer"]
    out["ref_group_sizes"] = np.array([len(g_["params"]) for g_ in ro["param_groups"]], dtype=np.int64)
    out["ref_group_lr"] = np.array([g_["lr"] for g_ in ro["param_groups"]], dtype=np.float64)
    for i, st in ro["state"].items():
        out[f"ref_state_{i}.step"] = np.float64(float(st["step"]))
        out[f"ref_state_{i}.exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"ref_state_{i}.exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    out["ref_n_state"] = np.int64(len(ro["state"]))
    out["ref_iteration"] = np.int64(ref_sd["iteration"])
    out["ref_active_sh_degree"] = np.int64(ref_sd["active_sh_degree"])
    np.savez_compressed(os.path.join(OUT, "checkpoint_golden.npz"), **out)


def densify_golden():
    """G13: the reference's own densify-and-prune, run here on the CPU (see the module docstring)."""
    import tempfile
    from torch.overrides import TorchFunctionMode
    from src.model.rodygs_dynamic import DynRoDyGS
    from src.trainer.rodygs_dynamic import DynTrainer
    from rodygs_amd import checkpoint as CK

    class OnCpu(TorchFunctionMode):
        """device="cuda" -> "cpu" on every torch call; torch.normal(mean=, std=) -> mean + std * z, z recorded."""

        def __init__(self, gen):
            super().__init__()
            self.gen, self.z = gen, []

        def __torch_function__(self, func, types, args=(), kwargs=None):
            kwargs = dict(kwargs or {})
            if str(kwargs.get("device", "")).startswith("cuda"):
                kwargs["device"] = "cpu"
            if func is torch.normal:
                mean, std = kwargs["mean"], kwargs["std"]
                z = torch.randn(std.shape, generator=self.gen)
                self.z.append(z)
                return mean + std * z
            return func(*args, **kwargs)

    out = {}
    cases = {"a": dict(seed=1313, percent_dense=0.01, max_screen_size=None, min_opacity=0.3, max_grad=0.0002),
             "b": dict(seed=1414, percent_dense=0.08, max_screen_size=20, min_opacity=0.45, max_grad=0.0003)}
    for tag, c in cases.items():
        fp, net, sp, g2t, cams, g = checkpoint_inputs(seed=c["seed"])
        P = CKPT_P
        sd = CK.export_state_dict(fp, 5, 3, CKPT_SCALE, net, g2t, cams, feature_lr_rest=CKPT_LR["feature_lr"] / 20.0,
                                  deform_state=sp, deform_lr=CKPT_DEFORM["deform_lr_init"])
        with tempfile.TemporaryDirectory() as td:
            CK.save_checkpoint(os.path.join(td, "dynamic_last.ckpt"), sd)
            loaded = torch.load(os.path.join(td, "dynamic_last.ckpt"), weights_only=False)[0]
        t_cuda, m_cuda = torch.Tensor.cuda, torch.nn.Module.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        try:
            model = DynRoDyGS(3, 128, 26, False, 16, inverse_motion=True)
            model.create_from_state_dict(loaded, CKPT_SCALE)
            tr_ = object.__new__(DynTrainer)
            tr_.model, tr_.spatial_lr_scale, tr_.is_optimizable_cam = model, CKPT_SCALE, False
            tr_.optim_setup(**CKPT_LR)
            tr_.append_motion_optim(**CKPT_DEFORM)
            tr_.optimizer.load_state_dict(loaded["optim"]["optimizer"])
            tr_.percent_dense = c["percent_dense"]
            # statistics: a third of the Gaussians never seen (denom 0 -> NaN average -> 0), gradients around the threshold
            denom = torch.randint(0, 4, (P, 1), generator=g).float()
            accum = torch.rand(P, 1, generator=g) * 3.0 * c["max_grad"] * denom
            radii = torch.rand(P, generator=g) * 40
            tr_.xyz_gradient_accum, tr_.denom, tr_.max_radii2D = accum.clone(), denom.clone(), radii.clone()
            max_s = model.get_scaling.max(dim=1).values.detach()
            extent = float(max_s.median()) / c["percent_dense"]          # half clone candidates, half split candidates
            names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "motion_coeff")
            single = {g_["name"]: g_["params"][0] for g_ in tr_.optimizer.param_groups if len(g_["params"]) == 1}
            assert tuple(single) == names
            for k in names:
                st_ = tr_.optimizer.state[single[k]]
                out[f"{tag}.in.{k}"] = single[k].detach().numpy().copy()
                out[f"{tag}.in.exp_avg.{k}"] = st_["exp_avg"].numpy().copy()
                out[f"{tag}.in.exp_avg_sq.{k}"] = st_["exp_avg_sq"].numpy().copy()
            out[f"{tag}.in.accum"], out[f"{tag}.in.denom"], out[f"{tag}.in.max_radii"] = accum.numpy(), denom.numpy(), radii.numpy()
            out[f"{tag}.in.gaussian_to_time"] = model.gaussian_to_time.numpy().copy()
            out[f"{tag}.in.gaussian_to_time_ind"] = model.gaussian_to_time_ind.numpy().astype(np.int64)
            mode = OnCpu(g)
            with mode:
                tr_.densify_and_prune(c["max_grad"], c["min_opacity"], extent, c["max_screen_size"])
            assert len(mode.z) == 1                                       # one torch.normal call: the split samples
        finally:
            torch.Tensor.cuda, torch.nn.Module.cuda = t_cuda, m_cuda
        out[f"{tag}.z"] = mode.z[0].numpy()
        out[f"{tag}.args"] = np.array([c["max_grad"], c["min_opacity"], extent, c["max_screen_size"] or 0,
                                       c["percent_dense"], 2], dtype=np.float64)
        single = {g_["name"]: g_["params"][0] for g_ in tr_.optimizer.param_groups if len(g_["params"]) == 1}
        attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                "scaling": "_scaling", "rotation": "_rotation", "motion_coeff": "_motion_coeff"}
        for k in names:
            assert getattr(model, attr[k]) is single[k]                  # the model's attributes ARE the optimizer's params
            st_ = tr_.optimizer.state[single[k]]
            out[f"{tag}.out.{k}"] = single[k].detach().numpy().copy()
            out[f"{tag}.out.exp_avg.{k}"] = st_["exp_avg"].numpy().copy()
            out[f"{tag}.out.exp_avg_sq.{k}"] = st_["exp_avg_sq"].numpy().copy()
            out[f"{tag}.out.step.{k}"] = np.float64(float(st_["step"]))
        out[f"{tag}.out.accum"], out[f"{tag}.out.denom"] = tr_.xyz_gradient_accum.numpy().copy(), tr_.denom.numpy().copy()
        out[f"{tag}.out.max_radii"] = tr_.max_radii2D.numpy().copy()
        out[f"{tag}.out.gaussian_to_time"] = model.gaussian_to_time.numpy().copy()
        out[f"{tag}.out.gaussian_to_time_ind"] = model.gaussian_to_time_ind.numpy().astype(np.int64)
        Pn = out[f"{tag}.out.xyz"].shape[0]
        print(f"G13 case {tag}: P {P} -> {Pn}, split samples {tuple(mode.z[0].shape)}, extent {extent:.3f}")
    np.savez_compressed(os.path.join(OUT, "densify_golden.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("rigidity", "depth", "motion", "pose", "optimizer", "checkpoint", "densify"):
        sys.dont_write_bytecode = True
        _stub_modules()
        sys.path.insert(0, REF)
        {"rigidity": rigidity_golden, "depth": depth_loss_golden, "motion": motion_reg_golden,
         "pose": eval_pose_golden, "optimizer": optimizer_golden, "checkpoint": checkpoint_golden,
         "densify": densify_golden}[sys.argv[1]]()
        print(sys.argv[1], "golden written to", OUT)
    else:
        main()
