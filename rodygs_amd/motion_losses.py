"""Host mirrors of the reference's motion regularisers (/root/reference/src/trainer/losses.py:363-525), the small
losses of the dynamic sub-step of config 5 next to the photometric, depth and rigidity terms
(/root/reference/configs/train/train_kubric_mrig.yaml:186-232).  They are elementwise torch expressions over the
motion coefficients [P,1,B] and the birth-time motion table [Tu,B,7]; nothing here is a kernel.  Pinned by
tests/golden/motion_reg_golden.npz (imported reference, make_golden.py G9).

Same class names (including the reference's spelling ``MotionBasisRegularizaiton``), constructors and
``forward(model, **kwargs)``: ``model`` needs ``_motion_coeff`` and ``get_total_motion_table()``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

# frequency-division weights of the reference (losses.py:386-470); only "vanilla" is used by the shipped configs, the
# others are kept so that a config naming them behaves the same
_BANK = {
    "gaussian": [2.368737348178644, 2.3218332060968687, 2.186620166400238, 1.9785357455909518, 1.7200563444604107,
                 1.4367118264767467, 1.1529882480025957, 0.8890134170352768, 0.6585973377702478, 0.4687700396753248,
                 0.3205737399288996, 0.2106319563365025, 0.13296850925636292, 0.08064947764026723,
                 0.04699834214974086, 0.026314295000921823],
    "sigmoid": [0.0, 0.006057306357564347, 0.019407599012746118, 0.04848852855754725, 0.11024831053568876,
                0.23462085565239668, 0.4602813915432914, 0.8016437593070956, 1.1983562406929047, 1.539718608456709,
                1.7653791443476032, 1.889751689464311, 1.9515114714424528, 1.9805924009872535, 1.9939426936424351, 2.0],
    "laplacian": [3.0235547043507864, 2.475477220065594, 2.0267493286116927, 1.6593620041145454, 1.3585707032576908,
                  1.112303614987853, 0.910677176350366, 0.7455994104042655, 0.6104451667747834, 0.49979023110633275,
                  0.40919363229470634, 0.3350194107233597, 0.274290694437278, 0.22457022681891523,
                  0.18386255092234366, 0.15053392477948924],
    "cum_exponential": [0.24858106424723717, 0.45210202617930384, 0.6187308966091, 0.7551550771806206,
                        0.8668497492779882, 0.9582976122790642, 1.0331687900213073, 1.0944681257580495,
                        1.1446557770689725, 1.1857459506219796, 1.219387739359138, 1.246931306386802,
                        1.2694820717618154, 1.2879450768797849, 1.3030613069641026, 1.3154374294047362],
}


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """/root/reference/src/utils/graphic_utils.py:76-102 (two_s = 2 / |q|^2), batched [...,4] -> [...,3,3]."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3))


class MotionL1Loss(nn.Module):
    def forward(self, model, **kwargs):
        return model._motion_coeff.abs().mean()


class MotionSparsityLoss(nn.Module):
    def forward(self, model, **kwargs):
        a = torch.abs(model._motion_coeff)
        peak = torch.max(a, dim=2).values
        return (a / (peak[..., None] + 1e-7)).mean()


class MotionBasisRegularizaiton(nn.Module):
    """Finite-difference smoothness of the motion table over the sorted birth times: degree 0 = velocity, 1 =
    acceleration, 2 = jerk; translations by subtraction, rotations by relative rotation matrices."""

    def __init__(self, transl_degree=0, rot_degree=0, freq_div_mode="vanilla"):
        super().__init__()
        self.degree = {"transl": transl_degree, "rot": rot_degree}
        if freq_div_mode != "vanilla" and freq_div_mode not in _BANK:
            raise AssertionError(f"Invalid freq_div_mode : {freq_div_mode}")
        if freq_div_mode == "vanilla":
            w = torch.ones(16)
        else:
            w = torch.tensor(_BANK[freq_div_mode])
            w = w / w.max() * 1.3
        self.register_buffer("reg_coeff", w, persistent=False)

    @staticmethod
    def _derive(x: torch.Tensor, times: int) -> torch.Tensor:
        # the reference's recursion calls first_derivate_motion WITHOUT is_rot (losses.py:497-503): rotation matrices
        # are differenced by subtraction too; kept
        for _ in range(times):
            x = x[1:] - x[:-1]
        return x

    def forward(self, model, **kwargs):
        table = model.get_total_motion_table()                                   # [Tu,B,7]
        if table.is_cuda and table.shape[1] == 16 and max(self.degree.values()) <= 2:
            # GPU tensors: value and gradient in three tiny HIP launches (csrc/rdg_motionreg.hip) instead of ~160
            # framework launches over a tensor of a few thousand floats; the torch expression below is the same
            # arithmetic and serves CPU tensors (golden tests)
            return _FusedBasisReg.apply(table, tuple(float(x) for x in self.reg_coeff.tolist()),
                                        self.degree["transl"], self.degree["rot"])
        w = self.reg_coeff.to(table.device)
        transl, rot = table[..., :3], table[..., 3:]
        rot_m = quaternion_to_matrix(rot.reshape(-1, 4)).reshape(*table.shape[:-1], 3, 3)
        d_t = self._derive(transl, self.degree["transl"] + 1)
        d_r = self._derive(rot_m, self.degree["rot"] + 1)
        t_norm = (torch.norm(d_t, dim=-1) * w[None]).mean()
        r_norm = (torch.norm(torch.eye(3, device=table.device)[None, None] - d_r, dim=(-1, -2)) * w[None]).mean()
        if self.degree["transl"] < 0:
            t_norm = 0
        if self.degree["rot"] < 0:
            r_norm = 0
        return t_norm + r_norm


class _FusedL1Sparsity(torch.autograd.Function):
    """w_l1 * MotionL1Loss + w_sparsity * MotionSparsityLoss over coeff [P,1,16] in one HIP pass each way
    (csrc/rdg_motionreg.hip)."""

    @staticmethod
    def forward(ctx, coeff, w_l1, w_sparsity, grad_sink):
        from . import _lib
        L = _lib.lib()
        if not coeff.is_cuda:
            raise RuntimeError("rodygs_amd.fused_motion_l1_sparsity: tensors must be on the GPU (no CPU fallback exists)")
        c = coeff.detach().to(torch.float32).contiguous()
        P, B = c.shape[0], c.shape[-1]
        sums = torch.empty(2, dtype=torch.float64, device=c.device)
        with torch.cuda.device(c.device):
            _lib.check(L.rdg_motion_reg_forward(P, B, _lib.ptr(c), _lib.ptr(sums), _lib.stream_ptr()), "rdg_motion_reg_forward")
        ctx.save_for_backward(c)
        ctx.w, ctx.sink, ctx.shape = (float(w_l1), float(w_sparsity)), grad_sink, coeff.shape
        return ((w_l1 * sums[0] + w_sparsity * sums[1]) / float(max(P * B, 1))).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        L = _lib.lib()
        (c,) = ctx.saved_tensors
        P, B = c.shape[0], c.shape[-1]
        gl = g.detach().to(torch.float32).reshape(1).contiguous()
        sink = ctx.sink
        if sink is not None and (sink.numel() != c.numel() or not sink.is_contiguous() or sink.dtype != torch.float32):
            raise RuntimeError("fused_motion_l1_sparsity grad_sink must be a contiguous float32 tensor shaped like coeff")
        d = sink if sink is not None else torch.empty_like(c)
        with torch.cuda.device(c.device):
            _lib.check(L.rdg_motion_reg_backward(P, B, _lib.ptr(c), _lib.ptr(gl), ctx.w[0], ctx.w[1], _lib.ptr(d),
                                                 1 if sink is not None else 0, _lib.stream_ptr()), "rdg_motion_reg_backward")
        return (None if sink is not None else d.view(ctx.shape)), None, None, None


def fused_motion_l1_sparsity(coeff: torch.Tensor, w_l1: float, w_sparsity: float, grad_sink=None) -> torch.Tensor:
    """``w_l1 * MotionL1Loss()(model) + w_sparsity * MotionSparsityLoss()(model)`` for ``model._motion_coeff = coeff``
    [P,1,16].  ``grad_sink``: a tensor shaped like coeff that the backward ADDS the gradient into (e.g. the flat gradient
    bucket's segment) instead of returning it through autograd."""
    return _FusedL1Sparsity.apply(coeff, w_l1, w_sparsity, grad_sink)


class _FusedBasisReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, w, transl_degree, rot_degree):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        t = table.detach().to(torch.float32).contiguous()
        Tu, B = t.shape[0], t.shape[1]
        dev = t.device
        wa = (C.c_float * 16)(*w)
        with torch.cuda.device(dev):
            ws = torch.empty(L.rdg_basis_reg_ws_bytes(Tu), dtype=torch.uint8, device=dev)
            loss = torch.empty(1, dtype=torch.float64, device=dev)
            d_table = torch.empty_like(t)
            _lib.check(L.rdg_basis_reg(Tu, B, int(transl_degree), int(rot_degree), wa, _lib.ptr(t), _lib.ptr(ws),
                                       _lib.ptr(loss), _lib.ptr(d_table), _lib.stream_ptr()), "rdg_basis_reg")
        ctx.save_for_backward(d_table)
        ctx.shape = table.shape
        return loss[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (d_table,) = ctx.saved_tensors
        return (d_table * g).view(ctx.shape), None, None, None
