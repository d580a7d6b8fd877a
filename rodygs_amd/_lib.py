"""ctypes binding of librodygs_hip.so (the C-ABI in include/rodygs_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDG_LIB_PATH: load another build of the same ABI (A/B runs of kernel variants on one GPU box)
LIB_PATH = os.environ.get("RDG_LIB_PATH") or os.path.join(_HERE, "csrc", "librodygs_hip.so")
ABI_VERSION = 7

_lib = None
_lock = threading.Lock()


class RdgRasterSettings(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("M", C.c_int32), ("sh_degree", C.c_int32), ("image_height", C.c_int32),
        ("image_width", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("enable_cov_grad", C.c_int32), ("enable_sh_grad", C.c_int32), ("render_normal", C.c_int32),
        ("bin_mode", C.c_int32), ("num_rendered_stats", C.c_int32), ("list_hints", C.c_int32),
        ("grad_rows_zeroed", C.c_int32), ("densify_row0", C.c_int32), ("zero_grad_ws", C.c_void_p),
        ("num_rendered_host", C.c_void_p), ("densify_grad_accum", C.c_void_p), ("densify_denom", C.c_void_p),
        ("densify_max_radii", C.c_void_p), ("densify_rows", C.c_int32), ("cull", C.c_int32),
        ("num_rendered_max", C.c_void_p), ("aux_stream", C.c_void_p),
    ]


class RdgAdamSeg(C.Structure):
    _fields_ = [("n", C.c_int64), ("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("lr_head", C.c_float), ("lr_tail", C.c_float), ("row_len", C.c_int32),
                ("head_len", C.c_int32), ("grad2", C.c_void_p)]


class RdgStepScalars(C.Structure):
    _fields_ = [("inv_bias_correction1", C.c_float), ("sqrt_bias_correction2", C.c_float), ("frame", C.c_int32),
                ("lr_from_table", C.c_int32), ("seg_lr_head", C.c_float * 12), ("seg_lr_tail", C.c_float * 12),
                ("sh_lr_head", C.c_float), ("sh_lr_tail", C.c_float), ("reserved", C.c_int32 * 2)]


STEP_SCALARS_FLOATS = C.sizeof(RdgStepScalars) // 4      # a device RdgStepScalars as a float32 tensor of this many words


STAGES = {
    "preprocess": 0, "scan_dup": 1, "sort": 2, "ranges": 3, "render_fwd": 4, "render_bwd": 5,
    "preprocess_bwd": 6, "deform_fwd": 7, "deform_bwd": 8, "adam": 9, "loss_fwd": 10, "loss_bwd": 11, "mlp_fwd": 12, "mlp_bwd": 13,
}

_vp = C.c_void_p
_SIGS = {
    "rdg_abi_version": (C.c_int, []),
    "rdg_pose_fork_prepare": (C.c_int, []),
    "rdg_settings_bytes": (C.c_size_t, []),
    "rdg_last_error": (C.c_char_p, []),
    "rdg_geom_bytes": (C.c_size_t, [C.c_int32]),
    "rdg_binning_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "rdg_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rdg_grad_bytes": (C.c_size_t, [C.c_int32]),
    "rdg_sort_tmp_bytes": (C.c_size_t, [C.c_int64]),
    "rdg_knn_tmp_bytes": (C.c_size_t, [C.c_int32]),
    "rdg_rasterize_forward": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 12 + [C.c_int64] + [_vp] * 8),
    "rdg_rasterize_backward": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 13 + [C.c_int64] + [_vp] * 16),
    "rdg_preprocess_forward": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 13),
    "rdg_geom_from_records": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 4),
    "rdg_composite_forward": (C.c_int, [C.POINTER(RdgRasterSettings), _vp, _vp, _vp, _vp, C.c_int64] + [_vp] * 7),
    "rdg_composite_backward": (C.c_int, [C.POINTER(RdgRasterSettings), _vp, _vp, _vp, C.c_int64] + [_vp] * 7),
    "rdg_det_bytes": (C.c_size_t, [C.c_int64]),
    "rdg_composite_backward_det": (C.c_int, [C.POINTER(RdgRasterSettings), _vp, _vp, _vp, C.c_int64] + [_vp] * 7 +
                                   [C.c_int64, _vp]),
    "rdg_preprocess_backward": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 22),
    "rdg_preprocess_backward_adam": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 18 + [C.c_int32] + [C.c_float] * 2 +
                                     [C.c_double, C.c_double, C.c_float, C.c_int32, _vp]),
    "rdg_preprocess_backward_adam_dev": (C.c_int, [C.POINTER(RdgRasterSettings)] + [_vp] * 18 + [C.c_int32] + [C.c_float] * 2 +
                                         [C.c_double, C.c_double, C.c_float, _vp, _vp]),
    "rdg_pose_view_forward_dev": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "rdg_pose_view_backward_dev": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdg_adam_step_multi_dev": (C.c_int, [C.c_int32, C.POINTER(RdgAdamSeg), C.c_double, C.c_double, C.c_float, _vp, _vp]),
    "rdg_preprocess_forward_views": (C.c_int, [C.POINTER(RdgRasterSettings), C.c_int32, C.c_int32] + [_vp] * 10),
    "rdg_preprocess_forward_views_rows": (C.c_int, [C.POINTER(RdgRasterSettings), C.c_int32, C.c_int32, C.c_int32]
                                          + [_vp] * 10),
    "rdg_preprocess_backward_views": (C.c_int, [C.POINTER(RdgRasterSettings), C.c_int32, C.c_int32] + [_vp] * 18),
    "rdg_geom_export": (C.c_int, [C.c_int32] + [_vp] * 8),
    "rdg_image_export": (C.c_int, [C.c_int32, C.c_int32] + [_vp] * 4),
    "rdg_bin_forward": (C.c_int, [C.POINTER(RdgRasterSettings), _vp, _vp, _vp, C.c_int64] + [_vp] * 8),
    "rdg_sort_pairs": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int32, _vp, _vp]),
    "rdg_deform_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    "rdg_deform_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.c_float] + [_vp] * 10),
    "rdg_dyn_getter_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "rdg_dyn_getter_forward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_float] + [_vp] * 9),
    "rdg_dyn_getter_backward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_float] + [_vp] * 18),
    "rdg_dyn_getter_views_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "rdg_deform_sorted_views_ws_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rdg_dyn_getter_views_forward": (C.c_int, [C.c_int32] * 4 + [_vp, _vp, _vp, C.c_float] + [_vp] * 9),
    "rdg_dyn_getter_views_backward": (C.c_int, [C.c_int32] * 4 + [_vp, _vp, _vp, C.c_float] + [_vp] * 17),
    "rdg_deform_sorted_ws_bytes": (C.c_size_t, [C.c_int32]),
    "rdg_dist2_knn3": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp]),
    "rdg_gather_rows": (C.c_int, [C.c_int64, C.c_int32] + [_vp] * 4),
    "rdg_split_children": (C.c_int, [C.c_int64, C.c_int32] + [_vp] * 8),
    "rdg_mask_rank_ws_bytes": (C.c_size_t, [C.c_int64]),
    "rdg_mask_rank": (C.c_int, [C.c_int64, _vp, _vp, _vp, _vp]),
    "rdg_densify_stats": (C.c_int, [C.c_int64, C.c_int64] + [_vp] * 6),
    "rdg_reset_opacity": (C.c_int, [C.c_int64, C.c_float, _vp, _vp, _vp, _vp]),
    "rdg_morton_codes": (C.c_int, [C.c_int64, _vp, _vp, C.c_int32, _vp, _vp]),
    "rdg_graph_points_backward": (C.c_int, [C.c_int64, C.c_int32] + [_vp] * 8),
    "rdg_graph_surface": (C.c_int, [C.c_int64, C.c_int32] + [_vp] * 9),
    "rdg_rigidity_pack_rows": (C.c_int, [C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "rdg_rigidity_dp_rows": (C.c_int, [C.c_int64, C.c_int32, C.c_int32] + [_vp] * 7 + [C.c_float] + [_vp] * 7),
    "rdg_motion_reg_forward": (C.c_int, [C.c_int64, C.c_int32, _vp, _vp, _vp]),
    "rdg_motion_reg_backward": (C.c_int, [C.c_int64, C.c_int32, _vp, _vp, C.c_float, C.c_float, _vp, C.c_int32, _vp]),
    "rdg_basis_reg_ws_bytes": (C.c_size_t, [C.c_int32]),
    "rdg_basis_reg": (C.c_int, [C.c_int32] * 4 + [C.POINTER(C.c_float), _vp, _vp, _vp, _vp, _vp]),
    "rdg_pearson_ws_bytes": (C.c_size_t, [C.c_int32]),
    "rdg_pearson_depth_forward": (C.c_int, [C.c_int32] * 5 + [_vp] * 5 + [C.c_float, C.c_float, _vp, _vp, _vp]),
    "rdg_pearson_depth_backward": (C.c_int, [C.c_int32] * 5 + [_vp] * 9),
    "rdg_knn_points_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_vp] * 6),
    "rdg_knn_points_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_vp] * 7),
    "rdg_knn_gather_forward": (C.c_int, [C.c_int64, C.c_int32] + [_vp] * 4),
    "rdg_knn_gather_backward": (C.c_int, [C.c_int64, C.c_int32, C.c_int64] + [_vp] * 4),
    "rdg_adam_step": (C.c_int, [C.c_int64, _vp, _vp, _vp, _vp, C.c_float, C.c_double, C.c_double, C.c_float,
                                C.c_int32, _vp]),
    "rdg_adam_step_rows": (C.c_int, [C.c_int64, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                     C.c_double, C.c_double, C.c_float, C.c_int32, _vp]),
    "rdg_adam_step_multi": (C.c_int, [C.c_int32, C.POINTER(RdgAdamSeg), C.c_double, C.c_double, C.c_float, C.c_int32,
                                      _vp]),
    "rdg_loss_ws_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rdg_photometric_loss_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    "rdg_photometric_loss_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.c_float, _vp, _vp, _vp,
                                                _vp]),
    "rdg_activate_forward": (C.c_int, [C.c_int32, C.c_int32] + [_vp] * 14),
    "rdg_activate_backward": (C.c_int, [C.c_int32, C.c_int32] + [_vp] * 15),
    "rdg_pose_view_forward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "rdg_pose_view_backward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdg_pose_views_forward": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "rdg_pose_views_backward": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp, _vp, _vp]),
    "rdg_mlp_ws_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rdg_mlp_forward": (C.c_int, [C.c_int32] * 5 + [_vp] * 14),
    "rdg_mlp_backward": (C.c_int, [C.c_int32] * 5 + [_vp] * 18),
    "rdg_timing_enable": (C.c_int, [C.c_int32]),
    "rdg_timing_select": (C.c_int, [C.c_uint32]),
    "rdg_timing_reset": (C.c_int, []),
    "rdg_stage_time_ms": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    """Load (once) and return the ctypes handle.  Raises RuntimeError if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"rodygs_amd: HIP extension not built ({LIB_PATH} missing). Run `python -c 'import "
                f"__graft_entry__ as g; g.build()'` or `make -C rodygs_amd/csrc`. There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if h.rdg_abi_version() != ABI_VERSION:
            raise RuntimeError("rodygs_amd: librodygs_hip.so ABI version mismatch -- rebuild")
        if h.rdg_settings_bytes() != C.sizeof(RdgRasterSettings):
            raise RuntimeError("rodygs_amd: RdgRasterSettings of the library and of this binding differ -- rebuild")
        _lib = h
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().rdg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg}")


def ptr(t):
    """Device pointer of a tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


_RAW_STREAM = None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  Called once per library launch: the raw-stream query
    (what torch's own compiled code uses) costs ~1 us, torch.cuda.current_stream() builds a Stream object through several
    Python layers for ~11 us -- 0.08 ms per train step at the sizes where the step is host-bound."""
    global _RAW_STREAM
    import torch
    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _RAW_STREAM:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def timing_enable(on: bool = True, stages=None):
    """Bracket stages with hipEvents on the launch stream.  stages: iterable of stage names (default: all).  Every
    timed stage costs ~10 us of stream gap, so bench.py times only the dominant kernel inside its timed region."""
    mask = 0xFFFFFFFF
    if stages is not None:
        mask = 0
        for name in stages:
            mask |= 1 << STAGES[name]
    lib().rdg_timing_select(mask)
    lib().rdg_timing_enable(1 if on else 0)


def timing_reset():
    lib().rdg_timing_reset()


def stage_times():
    """{stage: (total_ms, launches)} accumulated since the last reset (synchronises on the recorded events)."""
    out = {}
    for name, sid in STAGES.items():
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        lib().rdg_stage_time_ms(sid, C.byref(ms), C.byref(n))
        out[name] = (ms.value, n.value)
    return out
