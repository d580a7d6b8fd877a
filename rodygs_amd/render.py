"""``render()`` glue: the reference's signature and return dictionary
(/root/reference/src/trainer/renderer.py:17-114; the same glue is repeated at src/model/rodygs_static.py:184-296 and
src/evaluator/eval.py:84-191) in front of the MI355X-native rasterizer.

Contract kept from the reference (callers depend on it): the positional / keyword arguments, the eight keys of the
result, ``viewspace_points`` receiving dL/dmean2D in ``.grad`` after backward (read at src/trainer/rodygs.py:322-324),
``visibility_filter = radii > 0``, transposed ("glm storage") camera matrices, and the crossed gradient gates
(renderer.py:61-62 hands ``enable_sh_grad`` to the rasterizer's ``enable_cov_grad`` and the other way round; every
shipped caller passes equal flags)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, RasterState

_IMAGE_KEYS = ("rendered_image", "rendered_depth", "rendered_normal", "rendered_alpha")


def camera_settings(camera, bg: torch.Tensor, sh_degree: int, scale_modifier: float, cov_gate: bool,
                    sh_gate: bool) -> GaussianRasterizationSettings:
    """Rasterizer settings of a camera object with the FixedCameraTorch surface
    (/root/reference/src/data/utils.py:105-170): FoVx, FoVy, image_height, image_width, projection_matrix."""
    return GaussianRasterizationSettings(
        int(camera.image_height), int(camera.image_width), math.tan(0.5 * camera.FoVx), math.tan(0.5 * camera.FoVy), bg,
        scale_modifier, camera.projection_matrix.t(), sh_degree, False, False, cov_gate, sh_gate)


def render(xyz, active_sh_degree, opacity, scaling, rotation, features, viewpoint_camera, bg_color: torch.Tensor,
           scaling_modifier=1, override_color=None, enable_sh_grad=False, enable_cov_grad=False,
           raster_state: Optional[RasterState] = None):
    """One differentiable render of the cloud from ``viewpoint_camera``.  ``raster_state`` (outside the reference
    surface): the caller's ``RasterState``; None = the process default."""
    settings = camera_settings(viewpoint_camera, bg_color, active_sh_degree, scaling_modifier,
                               cov_gate=enable_sh_grad, sh_gate=enable_cov_grad)
    # the 2-D means carry no value into the rasterizer; the tensor exists to collect dL/dmean2D
    mean2d_sink = xyz.new_zeros(xyz.shape).requires_grad_()
    colour = {"colors_precomp": override_color} if override_color is not None else {"shs": features}
    *images, radii, extra = GaussianRasterizer(settings, state=raster_state)(
        means3D=xyz, means2D=mean2d_sink, opacities=opacity, scales=scaling, rotations=rotation,
        viewmatrix=viewpoint_camera.world_view_transform.t(), **colour)
    out = dict(zip(_IMAGE_KEYS, images))
    out.update(viewspace_points=mean2d_sink, visibility_filter=radii > 0, radii=radii, extra=extra)
    return out


def render_model(model, viewpoint_camera, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
                 enable_sh_grad=False, enable_cov_grad=False, translation=0.0, rotation=0.0,
                 raster_state: Optional[RasterState] = None, **kwargs):
    """The MODEL variant of the glue -- ``StaticRoDyGS.render`` (/root/reference/src/model/rodygs_static.py:184-296), the one
    ``PoseOptimizer`` (src/evaluator/eval.py:407-412) and ``ThreeDGSTrainer.train_iteration``
    (src/trainer/rodygs_static.py:375-381) call: the cloud comes from the model's getters (``get_xyz``, ``get_opacity``,
    ``get_scaling``, ``get_rotation``, ``get_features``, ``active_sh_degree``, ``isotropic``), ``translation`` is added to the
    means and ``rotation`` to the (activated) quaternions -- the latter only for an anisotropic model, as the reference
    does (:247-249) --, both default to the number 0.0, and both are handed back under the two extra keys ``translation`` and
    ``rotation`` of the result (:294-295).  ``viewspace_points`` is the reference's non-leaf ``zeros + 0`` with its gradient
    retained (:204-216): ``.grad`` holds dL/dmean2D after backward.  Unknown keyword arguments are accepted and ignored, as
    the reference's ``**kwargs`` does; ``raster_state`` is outside the reference surface."""
    xyz = model.get_xyz
    settings = camera_settings(viewpoint_camera, bg_color, model.active_sh_degree, scaling_modifier,
                               cov_gate=enable_sh_grad, sh_gate=enable_cov_grad)
    mean2d_sink = torch.zeros_like(xyz, requires_grad=True) + 0
    if mean2d_sink.requires_grad:          # (not under torch.no_grad(): the reference wraps the call in try / except)
        mean2d_sink.retain_grad()
    quats = model.get_rotation if getattr(model, "isotropic", False) else model.get_rotation + rotation
    colour = {"colors_precomp": override_color} if override_color is not None else {"shs": model.get_features}
    *images, radii, extra = GaussianRasterizer(settings, state=raster_state)(
        means3D=xyz + translation, means2D=mean2d_sink, opacities=model.get_opacity, scales=model.get_scaling,
        rotations=quats, viewmatrix=viewpoint_camera.world_view_transform.t(), **colour)
    out = dict(zip(_IMAGE_KEYS, images))
    out.update(viewspace_points=mean2d_sink, visibility_filter=radii > 0, radii=radii, extra=extra,
               translation=translation, rotation=rotation)
    return out
