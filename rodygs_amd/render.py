"""``render()`` glue with the reference's signature and return dict
(/root/reference/src/trainer/renderer.py:17-114; clones at src/model/rodygs_static.py:184-296 and
src/evaluator/eval.py:84-191), calling the MI355X-native rasterizer."""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def render(xyz, active_sh_degree, opacity, scaling, rotation, features, viewpoint_camera, bg_color: torch.Tensor,
           scaling_modifier=1, override_color=None, enable_sh_grad=False, enable_cov_grad=False):
    """Render the scene.  ``viewpoint_camera`` needs FoVx, FoVy, image_height, image_width, projection_matrix
    and world_view_transform (the FixedCameraTorch surface, /root/reference/src/data/utils.py:105-170)."""
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)

    # NOTE: the reference passes enable_cov_grad=enable_sh_grad and enable_sh_grad=enable_cov_grad
    # (renderer.py:61-62, names swapped; harmless there because callers pass equal flags).  Reproduced as is.
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        projmatrix=viewpoint_camera.projection_matrix.transpose(0, 1),  # glm storage
        sh_degree=active_sh_degree,
        prefiltered=False,
        debug=False,
        enable_cov_grad=enable_sh_grad,
        enable_sh_grad=enable_cov_grad,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    shs = None
    colors_precomp = None
    if override_color is None:
        shs = features
    else:
        colors_precomp = override_color

    rendered_image, rendered_depth, rendered_normal, rendered_alpha, radii, extra = rasterizer(
        means3D=xyz,
        means2D=screenspace_points,
        shs=shs,
        colors_precomp=colors_precomp,
        opacities=opacity,
        scales=scaling,
        rotations=rotation,
        cov3Ds_precomp=None,
        viewmatrix=viewpoint_camera.world_view_transform.transpose(0, 1),  # glm storage
    )
    return {
        "rendered_image": rendered_image,
        "rendered_depth": rendered_depth,
        "rendered_normal": rendered_normal,
        "rendered_alpha": rendered_alpha,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "extra": extra,
    }
