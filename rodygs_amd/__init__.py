"""rodygs_amd -- MI355X-native hot path of RoDyGS: differentiable Gaussian rasterizer, per-Gaussian time
deformation and simple_knn, as hand-written gfx950 HIP kernels behind a C-ABI (include/rodygs_hip.h).

Only the hot path lives here; the reference's trainer, data, CLI and evaluator are out of scope.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians)  # noqa: F401
from .knn import distCUDA2  # noqa: F401
from .deform import (MLPBasisNetwork, TimestepEmbedder, MLPMotionBasis, gaussian_deformation,  # noqa: F401
                     DeformationField)
from .render import render, render_model  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "distCUDA2",
           "MLPBasisNetwork", "TimestepEmbedder", "MLPMotionBasis", "gaussian_deformation", "DeformationField",
           "render", "render_model"]
