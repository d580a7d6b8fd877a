"""Import-name shim: ``from diff_gauss_pose import GaussianRasterizationSettings, GaussianRasterizer`` is what
the reference writes (/root/reference/src/trainer/renderer.py:14, src/model/rodygs_static.py:19,
src/evaluator/eval.py:25).  Everything resolves to the MI355X-native implementation in ``rodygs_amd``."""
from rodygs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                   rasterize_gaussians)
