"""``pytorch3d.ops`` surface used by RoDyGS (/root/reference/src/trainer/losses.py:235-331)."""
from rodygs_amd.knn import knn_gather, knn_points  # noqa: F401

__all__ = ["knn_points", "knn_gather"]
