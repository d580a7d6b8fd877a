"""``from simple_knn._C import distCUDA2`` (reference: /root/reference/src/model/rodygs_static.py:17,130-133)."""
from rodygs_amd.knn import distCUDA2  # noqa: F401
