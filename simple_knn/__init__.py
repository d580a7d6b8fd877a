"""Import-name shim for ``simple_knn`` (reference: /root/reference/src/model/rodygs_static.py:17)."""
