"""Memory layout of the Gaussian cloud: Z-curve (Morton) order of the canonical positions.

The rasterizer is indifferent to the order of its input rows (only ties in depth are broken by the row index), but the
kernels are not: with spatially coherent rows a wave's 64 tile instances fall into a handful of tiles (the count pass
of the binning issues one atomic per DISTINCT tile of a wave, csrc/rdg_binning.hip), a tile's splat list gathers
64-byte records that sit next to each other in HBM, and the scatter of the binning writes runs instead of single
slots.  The reference keeps whatever order its point cloud and its densification appends produce
(/root/reference/src/trainer/rodygs_static.py:218-299); a trainer built on this package orders the cloud once at
initialisation (and after a densification, where the row gather happens anyway)."""
from __future__ import annotations

import torch


def _spread3(v: torch.Tensor) -> torch.Tensor:
    """Insert two zero bits between the low 21 bits of every int64 element."""
    v = v & 0x1FFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def morton_codes(xyz: torch.Tensor, bits: int = 16) -> torch.Tensor:
    """63-bit-safe Z-curve index of every row of xyz [P,3] inside the cloud's bounding box (`bits` <= 21 per axis)."""
    if xyz.numel() == 0:
        return torch.zeros(0, dtype=torch.int64, device=xyz.device)
    if xyz.is_cuda and xyz.dtype == torch.float32:
        # one HIP launch (rdg_morton_codes: the same arithmetic in double) instead of ~35 framework kernels
        from . import _lib
        x = xyz.detach().contiguous()
        lo_hi = torch.cat([x.amin(dim=0), x.amax(dim=0)]).contiguous()
        codes = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().rdg_morton_codes(x.shape[0], _lib.ptr(x), _lib.ptr(lo_hi), int(bits), _lib.ptr(codes),
                                                   _lib.stream_ptr()), "rdg_morton_codes")
        return codes
    x = xyz.detach().to(torch.float64)
    lo, hi = x.min(dim=0).values, x.max(dim=0).values
    q = ((x - lo) / (hi - lo).clamp_min(1e-30) * ((1 << bits) - 1)).round().to(torch.int64)
    return _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)


def morton_order(xyz: torch.Tensor, bits: int = 16) -> torch.Tensor:
    """Permutation (int64 [P]) that sorts the rows of xyz along the Z curve; stable, so equal cells keep their order.
    GPU tensors go through the library's LSD radix sort on the 3 * bits code bits that exist (six 8-bit passes at bits = 16,
    where the framework's sort walks all 64 bits of an int64: 1.8 -> 0.4 ms at 1.1 M rows inside a densification)."""
    codes = morton_codes(xyz, bits)
    if codes.is_cuda and codes.numel() > 0 and codes.numel() < 2 ** 31:
        from .rigidity import _sort_by_key
        return _sort_by_key(codes, 3 * bits)[1]
    return torch.sort(codes, stable=True).indices
