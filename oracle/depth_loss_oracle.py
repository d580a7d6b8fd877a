"""CPU oracle for the Pearson depth losses (TEST INFRASTRUCTURE, never shipped or imported by the product path).

Restates /root/reference/src/utils/loss_utils.py:100-117 (pearson_depth_loss) and the box loop of
LocalPearsonDepthLoss (/root/reference/src/trainer/losses.py:131-182) in plain torch; pinned by
tests/golden/depth_loss_golden.npz, produced by the imported reference (tests/golden/make_golden.py G8)."""
from __future__ import annotations

import torch


def pearson_depth_loss(pred, gt, eps=1e-6, mask=None):
    a = pred * mask if mask is not None else pred
    b = gt * mask if mask is not None else gt
    a = a - a.mean()
    b = b - b.mean()
    return 1 - ((a / (a.std() + eps)) * (b / (b.std() + eps))).mean()


def local_pearson_depth_loss(pred, gt, rows, cols, box_p, n_corr, mask=None, eps=1e-6):
    """pred, gt [1,H,W]; rows/cols: box corners; mask: optional bool [1,H,W] (already ~motion or motion)."""
    total = torch.zeros((), dtype=pred.dtype)
    for r, c in zip(rows.tolist(), cols.tolist()):
        sl = (slice(None), slice(r, r + box_p), slice(c, c + box_p))
        m = None
        if mask is not None:
            m = mask[sl].reshape(-1)
            if m.sum() == 0:
                continue
        total = total + pearson_depth_loss(pred[sl].reshape(-1), gt[sl].reshape(-1), eps, m)
    return total / n_corr
