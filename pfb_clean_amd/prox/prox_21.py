"""
prox_21 (band-l2-NORM variant) -- pfb/prox/prox_21.py:5-20.  The live spotless worker
passes it to primal_dual_optimised, which never calls it (primal_dual.py:98 `prox` is
unused); kept for call-site compatibility as a thin device expression (torch ops on the
GPU tensor, not a hot path).
"""
import torch

from .. import _dev


def prox_21(v, sigma, weight=None, axis=0):
    vd = _dev.to_dev(v)
    wd = _dev.to_dev(weight, vd.dtype) if weight is not None else 1.0
    l2_norm = torch.linalg.vector_norm(vd, dim=axis)
    l2_soft = torch.clamp(l2_norm - sigma * wd, min=0.0)
    ratio = torch.where(l2_norm != 0, l2_soft / torch.where(l2_norm != 0, l2_norm, torch.ones_like(l2_norm)),
                        torch.zeros_like(l2_norm))
    out = vd * ratio.unsqueeze(axis)
    return out.cpu().numpy() if _dev.is_numpy(v) else out
