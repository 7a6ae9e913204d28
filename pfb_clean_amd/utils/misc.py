"""
The three pfb/utils/misc.py helpers that sit on the hot path.

    norm_diff(x, xp)                      misc.py:1316-1351
    l1reweight_func(psiH, outvar, ...)    misc.py:1070-1080
"""
import math

import numpy as np
import torch

from .. import _lib, _dev


def norm_diff_sums(x, xp):
    """Device fp64 pair (sum (x-xp)^2, sum x^2) -- a 2-element GPU tensor view that is
    overwritten by the next reduction on this thread/stream."""
    lib = _lib.load()
    ws, out = _dev.scratch()
    _lib.check(lib.pfb_norm_diff_sums(_dev.code(x.dtype), _dev.ptr(x), _dev.ptr(xp), x.numel(),
                                      _dev.ptr(out), _dev.ptr(ws), _dev.stream()))
    return out[:2]


def norm_diff(x, xp):
    """sqrt(sum((x-xp)^2) / (1e-12 + sum(x^2))) with fp64 accumulation
    (misc.py:1326-1351); 2-D or 3-D input only, like the reference."""
    if x.ndim not in (2, 3):
        raise ValueError("norm_diff is only implemented for 2D or 3D arrays")
    xd, xpd = _dev.to_dev(x), _dev.to_dev(xp)
    if xd.dtype != xpd.dtype or xd.shape != xpd.shape:
        raise ValueError("norm_diff: x and xp must have the same shape and dtype")
    num, den = norm_diff_sums(xd.contiguous(), xpd.contiguous()).tolist()
    return math.sqrt(num / (1e-12 + den))


def l1reweight_func(psiH, outvar, rmsfactor, rms_comps, model, alpha=4, group=None):
    """misc.py:1070-1080: weights (1+rmsfactor)/(1+(|sum_band psiH(model)|/rms_comps)^alpha).
    `psiH` is the analysis operator (Psi.dot) at the call site (spotless.py:243-245).
    The band-sum / power / divide run as torch device ops on the (nbasis, Nymax, Nxmax)
    plane -- executed once per reweighting, not per iteration.  group (extension): bands
    sharded over a torch.distributed group -> the band sum is all-reduced (GPU tensors)."""
    psiH(model, outvar)
    if group is not None:
        import torch.distributed as dist
        plane = torch.sum(outvar, dim=0)
        dist.all_reduce(plane, op=dist.ReduceOp.SUM, group=None if group is True else group)
        rc = rms_comps if not _dev.is_numpy(rms_comps) else torch.as_tensor(rms_comps, device=outvar.device)
        return (1 + rmsfactor) / (1 + torch.abs(plane) ** alpha / rc ** alpha)
    if _dev.is_numpy(outvar):
        mcomps = np.abs(np.sum(outvar, axis=0))
        return (1 + rmsfactor) / (1 + mcomps ** alpha / rms_comps ** alpha)
    mcomps = torch.abs(torch.sum(outvar, dim=0))
    rc = rms_comps if not _dev.is_numpy(rms_comps) else torch.as_tensor(rms_comps, device=outvar.device)
    return (1 + rmsfactor) / (1 + mcomps ** alpha / rc ** alpha)
