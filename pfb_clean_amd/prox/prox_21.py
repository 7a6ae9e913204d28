"""
The band-l2-NORM l21 operators on MI355X -- drop-in for pfb/prox/prox_21.py (the "m" variants of prox_21m.py threshold
|sum over bands|, these the Euclidean norm over bands):

    prox_21(v, sigma, weight=None, axis=0)                       prox_21.py:5-20    (array form, returns a new array)
    prox_21_numba(v, result, lam, sigma=1.0, weight=None)        prox_21.py:23-48   (writes `result`)
    dual_update(v, x, psiH, lam, sigma=1.0, weight=1.0)          prox_21.py:51-58   (array form)
    dual_update_numba(vp, v, lam, sigma=1.0, weight=None)        prox_21.py:62-88   (in place on v)

v, vp, result: (nband, nbasis, ...) with any trailing coefficient shape (the reference flattens it to ntot); weight:
v.shape[1:].  Elementwise HIP kernels with the band loop inside (pfb_prox_21 / pfb_dual_update_l2, csrc/wavelet.hip).
The live spotless worker hands prox_21 to primal_dual_optimised, which never calls it (primal_dual.py:98): these are
not on the hot path, they complete the module.
"""
import numpy as np
import torch

from .. import _lib, _dev
from .prox_21m import _prep, _writeback


def prox_21_numba(v, result, lam, sigma=1.0, weight=None):
    lib = _lib.load()
    vd, wd, nband, nper = _prep(v, weight)
    direct = isinstance(result, torch.Tensor) and result.is_cuda and result.is_contiguous() \
        and result.dtype == vd.dtype and result.shape == vd.shape
    rd = result if direct else torch.empty_like(vd)
    _lib.check(lib.pfb_prox_21(_dev.code(vd.dtype), _dev.ptr(vd), _dev.ptr(rd), _dev.ptr(wd), float(lam), float(sigma),
                               nband, nper, _dev.stream()))
    return _writeback(result, rd)


def prox_21(v, sigma, weight=None, axis=0):
    if axis != 0:
        raise ValueError("band axis must be 0")
    vd = _dev.to_dev(v)
    if weight is None:
        raise ValueError("weight is required")            # the reference multiplies by it unconditionally (:15)
    w = torch.as_tensor(weight, dtype=vd.dtype, device=vd.device).expand(vd.shape[1:]).contiguous() \
        if not isinstance(weight, (np.ndarray, torch.Tensor)) else weight
    res = torch.empty_like(vd)
    prox_21_numba(vd, res, sigma, sigma=1.0, weight=w)    # v max(||v|| - sigma w, 0) / ||v||
    return res.cpu().numpy() if _dev.is_numpy(v) else res


def dual_update_numba(vp, v, lam, sigma=1.0, weight=None):
    lib = _lib.load()
    direct = isinstance(v, torch.Tensor) and v.is_cuda and v.is_contiguous()
    vd, wd, nband, nper = _prep(v, weight)
    vpd = _dev.to_dev(vp, vd.dtype).contiguous()
    if vpd.shape != vd.shape:
        raise ValueError("vp and v must have the same shape")
    if not direct and isinstance(v, torch.Tensor):
        vd = vd.clone()
    _lib.check(lib.pfb_dual_update_l2(_dev.code(vd.dtype), _dev.ptr(vpd), _dev.ptr(vd), _dev.ptr(wd), float(lam),
                                      float(sigma), nband, nper, _dev.stream()))
    return _writeback(v, vd)


def dual_update(v, x, psiH, lam, sigma=1.0, weight=1.0):
    vd = _dev.to_dev(v)
    vout = torch.zeros_like(vd)
    psiH(_dev.to_dev(x), vout)
    w = torch.as_tensor(weight, dtype=vd.dtype, device=vd.device).expand(vd.shape[1:]).contiguous() \
        if not isinstance(weight, (np.ndarray, torch.Tensor)) else weight
    dual_update_numba(vd, vout, lam, sigma=sigma, weight=w)
    return vout.cpu().numpy() if _dev.is_numpy(v) else vout
