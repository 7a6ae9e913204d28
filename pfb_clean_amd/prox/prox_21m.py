"""
l21 ("m" = band-SUM variant) proximal operator and the fused dual update on MI355X --
drop-in for pfb/prox/prox_21m.py.

    prox_21m_numba(v, result, lam, sigma=1.0, weight=None)      prox_21m.py:31-61
    dual_update_numba(vp, v, lam, sigma=1.0, weight=None)       prox_21m.py:76-103  (in place on v)
    prox_21m(v, sigma, weight=1.0, axis=0)                      prox_21m.py:5-27    (array form)
    dual_update(v, x, psiH, lam, sigma=1.0, weight=1.0)         prox_21m.py:64-71   (array form)

v, vp, result: (nband, nbasis, nymax, nxmax); weight: (nbasis, nymax, nxmax).  One
elementwise HIP kernel with the band loop in registers (pfb_dual_update / pfb_prox_21m).
The two array-form helpers are not on the hot path (tests only in the reference) and are
expressed with the same kernels.
"""
import numpy as np
import torch

from .. import _lib, _dev


def _prep(v, weight):
    vd = _dev.to_dev(v).contiguous()
    if vd.ndim < 2:
        raise ValueError("expected (nband, ...) coefficient cube")
    nband = vd.shape[0]
    nper = vd[0].numel()
    if weight is None:
        raise ValueError("weight is required")
    wd = _dev.to_dev(weight, vd.dtype)
    if wd.ndim == 0:
        wd = wd.expand(vd.shape[1:])
    wd = wd.contiguous()
    if wd.numel() != nper:
        raise ValueError("weight must have shape v.shape[1:]")
    return vd, wd, nband, nper


def _writeback(dst, src):
    if dst is src:
        return dst
    if _dev.is_numpy(dst):
        dst[...] = src.cpu().numpy()
    else:
        dst.copy_(src)
    return dst


def prox_21m_numba(v, result, lam, sigma=1.0, weight=None):
    lib = _lib.load()
    vd, wd, nband, nper = _prep(v, weight)
    direct = isinstance(result, torch.Tensor) and result.is_cuda and result.is_contiguous() \
        and result.dtype == vd.dtype and result.shape == vd.shape
    rd = result if direct else torch.empty_like(vd)
    _lib.check(lib.pfb_prox_21m(_dev.code(vd.dtype), _dev.ptr(vd), _dev.ptr(rd), _dev.ptr(wd),
                                float(lam), float(sigma), nband, nper, _dev.stream()))
    return _writeback(result, rd)


def dual_update_numba(vp, v, lam, sigma=1.0, weight=None, vp_out=None, group=None):
    """In place on v.  vp_out (extension): also receives 2*v_new - vp, the next statement
    of primal_dual_optimised (primal_dual.py:137), saving one pass over the cube.
    group (extension): a torch.distributed process group over which the BAND axis is sharded;
    the band sum of prox_21m.py:99 then becomes local sum -> all-reduce of the
    (nbasis, nymax, nxmax) plane (RCCL) -> apply (GPU tensors only)."""
    if group is not None:
        return _dual_update_sharded(vp, v, lam, sigma, weight, vp_out, group)
    lib = _lib.load()
    direct = isinstance(v, torch.Tensor) and v.is_cuda and v.is_contiguous()
    vd, wd, nband, nper = _prep(v, weight)
    vpd = _dev.to_dev(vp, vd.dtype).contiguous()
    if vpd.shape != vd.shape:
        raise ValueError("vp and v must have the same shape")
    if not direct and isinstance(v, torch.Tensor):
        vd = vd.clone()
    _lib.check(lib.pfb_dual_update(_dev.code(vd.dtype), _dev.ptr(vpd), _dev.ptr(vd), _dev.ptr(wd),
                                   float(lam), float(sigma), nband, nper,
                                   _dev.ptr(vp_out) if vp_out is not None else None, _dev.stream()))
    return _writeback(v, vd)


def _dual_update_sharded(vp, v, lam, sigma, weight, vp_out, group):
    """Bands sharded over the ranks of `group`: local band sums -> sum over the ranks -> threshold, chunk by chunk with
    the exchange of one chunk overlapping the kernels of its neighbours (dist.exchange_plane_pipelined: reduce-scatter
    + all-gather per chunk on RCCL)."""
    from ..dist import exchange_plane_pipelined
    lib = _lib.load()
    if not (isinstance(v, torch.Tensor) and v.is_cuda and v.is_contiguous()):
        raise TypeError("band-sharded dual update works on contiguous GPU tensors")
    vd, wd, nband, nper = _prep(v, weight)
    vpd = _dev.to_dev(vp, vd.dtype).contiguous()
    if vp_out is not None and not (vp_out.is_contiguous() and vp_out.shape == vd.shape and vp_out.dtype == vd.dtype):
        raise ValueError("vp_out must be a contiguous tensor shaped like v")
    plane = torch.empty(nper, dtype=vd.dtype, device=vd.device)
    code = _dev.code(vd.dtype)
    es = vd.element_size()

    def at(t, off):
        return None if t is None else t.data_ptr() + off * es

    def bandsum(off, cnt):
        _lib.check(lib.pfb_dual_bandsum_chunk(code, at(vpd, off), at(vd, off), float(sigma), nband, cnt, nper,
                                              at(plane, off), _dev.stream()))

    def apply(off, cnt):
        _lib.check(lib.pfb_dual_apply_chunk(code, at(vpd, off), at(vd, off), at(wd, off), at(plane, off), float(lam),
                                            float(sigma), nband, cnt, nper, at(vp_out, off), _dev.stream()))
    exchange_plane_pipelined(plane, bandsum, apply, group)
    return v


def prox_21m(v, sigma, weight=1.0, axis=0):
    if axis != 0:
        raise ValueError("band axis must be 0")
    vd = _dev.to_dev(v)
    w = torch.as_tensor(weight, dtype=vd.dtype, device=vd.device).expand(vd.shape[1:]).contiguous() \
        if not isinstance(weight, (np.ndarray, torch.Tensor)) else weight
    res = torch.empty_like(vd)
    prox_21m_numba(vd, res, sigma, sigma=1.0, weight=w)
    return res.cpu().numpy() if _dev.is_numpy(v) else res


def dual_update(v, x, psiH, lam, sigma=1.0, weight=1.0):
    vd = _dev.to_dev(v)
    vout = torch.zeros_like(vd)
    psiH(_dev.to_dev(x), vout)
    w = torch.as_tensor(weight, dtype=vd.dtype, device=vd.device).expand(vd.shape[1:]).contiguous() \
        if not isinstance(weight, (np.ndarray, torch.Tensor)) else weight
    dual_update_numba(vd, vout, lam, sigma=sigma, weight=w)
    return vout.cpu().numpy() if _dev.is_numpy(v) else vout
