"""pfb.wavelets on MI355X: the stand-alone multi-level transforms with the reference's argument lists

    dwt2d (image, coeffs, cbuff, cbuffT, ix, iy, sx, sy, dec_lo, dec_hi, nlevel)         wavelets.py:175-213
    idwt2d(coeffs, image, alpha, cbuff, cbuffT, ix, iy, sx, sy, spx, spy, rec_lo, rec_hi, nlevel)   :261-315

plus the size helpers and the filter tables.  Both write their output argument in place (`coeffs` of dwt2d,
`image` of idwt2d) and leave the input alone, like the reference; the scratch buffers the numba kernels need
(`cbuff`, `cbuffT`, `alpha`) are accepted and ignored -- a level is ONE fused HIP kernel here (csrc/wavelet.hip) and
the inverse never modifies its input.  The packed, transposed coefficient layout is the reference's: `coeffs` is
(Ntoty, Ntotx), the level-k block sits at [iy[k][1] - 2 sy[k] : iy[k][1], ix[k][1] - 2 sx[k] : ix[k][1]], coarser
levels overwrite the approximation quadrant of finer ones.  The caller's bookkeeping (ix, iy, sx, sy, spx, spy) is
CHECKED against the sizes the filter length implies (psi.py:60-94) instead of being trusted: a mismatch raises
ValueError rather than writing somewhere else than the reference would.

The transforms run through a single-band, single-basis plan of the same kernels operators.psi.Psi uses
(pfb_psi_dot / pfb_psi_hdot), cached per (shape, filter, level count, dtype); numpy arguments are staged through the
GPU, torch-ROCm tensors stay resident.
"""
import ctypes as C
import threading

import numpy as np

from .filters import filter_bank, dwt_max_level  # noqa: F401


def coeff_size(nsignal, nfilter):
    """pfb/wavelets/wavelets.py:21-22"""
    return (nsignal + nfilter - 1) // 2


def signal_size(ncoeff, nfilter):
    """pfb/wavelets/wavelets.py:26-27"""
    return 2 * ncoeff - nfilter + 2


def level_sizes(nx, ny, nfilter, nlevel):
    """Per-level coefficient counts (sx, sy), signal sizes (spx, spy), packing ranges (ix, iy) and the packed
    extents (Ntotx, Ntoty) of an nlevel transform of an (nx, ny) image with an nfilter-tap filter pair
    (psi.py:60-94): what the reference's callers pass to dwt2d / idwt2d."""
    sx, sy, spx, spy = [], [], [], []
    Nx, Ny = int(nx), int(ny)
    for _ in range(nlevel):
        cx, cy = coeff_size(Nx, nfilter), coeff_size(Ny, nfilter)
        sx.append(cx)
        sy.append(cy)
        spx.append(signal_size(cx, nfilter))
        spy.append(signal_size(cy, nfilter))
        Nx, Ny = cx + cx % 2, cy + cy % 2
    ix, iy = {}, {}
    hx, hy = 2 * sx[-1], 2 * sy[-1]
    ix[nlevel - 1], iy[nlevel - 1] = (sx[-1], hx), (sy[-1], hy)
    for k in range(nlevel - 2, -1, -1):
        ix[k], iy[k] = (hx, hx + sx[k]), (hy, hy + sy[k])
        hx, hy = hx + sx[k], hy + sy[k]
    return tuple(sx), tuple(sy), tuple(spx), tuple(spy), ix, iy, hx, hy


_plans = {}
_plans_lock = threading.Lock()


def _single_basis_plan(nx, ny, f_lo, f_hi, analysis, nlevel, dtype):
    """pfb_psi plan for ONE band and ONE basis whose four filters follow from the pair handed in (orthogonal banks:
    the synthesis pair is the analysis pair reversed, psi.py:38-41)."""
    from .. import _lib, _dev
    lo = np.ascontiguousarray(f_lo, dtype=np.float64)
    hi = np.ascontiguousarray(f_hi, dtype=np.float64)
    F = lo.size
    if hi.size != F or F % 2 or not (2 <= F <= 18):
        raise ValueError(f"filter pair of lengths ({lo.size}, {hi.size}): need two even-length filters of 2..18 taps")
    key = (int(nx), int(ny), int(nlevel), dtype, bool(analysis), lo.tobytes(), hi.tobytes())
    with _plans_lock:
        hit = _plans.get(key)
        if hit is not None:
            return hit
        _dev.require_device()
        lib = _lib.load()
        filt = np.zeros((1, 4, 18), dtype=np.float64)
        if analysis:
            filt[0, 0, :F], filt[0, 1, :F], filt[0, 2, :F], filt[0, 3, :F] = lo, hi, lo[::-1], hi[::-1]
        else:
            filt[0, 0, :F], filt[0, 1, :F], filt[0, 2, :F], filt[0, 3, :F] = lo[::-1], hi[::-1], lo, hi
        ks = (C.c_int * 1)(F // 2)
        h = C.c_void_p()
        _lib.check(lib.pfb_psi_plan_create(1, int(nx), int(ny), 1, ks, filt.ctypes.data_as(C.POINTER(C.c_double)),
                                           int(nlevel), _dev.code(dtype), C.byref(h)))
        nym, nxm = C.c_int(), C.c_int()
        _lib.check(lib.pfb_psi_plan_dims(h, C.byref(nym), C.byref(nxm)))
        if len(_plans) >= 32:                    # a handful of shapes is the use case; never grow without bound
            _, (old, _, _) = _plans.popitem()
            lib.pfb_psi_plan_destroy(old)
        _plans[key] = (h, nym.value, nxm.value)
        return _plans[key]


def _check_bookkeeping(nx, ny, F, nlevel, coeffs_shape, ix, iy, sx, sy, spx=None, spy=None):
    wsx, wsy, wspx, wspy, wix, wiy, ntx, nty = level_sizes(nx, ny, F, nlevel)
    if tuple(int(v) for v in sx) != wsx or tuple(int(v) for v in sy) != wsy:
        raise ValueError(f"sx / sy {tuple(sx)} / {tuple(sy)} do not belong to a {nlevel}-level transform of a "
                         f"({nx}, {ny}) image with {F} taps (expected {wsx} / {wsy})")
    for k in range(nlevel):
        if int(ix[k][1]) != wix[k][1] or int(iy[k][1]) != wiy[k][1]:
            raise ValueError(f"packing ranges ix[{k}] / iy[{k}] = {tuple(ix[k])} / {tuple(iy[k])}, expected "
                             f"{wix[k]} / {wiy[k]}")
    if spx is not None and (tuple(int(v) for v in spx) != wspx or tuple(int(v) for v in spy) != wspy):
        raise ValueError(f"spx / spy {tuple(spx)} / {tuple(spy)}, expected {wspx} / {wspy}")
    if tuple(coeffs_shape) != (nty, ntx):
        raise ValueError(f"coeffs has shape {tuple(coeffs_shape)}, the packed layout is ({nty}, {ntx})")


def dwt2d(image, coeffs, cbuff, cbuffT, ix, iy, sx, sy, dec_lo, dec_hi, nlevel):
    """Multi-level 2-D image -> coeffs (wavelets.py:175-213); `coeffs` (Ntoty, Ntotx) is written in place (cells of
    the packed layout that belong to no level block are left untouched, as in the reference) and returned."""
    import torch
    from .. import _lib, _dev
    if image.ndim != 2 or coeffs.ndim != 2:
        raise ValueError("dwt2d expects a 2-D image and a 2-D coefficient array")
    nx, ny = (int(v) for v in image.shape)
    F = int(np.asarray(dec_lo).size)
    _check_bookkeeping(nx, ny, F, int(nlevel), coeffs.shape, ix, iy, sx, sy)
    xd = _dev.to_dev(image).contiguous()
    h, nym, nxm = _single_basis_plan(nx, ny, dec_lo, dec_hi, True, nlevel, xd.dtype)
    direct = isinstance(coeffs, torch.Tensor) and coeffs.is_cuda and coeffs.is_contiguous() and coeffs.dtype == xd.dtype
    cd = coeffs if direct else _dev.to_dev(coeffs, xd.dtype).contiguous()
    _lib.check(_lib.load().pfb_psi_dot(h, _dev.ptr(xd), _dev.ptr(cd), _dev.stream()))
    if not direct:
        if _dev.is_numpy(coeffs):
            coeffs[...] = cd.cpu().numpy()
        else:
            coeffs.copy_(cd)
    return coeffs


def idwt2d(coeffs, image, alpha, cbuff, cbuffT, ix, iy, sx, sy, spx, spy, rec_lo, rec_hi, nlevel):
    """Multi-level 2-D coeffs -> image (wavelets.py:261-315); `image` (nx, ny) is overwritten and returned, `coeffs`
    is not modified (the reference copies it into `alpha` for that; here nothing writes to it)."""
    import torch
    from .. import _lib, _dev
    if image.ndim != 2 or coeffs.ndim != 2:
        raise ValueError("idwt2d expects a 2-D coefficient array and a 2-D image")
    nx, ny = (int(v) for v in image.shape)
    F = int(np.asarray(rec_lo).size)
    _check_bookkeeping(nx, ny, F, int(nlevel), coeffs.shape, ix, iy, sx, sy, spx, spy)
    cd = _dev.to_dev(coeffs).contiguous()
    h, nym, nxm = _single_basis_plan(nx, ny, rec_lo, rec_hi, False, nlevel, cd.dtype)
    direct = isinstance(image, torch.Tensor) and image.is_cuda and image.is_contiguous() and image.dtype == cd.dtype
    xd = image if direct else torch.empty((nx, ny), dtype=cd.dtype, device=cd.device)
    _lib.check(_lib.load().pfb_psi_hdot(h, _dev.ptr(cd), _dev.ptr(xd), _dev.stream()))
    if not direct:
        if _dev.is_numpy(image):
            image[...] = xd.cpu().numpy()
        else:
            image.copy_(xd)
    return image


def clear_plans():
    """Destroy the cached single-basis plans (tests; a long-lived process that cycles through many shapes)."""
    from .. import _lib
    with _plans_lock:
        if _plans:
            import torch
            torch.cuda.synchronize()
            lib = _lib.load()
            for h, _, _ in _plans.values():
                lib.pfb_psi_plan_destroy(h)
            _plans.clear()
