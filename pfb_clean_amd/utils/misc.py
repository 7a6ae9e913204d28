"""
The three pfb/utils/misc.py helpers that sit on the hot path.

    norm_diff(x, xp)                      misc.py:1316-1351
    l1reweight_func(psiH, outvar, ...)    misc.py:1070-1080
    dds2cubes(dds, nband, ...)            misc.py:664-739   (cube assembly, device resident)
    freqmul(A, x), setup_parametrisation  misc.py:1366-1423 (band coupling of the fwdbwd parametrisations)
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


def dds2cubes(dds, nband, apparent=False, dual=True, modelname='MODEL'):
    """pfb/utils/misc.py:664-739 -- assembles the image cubes the solvers work on, here as
    DEVICE-RESIDENT tensors (no dask graph): returns
    `(dirty, model, residual, psf, psfhat, mean_beam, wsums, dual)` with the reference's
    normalisation (dirty/residual beam-weighted unless `apparent` and, like psf/psfhat, divided by
    the TOTAL wsum; mean_beam = sum(beam*wsum)/wsums[band]; datasets sharing a band are summed,
    model/dual come from the band's last dataset; absent variables give None).  `dds` is a list of
    the reference's per-band datasets or of anything exposing DIRTY, BEAM, WSUM [, RESIDUAL, PSF,
    PSFHAT, DUAL, <modelname>] with `.values` (numpy or tensors), `bandid`, `name in ds` and
    `ds[name]`.  Reading the `.dds` zarr store itself stays with xarray/zarr (not in this image)."""
    dev = _dev.require_device()
    d0 = dds[0]
    first = _dev.to_dev(d0.DIRTY.values)
    rt = first.dtype
    ct = torch.complex64 if rt == torch.float32 else torch.complex128
    nx, ny = first.shape

    def get(ds, name, dt=rt):
        return _dev.to_dev(ds[name].values, dt)

    dirty = torch.zeros((nband, nx, ny), dtype=rt, device=dev)
    model = torch.zeros_like(dirty)
    residual = torch.zeros_like(dirty) if 'RESIDUAL' in d0 else None
    wsums = torch.zeros(nband, dtype=rt, device=dev)
    psf = psfhat = None
    if 'PSF' in d0:
        psf = torch.zeros((nband,) + tuple(d0['PSF'].values.shape), dtype=rt, device=dev)
        psfhat = torch.zeros((nband,) + tuple(d0['PSFHAT'].values.shape), dtype=ct, device=dev)
    mean_beam = torch.zeros_like(dirty)
    dualc = (torch.zeros((nband,) + tuple(d0['DUAL'].values.shape), dtype=rt, device=dev)
             if (dual and 'DUAL' in d0) else None)
    for ds in dds:
        b = ds.bandid
        beam = get(ds, 'BEAM')
        w = float(ds['WSUM'].values[0])
        dirty[b] += get(ds, 'DIRTY') if apparent else get(ds, 'DIRTY') * beam
        if 'RESIDUAL' in ds:
            residual[b] += get(ds, 'RESIDUAL') if apparent else get(ds, 'RESIDUAL') * beam
        if 'PSF' in ds:
            psf[b] += get(ds, 'PSF')
            psfhat[b] += get(ds, 'PSFHAT', ct)
        if modelname in ds:
            model[b] = get(ds, modelname)
        if dual and 'DUAL' in ds:
            dualc[b] = get(ds, 'DUAL')
        mean_beam[b] += beam * w
        wsums[b] += w
    wsum = wsums.sum()
    dirty /= wsum
    if residual is not None:
        residual /= wsum
    if psf is not None:
        psf /= wsum
        psfhat /= wsum
    nz = wsums != 0
    mean_beam[nz] /= wsums[nz][:, None, None]
    return dirty, model, residual, psf, psfhat, mean_beam, wsums, dualc


def _freqmul(A, x, pre=None, post=None):
    lib = _lib.load()
    xd = _dev.to_dev(x).contiguous()
    Ad = _dev.to_dev(A, xd.dtype).contiguous()
    out = torch.empty_like(xd)
    nband = xd.shape[0]
    if tuple(Ad.shape) != (nband, nband):
        raise ValueError(f"A must be ({nband},{nband})")
    _lib.check(lib.pfb_freqmul(_dev.code(xd.dtype), _dev.ptr(Ad), _dev.ptr(xd), _dev.ptr(out), nband,
                               xd[0].numel(), _dev.ptr(pre), _dev.ptr(post), _dev.stream()))
    return out


def freqmul(A, x):
    """misc.py:1366-1375: out[k] = sum_l A[k, l] x[l] for an (nband, nx, ny) cube, on the GPU."""
    out = _freqmul(A, x)
    return out.cpu().numpy() if _dev.is_numpy(x) else out


def setup_parametrisation(mode='id', minval=1e-5, sigma=1.0, freq=None, lscale=1.0):
    """misc.py:1378-1423: x = f(s) with a squared-exponential band covariance K = L L^T.  Returns
    (func, finv, dfunc, dhfunc) working on GPU tensors or numpy cubes; the nband x nband factor is
    built on the host (numpy Cholesky), every cube-sized operation is one pfb_freqmul launch (the
    exp / product factors of mode='exp' fused into it).  finv applies L^-1 through the same kernel
    (the reference's scipy.solve_triangular on the cube)."""
    nu = np.asarray(freq, dtype=np.float64) / np.mean(freq)
    nband = nu.size
    K = sigma ** 2 * np.exp(-(nu[:, None] - nu[None, :]) ** 2 / (2 * lscale ** 2))
    L = np.linalg.cholesky(K + 1e-10 * np.eye(nband))
    LH = np.ascontiguousarray(L.T)
    Linv = np.linalg.solve(L, np.eye(nband))

    def back(t, like):
        return t.cpu().numpy() if _dev.is_numpy(like) else t

    if mode == 'id':
        def func(x):
            return back(_freqmul(L, x), x)

        def finv(x):
            return back(_freqmul(Linv, x), x)

        def dfunc(x0, v):
            return back(_freqmul(L, v), v)

        def dhfunc(x0, v):
            return back(_freqmul(LH, v), v)
    elif mode == 'exp':
        def func(x):
            return back(torch.exp(_freqmul(L, x)), x)

        def finv(x):
            t = _freqmul(Linv, x)
            return back(torch.log(torch.clamp(torch.abs(t), min=minval)), x)

        def dfunc(x0, v):
            e = torch.exp(_freqmul(L, x0))
            return back(_freqmul(L, v, post=e), v)                 # exp(L x0) * (L v)

        def dhfunc(x0, v):
            e = torch.exp(_freqmul(L, x0))
            return back(_freqmul(LH, v, pre=e), v)                 # L^T (v * exp(L x0))
    else:
        raise ValueError(f"Unknown mode - {mode}")
    return func, finv, dfunc, dhfunc
