"""
CPU restatement of the PSF convolution and the PSF-approximated Hessians.

Test infrastructure (see oracle/__init__.py).  Follows
  pfb/operators/psf.py:11-56      psf_convolve_slice / psf_convolve_cube
  pfb/operators/hessian.py:129-158 _hessian_psf_slice
  pfb/operators/hessian.py:254-281 hessian_psf_cube
  pfb/operators/gridder.py:712-714 + pfb/operators/fft.py:7-9   psfhat producer

ducc0.fft.r2c(inorm=0) == scipy.fft.rfftn (unnormalised forward);
ducc0.fft.c2r(inorm=2, lastsize=L) == scipy.fft.irfftn(s=(..., L)) (1/N backward).
"""
import os
import numpy as np
import scipy.fft as _sfft


def _workers(nthreads):
    if nthreads is None or nthreads <= 0:
        return os.cpu_count() or 1
    return int(nthreads)


def psfhat_from_psf(psf, nthreads=1):
    """gridder.py:712-714: psfhat = r2c(ifftshift(psf), axes=(0,1), inorm=0).
    psf is (nx_psf, ny_psf) or (nband, nx_psf, ny_psf)."""
    axes = (-2, -1)
    return _sfft.rfftn(_sfft.ifftshift(psf, axes=axes), axes=axes,
                       workers=_workers(nthreads))


def psf_convolve_slice(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    """psf.py:11-29.  Returns xout (aliased, like the reference)."""
    nx, ny = x.shape
    xpad[...] = 0.0                                   # psf.py:20
    xpad[0:nx, 0:ny] = x                              # psf.py:21
    xhat[...] = _sfft.rfftn(xpad, axes=(0, 1),        # psf.py:22-23
                            workers=_workers(nthreads))
    xhat *= psfhat                                    # psf.py:24
    xpad[...] = _sfft.irfftn(xhat, s=(xpad.shape[0], lastsize), axes=(0, 1),
                             workers=_workers(nthreads))      # psf.py:25-27
    xout[...] = xpad[0:nx, 0:ny]                      # psf.py:28
    return xout


def psf_convolve_cube(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    """psf.py:32-56 (axes=(1,2), batched over the band axis)."""
    _, nx, ny = x.shape
    xpad[...] = 0.0
    xpad[:, 0:nx, 0:ny] = x
    xhat[...] = _sfft.rfftn(xpad, axes=(1, 2), workers=_workers(nthreads))
    xhat *= psfhat
    xpad[...] = _sfft.irfftn(xhat, s=(xpad.shape[1], lastsize), axes=(1, 2),
                             workers=_workers(nthreads))
    xout[...] = xpad[:, 0:nx, 0:ny]
    return xout


def _hessian_psf_slice(xpad, xhat, xout, psfhat, beam, lastsize, x,
                       nthreads=1, sigmainv=1, wsum=None):
    """hessian.py:129-158.  Note the argument order (psfhat, beam)."""
    if beam is not None:
        psf_convolve_slice(xpad, xhat, xout, psfhat, lastsize, x * beam,
                           nthreads=nthreads)
    else:
        psf_convolve_slice(xpad, xhat, xout, psfhat, lastsize, x,
                           nthreads=nthreads)
    if beam is not None:
        xout *= beam
    if wsum is not None:
        xout /= wsum
    return xout + x * sigmainv


def hessian_psf_cube(xpad, xhat, xout, beam, psfhat, lastsize, x,
                     nthreads=1, sigmainv=1, wsum=None):
    """hessian.py:254-281.  Note the argument order (beam, psfhat)."""
    if beam is not None:
        psf_convolve_cube(xpad, xhat, xout, psfhat, lastsize, x * beam,
                          nthreads=nthreads)
    else:
        psf_convolve_cube(xpad, xhat, xout, psfhat, lastsize, x,
                          nthreads=nthreads)
    if beam is not None:
        xout *= beam
    if wsum is not None:
        xout /= wsum
    return xout + x * sigmainv


class hessian_psf_slice:
    """hessian.py:161-251: the per-band stateful operator.  `ds` is anything with the
    reference dataset's variables (`.values`) -- DIRTY, PSFHAT, PSF, BEAM, WSUM and optionally
    MODEL, DUAL, RESIDUAL -- and `bandid`.  __call__ = _hessian_psf_slice with the wsum given
    to set_wsum (the TOTAL wsum, not the band's).  compute_residual is visibility space
    (wgridder) and outside the hot path."""

    def __init__(self, ds, nbasis, nmax, nthreads, sigmainv, cell=None, do_wgridding=None,
                 epsilon=None, double_accum=None):
        self.nthreads, self.sigmainv = nthreads, sigmainv
        self.lastsize = ds.PSF.shape[-1]
        self.bandid = ds.bandid
        self.dirty = np.ascontiguousarray(ds.DIRTY.values)
        self.psfhat = np.ascontiguousarray(ds.PSFHAT.values)
        self.psf = np.ascontiguousarray(ds.PSF.values)
        self.beam = np.ascontiguousarray(ds.BEAM.values)
        self.wsumb = ds.WSUM.values[0]
        self.model = np.ascontiguousarray(ds.MODEL.values) if 'MODEL' in ds else np.zeros_like(self.dirty)
        if 'DUAL' in ds:
            self.dual = np.ascontiguousarray(ds.DUAL.values)
            assert self.dual.shape == (nbasis, nmax)
        else:
            self.dual = np.zeros((nbasis, nmax), dtype=self.dirty.dtype)
        self.residual = np.ascontiguousarray(ds.RESIDUAL.values) if 'RESIDUAL' in ds else self.dirty.copy()
        self.xout = np.empty(self.dirty.shape, dtype=self.dirty.dtype)
        self.xhat = np.empty(self.psfhat.shape, dtype=self.psfhat.dtype)
        self.xpad = np.empty(self.psf.shape, dtype=self.psf.dtype)

    def __call__(self, x):
        return _hessian_psf_slice(self.xpad, self.xhat, self.xout, self.psfhat, self.beam,
                                  self.lastsize, x, nthreads=self.nthreads, sigmainv=self.sigmainv,
                                  wsum=self.wsum)

    def set_wsum(self, wsum):
        self.wsum = wsum


def make_scratch(psfhat, lastsize, shape, dtype):
    """Allocate (xpad, xhat, xout) the way the workers do
    (pcg.py:270-275, spotless.py:176-180), minus make_noncritical."""
    if psfhat.ndim == 2:
        nx_psf, nyo2 = psfhat.shape
        xpad = np.empty((nx_psf, lastsize), dtype=dtype)
        xhat = np.empty((nx_psf, nyo2), dtype=psfhat.dtype)
    else:
        nb, nx_psf, nyo2 = psfhat.shape
        xpad = np.empty((nb, nx_psf, lastsize), dtype=dtype)
        xhat = np.empty((nb, nx_psf, nyo2), dtype=psfhat.dtype)
    xout = np.empty(shape, dtype=dtype)
    return xpad, xhat, xout


def direct_circular_convolution(psf, x):
    """Definition the FFT path must equal:
    y[i,j] = sum_{i',j'} psf[(P//2+i-i') % P, (Q//2+j-j') % Q] x[i',j'].
    O(N^2); only for tiny known-answer tests."""
    P, Q = psf.shape
    nx, ny = x.shape
    y = np.zeros((nx, ny), dtype=np.result_type(psf, x))
    ii = np.arange(nx)
    jj = np.arange(ny)
    for i in range(nx):
        ri = (P // 2 + i - ii) % P
        for j in range(ny):
            rj = (Q // 2 + j - jj) % Q
            y[i, j] = np.sum(psf[np.ix_(ri, rj)] * x)
    return y
