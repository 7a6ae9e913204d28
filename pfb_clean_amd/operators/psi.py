"""
Wavelet dictionary operator psi / psi^H on MI355X -- drop-in for pfb/operators/psi.py:269-310.

    psi = Psi(nband, nx, ny, bases, nlevel, nthreads)
    psi.dot(x, alphao)     analysis : x (nband,nx,ny) -> alphao (nband,nbasis,Nymax,Nxmax)
    psi.hdot(alpha, xo)    synthesis: alpha -> xo (nband,nx,ny), summed over bases

Same attributes (nband, nx, ny, nbasis, nthreads, Nxmax, Nymax), same in-place contract
(outputs may hold garbage on entry, tests/test_psi_operator.py:36-45), same packed
transposed coefficient layout with never-written margin cells, same un-normalised
convention hdot(dot(x)) = nbasis * x.  bases: 'self' | 'db1'..'db9'.

The reference's per-band ThreadPool + numba prange row loops become one fused HIP kernel
per (basis, level) covering all bands (pfb_psi_dot / pfb_psi_hdot, csrc/wavelet.hip).
The reference's psi buffers are float64 only (psi.py:130-133); here the dtype follows
the arrays (float64 or float32).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib, _dev
from ..wavelets import filters as _filters


class Psi(object):
    def __init__(self, nband, nx, ny, bases, nlevel, nthreads=1, dtype=None):
        self.nband, self.nx, self.ny = int(nband), int(nx), int(ny)
        self.bases = list(bases)
        self.nbasis = len(self.bases)
        self.nlevel = int(nlevel)
        self.nthreads = nthreads
        ks = []
        filt = np.zeros((self.nbasis, 4, 18), dtype=np.float64)
        for i, w in enumerate(self.bases):
            if w == 'self':
                ks.append(0)
                continue
            fb = _filters.filter_bank(w)
            max_level = _filters.dwt_max_level(min(nx, ny), w)
            if self.nlevel > max_level:
                raise ValueError(f"The requested decomposition level {nlevel} "
                                 "is not possible")
            K = int(w[2:])
            ks.append(K)
            for q in range(4):
                filt[i, q, :2 * K] = fb[q]
        self._ks = (C.c_int * self.nbasis)(*ks)
        self._filt = filt
        self._plans = {}
        self._lib = _lib.load()
        # dims do not depend on dtype: build the default plan now (also validates sizes)
        self._default = torch.float64 if dtype is None else dtype
        p = self._plan(self._default)
        ny_, nx_ = C.c_int(), C.c_int()
        _lib.check(self._lib.pfb_psi_plan_dims(p, C.byref(ny_), C.byref(nx_)))
        self.Nymax, self.Nxmax = ny_.value, nx_.value

    def _plan(self, dtype):
        p = self._plans.get(dtype)
        if p is None:
            _dev.require_device()
            h = C.c_void_p()
            _lib.check(self._lib.pfb_psi_plan_create(
                self.nband, self.nx, self.ny, self.nbasis, self._ks,
                self._filt.ctypes.data_as(C.POINTER(C.c_double)), self.nlevel, _dev.code(dtype),
                C.byref(h)))
            p = self._plans[dtype] = h
        return p

    def _check(self, x, alpha):
        if tuple(x.shape) != (self.nband, self.nx, self.ny):
            raise ValueError(f"image cube has shape {tuple(x.shape)}, expected "
                             f"{(self.nband, self.nx, self.ny)}")
        if tuple(alpha.shape) != (self.nband, self.nbasis, self.Nymax, self.Nxmax):
            raise ValueError(f"coefficient cube has shape {tuple(alpha.shape)}, expected "
                             f"{(self.nband, self.nbasis, self.Nymax, self.Nxmax)}")

    def dot(self, x, alphao):
        """image to coeffs (psi.py:284-295), in place on alphao."""
        self._check(x, alphao)
        xd = _dev.to_dev(x).contiguous()
        direct = isinstance(alphao, torch.Tensor) and alphao.is_cuda and alphao.is_contiguous() \
            and alphao.dtype == xd.dtype
        ad = alphao if direct else _dev.to_dev(alphao, xd.dtype).contiguous()
        _lib.check(self._lib.pfb_psi_dot(self._plan(xd.dtype), _dev.ptr(xd), _dev.ptr(ad), _dev.stream()))
        if not direct:
            if _dev.is_numpy(alphao):
                alphao[...] = ad.cpu().numpy()
            else:
                alphao.copy_(ad)
        return alphao

    def hdot(self, alpha, xo):
        """coeffs to image (psi.py:297-310), in place on xo."""
        self._check(xo, alpha)
        ad = _dev.to_dev(alpha).contiguous()
        direct = isinstance(xo, torch.Tensor) and xo.is_cuda and xo.is_contiguous() \
            and xo.dtype == ad.dtype
        xd = xo if direct else torch.empty((self.nband, self.nx, self.ny), dtype=ad.dtype, device=ad.device)
        _lib.check(self._lib.pfb_psi_hdot(self._plan(ad.dtype), _dev.ptr(ad), _dev.ptr(xd), _dev.stream()))
        if not direct:
            if _dev.is_numpy(xo):
                xo[...] = xd.cpu().numpy()
            else:
                xo.copy_(xd)
        return xo

    def close(self):
        if self._plans:
            torch.cuda.synchronize()
            for h in self._plans.values():
                self._lib.pfb_psi_plan_destroy(h)
            self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class psi_band(object):
    """One band of the dictionary -- the object pfb/operators/psi.py:17-123 `psi_band_maker` returns (a numba jitclass
    there, psi.py:141-256): `dot(x (nx, ny), alphao (nbasis, Nymax, Nxmax))`, `hdot(alpha, xo (nx, ny))`, both in place,
    attributes `bases, nbasis, nx, ny, Nx, Ny, Nxmax, Nymax`.  Backed by a one-band Psi plan (same kernels)."""

    def __init__(self, psi):
        self._psi = psi
        self.bases, self.nbasis = psi.bases, psi.nbasis
        self.nx, self.ny = psi.nx, psi.ny
        self.Nx, self.Ny = psi.nx, psi.ny          # extents of the 'self' block (psi.py:183-184)
        self.Nxmax, self.Nymax = psi.Nxmax, psi.Nymax

    def dot(self, x, alphao):
        """signal to coeffs (psi.py:187-218)"""
        if x.ndim != 2 or alphao.ndim != 3:
            raise ValueError("psi_band.dot: x is (nx, ny), alphao (nbasis, Nymax, Nxmax)")
        self._psi.dot(x[None], alphao[None])
        return alphao

    def hdot(self, alpha, xo):
        """coeffs to signal (psi.py:220-256)"""
        if xo.ndim != 2 or alpha.ndim != 3:
            raise ValueError("psi_band.hdot: alpha is (nbasis, Nymax, Nxmax), xo (nx, ny)")
        self._psi.hdot(alpha[None], xo[None])
        return xo


def psi_band_maker(nx, ny, bases, nlevel):
    """pfb/operators/psi.py:17-123: the per-band operator for an (nx, ny) image; ValueError for an impossible
    decomposition level (:44-46)."""
    return psi_band(Psi(1, nx, ny, bases, nlevel, 1))
