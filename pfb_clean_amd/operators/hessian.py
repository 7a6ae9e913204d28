"""
PSF-approximated, Tikhonov-regularised Hessians on MI355X -- drop-in for
pfb/operators/hessian.py:129-158 (_hessian_psf_slice) and :254-281 (hessian_psf_cube).

    out = [beam*] psf_convolve([beam*] x) [/ wsum] + sigmainv * x

beam multiply, 1/wsum, the Tikhonov term and (optionally) <p, A p> are fused into the
row kernels of the convolution; the reference's 3-4 extra numpy passes disappear.
Argument orders are the reference's (they differ between the two functions!).
Unlike psf_convolve_* these return a NEW array (hessian.py:158,281: `xout + x*sigmainv`),
and so do we; `xout` additionally receives the result (the reference leaves the
un-regularised convolution there, which no caller reads).

HessianPsf is the object form used by the fused PCG (opt/pcg.py): it carries the plan
and the operator parameters so that `pcg(A, b, ...)` can run the whole solve inside
libpfb_hip.so when A is one of these.
"""
import torch

from .. import _dev
from .psf import plan_for, PsfConvPlan


class HessianPsf:
    """A(x) = [beam*]conv([beam*]x)[/wsum] + sigmainv*x on bands [band0, band0+nb).

    psfhat: (nband, nx_psf, nyo2) | (nx_psf, nyo2) complex (numpy or tensor) or an
    existing PsfConvPlan.  beam: None | (nb, nx, ny) | (nx, ny).  Callable on GPU
    tensors or numpy arrays of shape (nb, nx, ny) or (nx, ny)."""

    def __init__(self, psfhat, nx, ny, lastsize, beam=None, sigmainv=0.0, wsum=None,
                 band0=0, nb=None):
        self.plan = psfhat if isinstance(psfhat, PsfConvPlan) else plan_for(psfhat, nx, ny, lastsize)
        self.nx, self.ny = int(nx), int(ny)
        self.band0 = int(band0)
        self.nb = int(nb) if nb is not None else self.plan.nband - self.band0
        self.sigmainv = float(sigmainv)
        self.wsum = None if wsum is None else float(wsum)
        self.beam = None
        if beam is not None:
            b = _dev.to_dev(beam, self.plan.rdtype)
            if b.ndim == 2:
                b = b[None].expand(self.nb, -1, -1)
            if tuple(b.shape) != (self.nb, self.nx, self.ny):
                raise ValueError('Beam has incorrect shape')
            self.beam = b.contiguous()

    @property
    def dtype(self):
        return self.plan.rdtype

    def __call__(self, x, out=None):
        xd = _dev.to_dev(x)
        squeeze = xd.ndim == 2
        if squeeze and self.nb != 1:
            raise ValueError("2-D input to a multi-band HessianPsf")
        res = self.plan.apply(xd, out=out, beam=self.beam, wsum=self.wsum,
                              sigmainv=self.sigmainv, band0=self.band0)
        return res.cpu().numpy() if _dev.is_numpy(x) else res


class hessian_psf_slice:
    """pfb/operators/hessian.py:161-251 -- the per-band stateful operator (only referenced
    from the commented-out distributed spotless, workers/spotless.py:439-520).
    Same constructor signature and attributes; the image-space members live on the GPU
    (torch tensors), `__call__` is _hessian_psf_slice through the band's own plan with the
    wsum given to set_wsum.  `ds` is the reference's per-band dataset or anything exposing the
    same variables with `.values` (DIRTY, PSFHAT, PSF, BEAM, WSUM [, MODEL, DUAL, RESIDUAL])
    and `bandid`.  compute_residual is visibility space (wgridder) -- outside this package."""

    def __init__(self, ds, nbasis, nmax, nthreads, sigmainv, cell=None, do_wgridding=None,
                 epsilon=None, double_accum=None):
        self.nthreads = nthreads
        self.sigmainv = sigmainv
        self.cell, self.do_wgridding, self.epsilon, self.double_accum = cell, do_wgridding, epsilon, double_accum
        self.lastsize = ds.PSF.shape[-1]
        self.bandid = ds.bandid
        self.dirty = _dev.to_dev(ds.DIRTY.values).contiguous()
        rdt = self.dirty.dtype
        self.psfhat = _dev.to_dev(ds.PSFHAT.values).contiguous()
        self.psf = _dev.to_dev(ds.PSF.values, rdt).contiguous()
        self.beam = _dev.to_dev(ds.BEAM.values, rdt).contiguous()
        self.wsumb = ds.WSUM.values[0]
        self.model = (_dev.to_dev(ds.MODEL.values, rdt).contiguous() if 'MODEL' in ds
                      else torch.zeros_like(self.dirty))
        if 'DUAL' in ds:
            self.dual = _dev.to_dev(ds.DUAL.values, rdt).contiguous()
            assert tuple(self.dual.shape) == (nbasis, nmax)
        else:
            self.dual = torch.zeros((nbasis, nmax), dtype=rdt, device=self.dirty.device)
        self.residual = (_dev.to_dev(ds.RESIDUAL.values, rdt).contiguous() if 'RESIDUAL' in ds
                         else self.dirty.clone())
        nx, ny = self.dirty.shape
        self._op = HessianPsf(self.psfhat, nx, ny, self.lastsize, beam=self.beam, sigmainv=sigmainv)

    def __call__(self, x):
        self._op.wsum = None if self.wsum is None else float(self.wsum)
        self._op.sigmainv = float(self.sigmainv)
        return self._op(x)

    def compute_residual(self, x):
        raise NotImplementedError("visibility-space residual (wgridder) is outside the PSF-convolution "
                                  "hot path; use the reference's _hessian_impl")

    def set_wsum(self, wsum):
        self.wsum = wsum


def _hess(psfhat, beam, lastsize, x, xout, sigmainv, wsum):
    xd = _dev.to_dev(x)
    nx, ny = xd.shape[-2:]
    plan = plan_for(psfhat, nx, ny, lastsize)
    if xd.dtype != plan.rdtype:
        raise TypeError(f"x is {xd.dtype} but psfhat is {psfhat.dtype}")
    bd = None
    if beam is not None:
        bd = _dev.to_dev(beam, plan.rdtype)
        if tuple(bd.shape) != tuple(xd.shape):
            raise ValueError('Beam has incorrect shape')
    res = plan.apply(xd, beam=bd, wsum=wsum, sigmainv=sigmainv)
    if xout is not None:
        if _dev.is_numpy(xout):
            xout[...] = res.cpu().numpy()
        elif isinstance(xout, torch.Tensor):
            xout.copy_(res)
    return res.cpu().numpy() if _dev.is_numpy(x) else res


def _hessian_psf_slice(xpad, xhat, xout, psfhat, beam, lastsize, x,
                       nthreads=1, sigmainv=1, wsum=None):
    """pfb/operators/hessian.py:129-158 -- note (psfhat, beam) order."""
    if x.ndim != 2:
        raise ValueError("_hessian_psf_slice expects a 2-D image")
    return _hess(psfhat, beam, lastsize, x, xout, sigmainv, wsum)


def hessian_psf_cube(xpad, xhat, xout, beam, psfhat, lastsize, x,
                     nthreads=1, sigmainv=1, wsum=None):
    """pfb/operators/hessian.py:254-281 -- note (beam, psfhat) order."""
    if x.ndim != 3:
        raise ValueError("hessian_psf_cube expects a (nband, nx, ny) cube")
    return _hess(psfhat, beam, lastsize, x, xout, sigmainv, wsum)
