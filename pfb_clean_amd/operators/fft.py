"""
PSFHAT production -- pfb/operators/gridder.py:712-714 with pfb/operators/fft.py:7-9:

    psfhat = r2c(ifftshift(psf), axes=(0, 1), inorm=0)

This runs ONCE per gridding run (plan time), not inside the PCG / PD loops.  It is produced by the library's own
kernels for every grid whose lengths are 13-smooth with an even ny_psf (what pfb's grid worker makes, grid.py:276-285):

  * power-of-two grids in the fast path's range (nx_psf <= 16384, ny_psf <= 32768 fp32 / 16384 fp64: every BASELINE
    size) on the register-FFT row / column kernels of the convolution itself (pfb_psfconv_set_psf on a fast plan);
  * any other grid through pfb_psfhat_from_psf: one workgroup per line in LDS while a line fits, multi-launch
    Stockham passes in global memory beyond that -- no length limit.

An ODD ny_psf (which the reference never produces), a length with a prime factor above 13 and a grid too large for
the throw-away plan fall back to torch.fft on the device tensor: like the reference, no size is refused.
`PsfConvPlan.from_psf` builds the convolution plan from the PSF without the transform ever leaving the library.
"""
import ctypes as C

import torch

from .. import _dev, _lib
from .psf import PsfConvPlan, NX_FAST_MAX, NY_FAST_MAX


def _is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def psfhat_from_psf(psf):
    """psf: (nx_psf, ny_psf) or (nband, nx_psf, ny_psf) real, peak at the centre.
    Returns complex psfhat (..., nx_psf, ny_psf//2 + 1), numpy in -> numpy out."""
    p = _dev.to_dev(psf)
    squeeze = p.ndim == 2
    if squeeze:
        p = p[None]
    if p.ndim != 3:
        raise ValueError("psf must be (nx_psf, ny_psf) or (nband, nx_psf, ny_psf)")
    nband, P, Q = (int(v) for v in p.shape)
    out = None
    if Q % 2 == 0 and p.dtype in (torch.float32, torch.float64):
        p = p.contiguous()
        # The reference's r2c(ifftshift(psf)) takes ANY size.  What the library's kernels do not take -- a prime factor
        # above 13 (PFB_ERR_UNSUPPORTED), or no room for the throw-away plan of a huge grid (PFB_ERR_ALLOC) -- degrades
        # to torch.fft on the same device tensor instead of failing; every other error is a real one and propagates.
        soft = (_lib.PFB_ERR_UNSUPPORTED, _lib.PFB_ERR_ALLOC)
        try:
            if (_is_pow2(P) and _is_pow2(Q) and 128 <= P <= 2 * NX_FAST_MAX and 256 <= Q <= 2 * NY_FAST_MAX[p.dtype]):
                # a fast plan for the (P/2, Q/2) image this grid oversamples by 2: its row / column kernels do the work
                plan, out = PsfConvPlan.from_psf(p, P // 2, Q // 2, want_psfhat=True)
                plan.close()
            else:
                lib = _lib.load()
                out = torch.empty((nband, P, Q // 2 + 1), dtype=_dev.CPLX_OF[p.dtype], device=p.device)
                _lib.check(lib.pfb_psfhat_from_psf(_dev.code(p.dtype), _dev.ptr(p), nband, P, Q, _dev.ptr(out), _dev.stream()))
        except _lib.PfbHipError as e:
            if e.code not in soft:
                raise
            out = None
    if out is None:       # odd ny_psf (the reference's grid worker never makes one), or a size the kernels refuse
        out = torch.fft.rfft2(torch.fft.ifftshift(p, dim=(-2, -1)), dim=(-2, -1))
    if squeeze:
        out = out[0]
    return out.cpu().numpy() if _dev.is_numpy(psf) else out
