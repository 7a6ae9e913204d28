"""
PSFHAT production -- pfb/operators/gridder.py:712-714 with pfb/operators/fft.py:7-9:

    psfhat = r2c(ifftshift(psf), axes=(0, 1), inorm=0)

This runs ONCE per gridding run (plan time), not inside the PCG / PD loops.  It is produced by
the library's own line-FFT kernels (pfb_psfconv_set_psf: one workgroup per PSF row / column,
mixed-radix Stockham in LDS) whenever a PSF line fits the LDS (nx_psf <= 10240 fp32 / 5120
fp64, lengths 13-smooth, ny_psf even); larger or odd grids go through torch.fft on the device
tensor (rocFFT) so that a worker still stays device resident.  `PsfConvPlan.from_psf` builds
the convolution plan from the PSF without the transform ever leaving the library.
"""
import torch

from .. import _dev, _lib
from .._lib import PfbHipError
from .psf import PsfConvPlan


def psfhat_from_psf(psf):
    """psf: (nx_psf, ny_psf) or (nband, nx_psf, ny_psf) real, peak at the centre.
    Returns complex psfhat (..., nx_psf, ny_psf//2 + 1), numpy in -> numpy out."""
    p = _dev.to_dev(psf)
    squeeze = p.ndim == 2
    out = None
    if p.shape[-1] % 2 == 0 and p.dtype in (torch.float32, torch.float64):
        try:
            # a throw-away plan for a 1 x 2 image: only its tables and the PSF kernels are used
            plan, out = PsfConvPlan.from_psf(p, 1, 2, want_psfhat=True)
            plan.close()
        except PfbHipError as e:
            if e.code != _lib.PFB_ERR_UNSUPPORTED:
                raise
            out = None
    if out is None:
        out = torch.fft.rfft2(torch.fft.ifftshift(p, dim=(-2, -1)), dim=(-2, -1))
    elif squeeze:
        out = out[0]
    return out.cpu().numpy() if _dev.is_numpy(psf) else out
