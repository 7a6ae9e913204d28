"""
PSFHAT production -- pfb/operators/gridder.py:712-714 with pfb/operators/fft.py:7-9:

    psfhat = r2c(ifftshift(psf), axes=(0, 1), inorm=0)

This runs ONCE per gridding run (plan time), not inside the PCG / PD loops, so it is glue and
not one of the hand-written hot kernels: it is expressed with torch.fft on the device tensor
(rocFFT underneath) purely so that a worker can stay device-resident between the gridder's
PSF and `PsfConvPlan`.  SURVEY 8f ranks a native version as "next" (f2).
"""
import torch

from .. import _dev


def psfhat_from_psf(psf):
    """psf: (nx_psf, ny_psf) or (nband, nx_psf, ny_psf) real, peak at the centre.
    Returns complex psfhat (..., nx_psf, ny_psf//2 + 1)."""
    p = _dev.to_dev(psf)
    out = torch.fft.rfft2(torch.fft.ifftshift(p, dim=(-2, -1)), dim=(-2, -1))
    return out.cpu().numpy() if _dev.is_numpy(psf) else out
