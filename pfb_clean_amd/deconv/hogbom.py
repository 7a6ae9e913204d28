"""
Hogbom CLEAN on MI355X -- drop-in for pfb/deconv/hogbom.py:8-74 (imported by the klean worker,
workers/klean.py:68).  The whole loop runs in libpfb_hip.so (pfb_hogbom): two launches per iteration
(subtract the shifted PSF + arg-max partials; final arg-max, stop test, model update), loop state on
the device, the host looks once per 64 iterations.  Returns (model, status): 0 converged, 1 maxit.
"""
import ctypes as C
import sys

import torch

from .. import _lib, _dev


def hogbom(ID, PSF, threshold=0, gamma=0.1, pf=0.1, maxit=10000, report_freq=1000, verbosity=1):
    lib = _lib.load()
    as_numpy = _dev.is_numpy(ID)
    IR = _dev.to_dev(ID).contiguous().clone()
    dt = IR.dtype
    psf = _dev.to_dev(PSF, dt).contiguous()
    nband, nx, ny = IR.shape
    _, nx_psf, ny_psf = psf.shape
    wsums = psf.amax(dim=(1, 2)).contiguous()                       # hogbom.py:27
    if not bool((wsums > 0).all().item()):
        raise ValueError("hogbom: every band's PSF peak must be positive (hogbom.py:28,33-34 broadcast)")
    model = torch.zeros_like(IR)
    work = torch.zeros(17408, dtype=torch.uint8, device=IR.device)
    k, irmax = C.c_int(), C.c_double()
    _lib.check(lib.pfb_hogbom(_dev.code(dt), _dev.ptr(IR), _dev.ptr(psf), _dev.ptr(model), _dev.ptr(wsums), nband,
                              nx, ny, nx_psf, ny_psf, float(gamma), float(pf), float(threshold), int(maxit),
                              _dev.ptr(work), work.numel(), C.addressof(k), C.addressof(irmax), _dev.stream()))
    status = 1 if k.value >= maxit else 0
    if verbosity:
        quiet = ~(model != 0).any(dim=0)
        rms = float(IR.sum(dim=0)[quiet].std(unbiased=False).item()) if bool(quiet.any().item()) else float('nan')
        msg = "Max iters reached. " if status else f"Success, converged after {k.value} iterations. "
        print(f"{msg}Max resid = {irmax.value:.3e}, rms = {rms:.3e}", file=sys.stderr)
    return (model.cpu().numpy() if as_numpy else model), status
