"""
Clark CLEAN on MI355X -- drop-in for pfb/deconv/clark.py:86-177 (`clark`), the minor cycle of the
klean worker (workers/klean.py:206) and the third caller of the PSF convolution.

Per major iteration (reference statement -> device work):
    IRsearch = sum_b(IR)^2, arg-max, threshold        image-wide elementwise passes + reductions
                                                      (torch device ops: plumbing, once per iteration)
    Ip, Iq = where(IRsearch > subth^2)                torch.nonzero (dynamic size)
    model = subminor(IR[:, Ip, Iq], PSF, ...)         pfb_clark_subminor: the sequential greedy loop
                                                      (<= submaxit arg-max/subtract steps) in ONE
                                                      resident workgroup, no per-step launch
    psf_convolve_cube(..., PSFHAT, ny_psf, model)     the fused convolution kernels
    IR = ID - xout
Everything stays on the GPU; one scalar (the new peak) comes back per major iteration for the
`while IRmax > tol` test.  Quirks kept from the reference: the sub-minor loop subtracts the FULL
component from the active set while the model only receives gamma * component / wsum
(clark.py:66,72-75); `stall_count += stall_count` never leaves zero (clark.py:155); wsums must sum
to one.  Returns (model, status) with status 0 = converged, 1 = maxit.
"""
import sys

import numpy as np
import torch

from .. import _lib, _dev
from ..operators.psf import psf_convolve_cube


def clark(ID, PSF, PSFHAT, wsums, threshold=0, gamma=0.05, pf=0.05, maxit=50, subpf=0.5,
          submaxit=1000, report_freq=1, verbosity=1, psfopts=None, sigmathreshold=2, nthreads=1):
    lib = _lib.load()
    as_numpy = _dev.is_numpy(ID)
    IDd = _dev.to_dev(ID).contiguous()
    dt = IDd.dtype
    code = _dev.code(dt)
    psf = _dev.to_dev(PSF, dt).contiguous()
    psfhat = _dev.to_dev(PSFHAT)
    w = _dev.to_dev(wsums, dt).contiguous()
    nband, nx, ny = IDd.shape
    _, nx_psf, ny_psf = psf.shape
    if not np.allclose(float(w.sum().item()), 1.0):
        raise AssertionError("clark: wsums must sum to one (clark.py:108)")
    if not bool((w > 0).any().item()):
        raise ValueError("wsums are all zero")
    model = torch.zeros_like(IDd)
    IR = IDd.clone()
    iters = torch.zeros(1, dtype=torch.int32, device=IDd.device)

    def peak(IR):
        s = IR.sum(dim=0)
        IRsearch = s * s
        pq = int(torch.argmax(IRsearch).item())
        return IRsearch, float(torch.sqrt(IRsearch.reshape(-1)[pq]).item())

    IRsearch, IRmax = peak(IR)
    tol = max(pf * IRmax, threshold)
    k = 0
    stall_count = 0
    while IRmax > tol and k < maxit and stall_count < 5:
        subth = subpf * IRmax
        act = torch.nonzero(IRsearch > subth ** 2)              # row-major order, like np.where
        Ip = act[:, 0].to(torch.int32).contiguous()
        Iq = act[:, 1].to(torch.int32).contiguous()
        A = IR[:, act[:, 0], act[:, 1]].contiguous()
        _lib.check(lib.pfb_clark_subminor(code, _dev.ptr(A), A.shape[1], nband, _dev.ptr(psf), nx_psf, ny_psf,
                                          _dev.ptr(Ip), _dev.ptr(Iq), _dev.ptr(model), nx, ny, _dev.ptr(w),
                                          float(gamma), float(subth), int(submaxit), _dev.ptr(iters),
                                          _dev.stream()))
        xout = psf_convolve_cube(None, None, None, psfhat, ny_psf, model, nthreads=nthreads)
        IR = IDd - xout
        IRsearch, IRmaxn = peak(IR)
        IRmaxp, IRmax = IRmax, IRmaxn
        k += 1
        if abs(IRmaxp - IRmax) / abs(IRmaxp) < 1e-3:
            stall_count += stall_count                          # sic (clark.py:155)
        if not k % report_freq and verbosity > 1:
            print(f"At iteration {k} max resid = {IRmax}", file=sys.stderr)
    status = 1 if (k >= maxit or stall_count >= 5) else 0
    if verbosity:
        quiet = ~(model != 0).any(dim=0)
        rms = float(IR.sum(dim=0)[quiet].std(unbiased=False).item()) if bool(quiet.any().item()) else float('nan')
        msg = ("Max iters reached. " if k >= maxit else "Stalled. " if stall_count >= 5
               else f"Success, converged after {k} iterations. ")
        print(f"{msg}Max resid = {IRmax:.3e}, rms = {rms:.3e}", file=sys.stderr)
    return (model.cpu().numpy() if as_numpy else model), status
