"""
CPU restatement of the Clark CLEAN minor cycle, pfb/deconv/clark.py:11-177 (test infrastructure, see
oracle/__init__.py).  Vectorised numpy where the reference loops (subtract), same statement order
otherwise; requires the PSF to cover every offset of the image (nx_psf/2 >= nx - 1), where the
reference's overlap mask (clark.py:68-75) is all true.
"""
import numpy as np
from . import fftconv as _fc


def subminor(A, psf, Ip, Iq, model, wsums, gamma=0.05, th=0.0, maxit=10000):
    """clark.py:29-84.  Returns (model, number of components taken)."""
    nband, nx_psf, ny_psf = psf.shape
    nxo2, nyo2 = nx_psf // 2, ny_psf // 2
    A = A.copy()
    Asearch = np.sum(A, axis=0) ** 2
    pq = Asearch.argmax()
    p, q = Ip[pq], Iq[pq]
    Amax = np.sqrt(Asearch[pq])
    fsel = wsums > 0
    if fsel.sum() == 0:
        raise ValueError("wsums are all zero")
    k = 0
    while Amax > th and k < maxit:
        xhat = A[:, pq].copy()
        model[fsel, p, q] += gamma * xhat[fsel] / wsums[fsel]
        A = A - xhat[:, None] * psf[:, nxo2 - (p - Ip), nyo2 - (q - Iq)]     # the FULL component (sic)
        Asearch = np.sum(A, axis=0) ** 2
        pq = Asearch.argmax()
        p, q = Ip[pq], Iq[pq]
        Amax = np.sqrt(Asearch[pq])
        k += 1
    return model, k


def clark(ID, PSF, PSFHAT, wsums, threshold=0, gamma=0.05, pf=0.05, maxit=50, subpf=0.5,
          submaxit=1000, nthreads=1):
    """clark.py:86-177.  Returns (model, status)."""
    nband, nx, ny = ID.shape
    _, nx_psf, ny_psf = PSF.shape
    assert np.allclose(wsums.sum(), 1)
    model = np.zeros((nband, nx, ny), dtype=ID.dtype)
    IR = ID.copy()
    xpad, xhat, xout = _fc.make_scratch(PSFHAT, ny_psf, ID.shape, ID.dtype)
    IRsearch = np.sum(IR, axis=0) ** 2
    pq = IRsearch.argmax()
    p = pq // ny
    q = pq - p * ny
    IRmax = np.sqrt(IRsearch[p, q])
    tol = np.maximum(pf * IRmax, threshold)
    k = 0
    stall_count = 0
    while IRmax > tol and k < maxit and stall_count < 5:
        subth = subpf * IRmax
        Ip, Iq = np.where(IRsearch > subth ** 2)
        model, _ = subminor(IR[:, Ip, Iq], PSF, Ip, Iq, model, wsums, gamma=gamma, th=subth, maxit=submaxit)
        _fc.psf_convolve_cube(xpad, xhat, xout, PSFHAT, ny_psf, model, nthreads=nthreads)
        IR = ID - xout
        IRsearch = np.sum(IR, axis=0) ** 2
        pq = IRsearch.argmax()
        p = pq // ny
        q = pq - p * ny
        IRmaxp = IRmax
        IRmax = np.sqrt(IRsearch[p, q])
        k += 1
        if np.abs(IRmaxp - IRmax) / np.abs(IRmaxp) < 1e-3:
            stall_count += stall_count
    return model, (1 if (k >= maxit or stall_count >= 5) else 0)


def hogbom(ID, PSF, threshold=0, gamma=0.1, pf=0.1, maxit=10000):
    """pfb/deconv/hogbom.py:8-74.  Returns (model, status, residual)."""
    nband, nx, ny = ID.shape
    _, nx_psf, ny_psf = PSF.shape
    nx0, ny0 = nx_psf // 2, ny_psf // 2
    x = np.zeros((nband, nx, ny), dtype=ID.dtype)
    IR = ID.copy()
    IRsearch = np.sum(IR, axis=0) ** 2
    pq = IRsearch.argmax()
    p = pq // ny
    q = pq - p * ny
    IRmax = np.sqrt(IRsearch[p, q])
    wsums = np.amax(PSF, axis=(1, 2))
    tol = np.maximum(pf * IRmax, threshold)
    k = 0
    while IRmax > tol and k < maxit:
        xhat = IR[:, p, q] / wsums
        x[:, p, q] += gamma * xhat
        IR = IR - gamma * xhat[:, None, None] * PSF[:, nx0 - p:nx0 + nx - p, ny0 - q:ny0 + ny - q]
        IRsearch = np.sum(IR, axis=0) ** 2
        pq = IRsearch.argmax()
        p = pq // ny
        q = pq - p * ny
        IRmax = np.sqrt(IRsearch[p, q])
        k += 1
    return x, (1 if k >= maxit else 0), IR
