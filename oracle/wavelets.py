"""
CPU restatement of the separable multi-level 2-D DWT / IDWT (zero-extension mode) and
of the Psi dictionary operator built on it.

Test infrastructure (see oracle/__init__.py).  Follows
  pfb/wavelets/wavelets.py:21-27    coeff_size / signal_size
  pfb/wavelets/wavelets.py:31-95    downsampling_convolution  (analysis, step 2)
  pfb/wavelets/wavelets.py:99-123   upsampling_convolution_valid_sf (synthesis)
  pfb/wavelets/wavelets.py:127-213  dwt2d_level / dwt2d
  pfb/wavelets/wavelets.py:217-315  idwt2d_level / idwt2d
  pfb/operators/psi.py:17-123       psi_band_maker (bookkeeping)
  pfb/operators/psi.py:187-256      psi_band.dot / hdot
  pfb/operators/psi.py:269-310      Psi

The reference's per-row numba loops are vectorised over rows with numpy; the 1-D
formulas are
  analysis :  out[o]     = sum_{j<F} filt[j] in[2o+1-j],  o < (N+F-1)//2, zero outside
  synthesis:  out[2m]   += sum_{j<F/2} filt[2j]   in[m+F/2-1-j]
              out[2m+1] += sum_{j<F/2} filt[2j+1] in[m+F/2-1-j],  m <= N-F/2
Coefficients are packed transposed (y-major), coarser blocks overwrite the approx
quadrant of finer ones, margin cells are never written by dot nor read by hdot.
"""
import numpy as np
from . import daubechies as _db


def coeff_size(nsignal, nfilter):
    return (nsignal + nfilter - 1) // 2


def signal_size(ncoeff, nfilter):
    return 2 * ncoeff - nfilter + 2


def analysis_rows(inp, filt):
    """downsampling_convolution(step=2) applied to every row of `inp` (R, N)."""
    R, N = inp.shape
    F = filt.size
    C = coeff_size(N, F)
    pad = np.zeros((R, N + 2 * F), dtype=inp.dtype)
    pad[:, F - 1:F - 1 + N] = inp
    out = np.zeros((R, C), dtype=inp.dtype)
    for j in range(F):
        # in[2o+1-j] lives at pad[2o + F - j]
        out += filt[j] * pad[:, F - j:F - j + 2 * C:2]
    return out


def synthesis_rows_acc(inp, filt, out):
    """upsampling_convolution_valid_sf on every row; ACCUMULATES into out (R, O)."""
    R, N = inp.shape
    F = filt.size
    h = F // 2
    nm = N - h + 1                      # m = 0 .. N - F/2
    if nm <= 0:
        return
    O = out.shape[1]
    nm = min(nm, O // 2)                # `o < O` guard of the reference loop
    ev = np.zeros((R, nm), dtype=out.dtype)
    od = np.zeros((R, nm), dtype=out.dtype)
    for j in range(h):
        seg = inp[:, h - 1 - j:h - 1 - j + nm]
        ev += filt[2 * j] * seg
        od += filt[2 * j + 1] * seg
    out[:, 0:2 * nm:2] += ev
    out[:, 1:2 * nm:2] += od


class Bookkeeping:
    """psi.py:49-94 for one wavelet: per-level coefficient counts sx/sy, signal
    sizes spx/spy, packing indices ix/iy and totals Ntotx/Ntoty."""

    def __init__(self, nx, ny, F, nlevel):
        self.F = F
        self.nlevel = nlevel
        self.sx, self.sy, self.spx, self.spy = [], [], [], []
        Nx, Ny = nx, ny
        totx = toty = 0
        for _ in range(nlevel):
            Cx = coeff_size(Nx, F)
            Cy = coeff_size(Ny, F)
            totx += Cx
            toty += Cy
            self.sx.append(Cx)
            self.sy.append(Cy)
            Nx = Cx + Cx % 2
            Ny = Cy + Cy % 2
            self.spx.append(signal_size(Cx, F))
            self.spy.append(signal_size(Cy, F))
        self.Ntotx = totx + self.sx[-1]
        self.Ntoty = toty + self.sy[-1]
        self.ix = {}
        self.iy = {}
        lowx, lowy = self.sx[-1], self.sy[-1]
        self.ix[nlevel - 1] = (lowx, 2 * lowx)
        self.iy[nlevel - 1] = (lowy, 2 * lowy)
        lowx *= 2
        lowy *= 2
        for k in reversed(range(nlevel - 1)):
            self.ix[k] = (lowx, lowx + self.sx[k])
            self.iy[k] = (lowy, lowy + self.sy[k])
            lowx += self.sx[k]
            lowy += self.sy[k]


def dwt2d(image, coeffs, bk, dec_lo, dec_hi):
    """wavelets.py:175-213 (with dwt2d_level :127-171 inlined).
    image (nx, ny) -> coeffs (Ntoty, Ntotx), written in place."""
    approx = image
    for lev in range(bk.nlevel):
        Cx, Cy = bk.sx[lev], bk.sy[lev]
        highx = bk.ix[lev][1]
        highy = bk.iy[lev][1]
        lowx = highx - 2 * Cx
        lowy = highy - 2 * Cy
        nx = approx.shape[0]
        # pass 1 along y (contiguous) for every image row -> cbuff (nx, 2Cy)
        cb = np.empty((nx, 2 * Cy), dtype=coeffs.dtype)
        cb[:, :Cy] = analysis_rows(approx, dec_lo)
        cb[:, Cy:] = analysis_rows(approx, dec_hi)
        cbT = np.ascontiguousarray(cb.T)            # (2Cy, nx)  -- copyT
        # pass 2 along x on the transposed buffer -> coeffs block (2Cy, 2Cx)
        blk = coeffs[lowy:highy, lowx:highx]
        blk[:, :Cx] = analysis_rows(cbT, dec_lo)
        blk[:, Cx:] = analysis_rows(cbT, dec_hi)
        approx = blk[0:Cy, 0:Cx].T.copy()           # (Cx, Cy) next level input


def idwt2d(coeffs, image, bk, rec_lo, rec_hi):
    """wavelets.py:261-315 (with idwt2d_level :217-257 inlined).
    coeffs (Ntoty, Ntotx) -> image (nx, ny) written in place; coeffs untouched."""
    alpha = coeffs.copy()
    nx, ny = image.shape
    # the reference reuses one (nx, ny) image buffer across levels
    work = image
    for lev in range(bk.nlevel - 1, -1, -1):
        nax, nay = bk.sx[lev], bk.sy[lev]
        highx = bk.ix[lev][1]
        highy = bk.iy[lev][1]
        lowx = highx - 2 * nax
        lowy = highy - 2 * nay
        nxo, nyo = bk.spx[lev], bk.spy[lev]
        if lev < bk.nlevel - 1:
            alpha[lowy:lowy + nay, lowx:lowx + nax] = work[0:nax, 0:nay].T
        blk = alpha[lowy:highy, lowx:highx]         # (2nay, 2nax)
        cbT = np.zeros((2 * nay, 2 * nax), dtype=image.dtype)
        synthesis_rows_acc(blk[:, 0:nax], rec_lo, cbT)
        synthesis_rows_acc(blk[:, nax:], rec_hi, cbT)
        cb = np.ascontiguousarray(cbT.T)            # (2nax, 2nay)
        out = work[0:nxo, 0:nyo]
        out[...] = 0.0
        synthesis_rows_acc(cb[0:nxo, 0:nay], rec_lo, out)
        synthesis_rows_acc(cb[0:nxo, nay:2 * nay], rec_hi, out)


class Psi:
    """psi.py:269-310 (+ psi_band :154-256).  bases: list of 'self' | 'dbK'."""

    def __init__(self, nband, nx, ny, bases, nlevel, nthreads=1, filter_bank=None):
        fb = filter_bank or _db.filter_bank
        self.nband, self.nx, self.ny = nband, nx, ny
        self.bases = list(bases)
        self.nbasis = len(self.bases)
        self.nlevel = nlevel
        self.nthreads = nthreads
        self.bk = {}
        self.filters = {}
        self.Nxmax = 0
        self.Nymax = 0
        for w in self.bases:
            if w == 'self':
                continue
            max_level = _db.dwt_max_level(min(nx, ny), w)
            if nlevel > max_level:
                raise ValueError(f"The requested decomposition level {nlevel} "
                                 "is not possible")
            F = int(w[-1]) * 2                       # psi.py:59
            self.filters[w] = fb(w)
            self.bk[w] = Bookkeeping(nx, ny, F, nlevel)
            self.Nxmax = max(self.Nxmax, self.bk[w].Ntotx)
            self.Nymax = max(self.Nymax, self.bk[w].Ntoty)

    def dot(self, x, alphao):
        """analysis: x (nband, nx, ny) -> alphao (nband, nbasis, Nymax, Nxmax)."""
        for b in range(self.nband):
            for i, w in enumerate(self.bases):
                if w == 'self':
                    alphao[b, i, 0:self.ny, 0:self.nx] = x[b].T
                    continue
                bk = self.bk[w]
                dl, dh, _, _ = self.filters[w]
                dwt2d(x[b], alphao[b, i, 0:bk.Ntoty, 0:bk.Ntotx], bk, dl, dh)

    def hdot(self, alpha, xo):
        """synthesis: alpha -> xo (nband, nx, ny), accumulated over bases."""
        img = np.zeros((self.nx, self.ny), dtype=xo.dtype)
        for b in range(self.nband):
            xo[b] = 0.0
            for i, w in enumerate(self.bases):
                if w == 'self':
                    xo[b] += alpha[b, i, 0:self.ny, 0:self.nx].T
                    continue
                bk = self.bk[w]
                _, _, rl, rh = self.filters[w]
                idwt2d(alpha[b, i, 0:bk.Ntoty, 0:bk.Ntotx], img, bk, rl, rh)
                xo[b] += img
