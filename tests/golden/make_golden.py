#!/usr/bin/env python
"""
Golden-vector generator: runs the REFERENCE's own source files
(/root/reference/pfb/{operators/psf,operators/hessian,opt/pcg,opt/power_method,
opt/primal_dual,operators/psi,wavelets/wavelets,prox/prox_21m,prox/prox_21}.py) under
the stub third-party modules of _refstubs.py and stores inputs + reference outputs
as small .npz fixtures next to this file.

Run in the BUILD container only:   python tests/golden/make_golden.py
(/root/reference is absent on the GPU box; the tests only read the .npz files.)

Seeds: numpy.random.default_rng(420 + case index); 420 is the reference's own test
seed (tests/test_spotless.py:18).
"""
import os
import sys
from functools import partial
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _refstubs  # noqa: E402

_refstubs.install(ROOT)

from pfb.operators.psf import psf_convolve_slice, psf_convolve_cube  # noqa: E402
from pfb.operators.hessian import _hessian_psf_slice, hessian_psf_cube, hessian_psf_slice  # noqa: E402
from pfb.opt.pcg import pcg, _pcg_psf_impl, pcg_dist, cg_dct, cg  # noqa: E402
from pfb.opt.power_method import power_method  # noqa: E402
from pfb.opt.primal_dual import primal_dual_optimised, primal_dual  # noqa: E402
from pfb.operators.psi import Psi  # noqa: E402
from pfb.prox.prox_21m import (prox_21m, prox_21m_numba, dual_update,  # noqa: E402
                               dual_update_numba)
from pfb.prox.prox_21 import prox_21  # noqa: E402
from pfb.prox.prox_21 import prox_21_numba as prox_21_numba_l2, dual_update_numba as dual_update_numba_l2  # noqa: E402
from pfb.deconv.clark import clark, subminor  # noqa: E402
from pfb.deconv.hogbom import hogbom  # noqa: E402
import scipy.fft as sfft  # noqa: E402


def rand_psfhat(rng, nb, P, Q, dtype=np.float64):
    """Random NON-symmetric real PSF peaked at the centre -> complex psfhat
    (gridder.py:712: r2c(ifftshift(psf)))."""
    psf = 0.05 * rng.standard_normal((nb, P, Q))
    u = (np.arange(P) - P // 2)[:, None]
    v = (np.arange(Q) - Q // 2)[None, :]
    psf += np.exp(-(u ** 2 + v ** 2) / (2 * 2.5 ** 2))[None]
    psf = psf.astype(dtype)
    psfhat = sfft.rfftn(sfft.ifftshift(psf, axes=(1, 2)), axes=(1, 2))
    return psf, psfhat


def psd_psfhat(rng, nb, P, Q):
    """PSD operator: psfhat = non-negative uv weights, sum_b psf peaks at 1."""
    u = sfft.fftfreq(P)[:, None]
    v = sfft.rfftfreq(Q)[None, :]
    lam = 4 * np.exp(-(u ** 2 + v ** 2) / (2 * 0.12 ** 2))
    W = rng.poisson(lam, size=(nb, P, Q // 2 + 1)).astype(np.float64)
    psf = sfft.irfftn(W, s=(P, Q), axes=(1, 2))
    W /= nb * psf.max(axis=(1, 2))[:, None, None]
    return W.astype(np.complex128)


def scratch(psfhat, Q, shape):
    if psfhat.ndim == 2:
        P, nyo2 = psfhat.shape
        return (np.empty((P, Q)), np.empty((P, nyo2), dtype=psfhat.dtype),
                np.empty(shape))
    nb, P, nyo2 = psfhat.shape
    return (np.empty((nb, P, Q)), np.empty((nb, P, nyo2), dtype=psfhat.dtype),
            np.empty(shape))


def gen_conv():
    out = {}
    cases = [(32, 32, 64, 64), (48, 40, 96, 80), (64, 64, 128, 128),
             (32, 32, 48, 40), (30, 50, 60, 100), (16, 128, 32, 256)]
    out['cases'] = np.array(cases)
    for c, (nx, ny, P, Q) in enumerate(cases):
        rng = np.random.default_rng(420 + c)
        nb = 2
        psf, psfhat = rand_psfhat(rng, nb, P, Q)
        x = rng.standard_normal((nb, nx, ny))
        beam = 0.5 + rng.random((nb, nx, ny))
        # slice
        xpad, xhat, xout = scratch(psfhat[0], Q, (nx, ny))
        ys = psf_convolve_slice(xpad, xhat, xout, psfhat[0], Q, x[0]).copy()
        # cube
        xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
        yc = psf_convolve_cube(xpad, xhat, xout, psfhat, Q, x).copy()
        # hessians: beam+wsum+sigmainv, and bare
        xpad, xhat, xout = scratch(psfhat[1], Q, (nx, ny))
        h1 = _hessian_psf_slice(xpad, xhat, xout, psfhat[1], beam[1], Q, x[1],
                                nthreads=1, sigmainv=0.37, wsum=2.5).copy()
        h2 = _hessian_psf_slice(xpad, xhat, xout, psfhat[1], None, Q, x[1],
                                nthreads=1, sigmainv=0.0, wsum=None).copy()
        xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
        h3 = hessian_psf_cube(xpad, xhat, xout, beam, psfhat, Q, x,
                              nthreads=1, sigmainv=1.25, wsum=3.0).copy()
        h4 = hessian_psf_cube(xpad, xhat, xout, None, psfhat, Q, x,
                              nthreads=1, sigmainv=0.5, wsum=None).copy()
        out.update({f'c{c}_psf': psf, f'c{c}_psfhat': psfhat, f'c{c}_x': x,
                    f'c{c}_beam': beam, f'c{c}_slice': ys, f'c{c}_cube': yc,
                    f'c{c}_h_slice_full': h1, f'c{c}_h_slice_bare': h2,
                    f'c{c}_h_cube_full': h3, f'c{c}_h_cube_bare': h4})
    np.savez_compressed(os.path.join(HERE, 'conv.npz'), **out)
    print('conv.npz', len(out))


def gen_pcg():
    out = {}
    rng = np.random.default_rng(431)
    nb, nx, ny, P, Q = 3, 48, 40, 96, 80
    psfhat = psd_psfhat(rng, nb, P, Q)
    model = np.zeros((nb, nx, ny))
    for _ in range(6):
        i, j = rng.integers(5, nx - 5), rng.integers(5, ny - 5)
        model[:, i, j] = 1 + np.exp(rng.standard_normal())
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx))
    model += 0.5 * np.exp(-((xx - 20) ** 2 + (yy - 22) ** 2) / 18.0)[None]
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
    b = psf_convolve_cube(xpad, xhat, xout, psfhat, Q, model).copy()
    b += 1e-3 * rng.standard_normal(b.shape)
    beam = 0.7 + 0.3 * np.exp(-((xx - nx / 2) ** 2 + (yy - ny / 2) ** 2) / 400.0)
    beam = np.tile(beam[None], (nb, 1, 1))
    sigmainv = 1e-3 * np.abs(b).max()
    out.update(psfhat=psfhat, b=b, beam=beam, sigmainv=sigmainv, Q=Q)

    # (1) per-band PCG (klean flux-mop semantics, pcg.py:243-291), iterate history
    for tag, bm in (('nobeam', None), ('beam', beam)):
        for k in (1, 2, 5, 20):
            for bt in (True, False):
                m = _pcg_psf_impl(psfhat, b, np.zeros_like(b),
                                  bm if bm is not None else [None] * nb, Q, 1, sigmainv,
                                  tol=0.0, maxit=k, minit=k, verbosity=0,
                                  backtrack=bt)
                out[f'band_{tag}_k{k}_bt{int(bt)}'] = m
    # convergence-controlled exit (tol hit before maxit; minit forcing)
    m = _pcg_psf_impl(psfhat, b, np.zeros_like(b), [None] * nb, Q, 1, sigmainv,
                      tol=1e-2, maxit=100, minit=1, verbosity=0, backtrack=True)
    out['band_tol1e-2'] = m
    m = _pcg_psf_impl(psfhat, b, np.zeros_like(b), [None] * nb, Q, 1, sigmainv,
                      tol=1e-2, maxit=100, minit=15, verbosity=0, backtrack=True)
    out['band_tol1e-2_minit15'] = m

    # (2) cube PCG with global dots (fluxmop semantics, fluxmop.py:160-199)
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
    A = partial(hessian_psf_cube, xpad, xhat, xout, beam, psfhat, Q,
                nthreads=1, sigmainv=sigmainv, wsum=1.0)
    for k in (1, 3, 10, 25):
        x, r = pcg(A, beam * b, np.zeros_like(b), tol=0.0, maxit=k, minit=k,
                   verbosity=0, backtrack=True, return_resid=True)
        out[f'cube_k{k}_x'] = x
        out[f'cube_k{k}_r'] = r
    # with diagonal preconditioner and non-zero x0
    x0 = 0.1 * rng.standard_normal(b.shape)
    out['cube_x0'] = x0
    x = pcg(A, beam * b, x0, M=lambda v: v / sigmainv, tol=0.0, maxit=7, minit=7,
            verbosity=0, backtrack=True)
    out['cube_M_x0_k7'] = x
    # zero-residual early exit returns x0 (pcg.py:73-75)
    xz = pcg(A, A(x0).copy(), x0, tol=1e-5, maxit=5, minit=1, verbosity=0)
    out['cube_zero_resid_is_x0'] = np.array(xz is x0)

    # (3) operator for which backtracking actually triggers: indefinite "psf"
    rng2 = np.random.default_rng(777)
    ph = psfhat[0].copy()
    ph -= 0.35 * ph.real.max()          # makes A indefinite -> rnorm can grow
    xpad, xhat, xout = scratch(ph, Q, (nx, ny))
    A2 = partial(_hessian_psf_slice, xpad, xhat, xout, ph, None, Q,
                 nthreads=1, sigmainv=sigmainv)
    bb = rng2.standard_normal((nx, ny))
    out['indef_psfhat'] = ph
    out['indef_b'] = bb
    for bt in (True, False):
        for k in (3, 8):
            out[f'indef_k{k}_bt{int(bt)}'] = pcg(A2, bb, None, tol=0.0, maxit=k, minit=k,
                                                 verbosity=0, backtrack=bt)

    # (4) power method with fixed start (power_method.py:11-49)
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
    conv = partial(psf_convolve_cube, xpad, xhat, xout, psfhat, Q)
    b0 = rng.standard_normal((nb, nx, ny))
    beta, bvec = power_method(conv, (nb, nx, ny), b0=b0.copy(), tol=1e-3, maxit=40,
                              verbosity=0)
    out.update(pm_b0=b0, pm_beta=beta, pm_b=bvec.copy())
    np.savez_compressed(os.path.join(HERE, 'pcg.npz'), **out)
    print('pcg.npz', len(out))


def gen_psi():
    out = {}
    cases = [  # (nband, nx, ny, bases, nlevel)
        (2, 128, 64, ['self', 'db1', 'db2', 'db3', 'db4', 'db5'], 2),
        (1, 250, 78, ['self', 'db1', 'db2', 'db3', 'db4', 'db5'], 1),
        (1, 64, 48, ['db2'], 3),
        (1, 128, 256, ['db1', 'db4', 'db5'], 3),
        (1, 120, 150, ['db3', 'self'], 2),
    ]
    out['ncases'] = len(cases)
    for c, (nband, nx, ny, bases, nlevel) in enumerate(cases):
        rng = np.random.default_rng(440 + c)
        psi = Psi(nband, nx, ny, bases, nlevel, 1)
        x = rng.standard_normal((nband, nx, ny))
        alpha = np.full((nband, len(bases), psi.Nymax, psi.Nxmax), np.nan)
        psi.dot(x, alpha)
        coef_in = rng.standard_normal(alpha.shape)
        xrec = np.full((nband, nx, ny), np.nan)
        psi.hdot(coef_in, xrec)
        out[f'p{c}_meta'] = np.array([nband, nx, ny, nlevel, psi.Nymax, psi.Nxmax])
        out[f'p{c}_bases'] = np.array(bases)
        out[f'p{c}_x'] = x
        out[f'p{c}_alpha'] = alpha           # NaN marks never-written cells
        out[f'p{c}_coef_in'] = coef_in
        out[f'p{c}_xrec'] = xrec
    np.savez_compressed(os.path.join(HERE, 'psi.npz'), **out)
    print('psi.npz', len(out))


def gen_prox():
    out = {}
    rng = np.random.default_rng(450)
    nband, nbasis, nymax, nxmax = 3, 2, 20, 24
    v = rng.standard_normal((nband, nbasis, nymax, nxmax))
    v[:, 0, 3, 4] = 0.0                       # exercise the "sum == 0" branch
    v[:, 1, 5, 6] = [1.0, -1.0, 0.0]          # sum exactly zero, entries not
    vp = rng.standard_normal(v.shape)
    w = rng.random((nbasis, nymax, nxmax))
    out.update(v=v, vp=vp, w=w)
    grid = [(1.0, 75.0), (1.0, 1.0), (1e-1, 1e-3), (1e-3, 1.0), (1e-1, 75.0), (1e-3, 1e-3)]
    out['grid'] = np.array(grid)
    for g, (lam, sigma) in enumerate(grid):
        res = np.full(v.shape, np.nan)
        prox_21m_numba(v, res, lam, sigma=sigma, weight=w)
        out[f'g{g}_prox21m_numba'] = res
        vv = v.copy()
        dual_update_numba(vp, vv, lam, sigma=sigma, weight=w)
        out[f'g{g}_dual_update_numba'] = vv
        out[f'g{g}_prox21m'] = prox_21m(v, lam, weight=w)
        out[f'g{g}_prox21'] = prox_21(v, lam, weight=w)
        # the band-l2-NORM variants of prox/prox_21.py (:23-48 prox_21_numba, :62-88 dual_update_numba): their arrays are
        # (nband, nbasis, ntot) with the coefficient plane flattened
        v3, vp3, w3 = v.reshape(nband, nbasis, -1), vp.reshape(nband, nbasis, -1), w.reshape(nbasis, -1)
        res = np.full(v3.shape, np.nan)
        prox_21_numba_l2(v3, res, lam, sigma=sigma, weight=w3)
        out[f'g{g}_prox21_numba'] = res
        vv = v3.copy()
        dual_update_numba_l2(vp3, vv, lam, sigma=sigma, weight=w3)
        out[f'g{g}_dual_update_numba_l2'] = vv
    np.savez_compressed(os.path.join(HERE, 'prox.npz'), **out)
    print('prox.npz', len(out))


def gen_pd():
    """primal_dual_optimised trajectory (primal_dual.py:91-180) with the spotless
    call-site wiring (workers/spotless.py:259-285)."""
    out = {}
    rng = np.random.default_rng(460)
    nb, nx, ny, P, Q = 2, 32, 32, 64, 64
    bases = ['self', 'db1', 'db2']
    nlevel = 2
    psfhat = psd_psfhat(rng, nb, P, Q)
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
    conv = partial(psf_convolve_cube, xpad, xhat, xout, psfhat, Q)
    truth = np.zeros((nb, nx, ny))
    truth[:, 10, 12] = 1.0
    truth[:, 20, 8] = 0.5
    truth[:, 15:18, 20:23] = 0.2
    residual = conv(truth).copy() + 1e-4 * rng.standard_normal(truth.shape)
    model = np.zeros_like(truth)
    psi = Psi(nb, nx, ny, bases, nlevel, 1)
    nbasis = len(bases)
    dual = np.zeros((nb, nbasis, psi.Nymax, psi.Nxmax))
    l1weight = np.ones((nbasis, psi.Nymax, psi.Nxmax))
    hessnorm = 1.05 * power_method(conv, (nb, nx, ny), b0=rng.standard_normal((nb, nx, ny)),
                                   tol=1e-4, maxit=100, verbosity=0)[0]
    data = residual + conv(model)
    data = data.copy()

    def grad21(x):
        return conv(x) - data
    lam = 1e-3
    out.update(psfhat=psfhat, residual=residual, hessnorm=hessnorm, lam=lam,
               bases=np.array(bases), nlevel=nlevel, Q=Q, data=data)
    for tag, pos, maxit in (('pos1_it10', 1, 10), ('pos0_it4', 0, 4), ('pos2_it6', 2, 6)):
        x_in = model.copy()
        v_in = dual.copy()
        x, v = primal_dual_optimised(x_in, v_in, lam, psi.hdot, psi.dot, hessnorm,
                                     None, l1weight, None, grad21, nu=nbasis,
                                     tol=0.0, maxit=maxit, positivity=pos,
                                     report_freq=100, gamma=1.0, verbosity=0)
        out[f'{tag}_x'] = x.copy()
        out[f'{tag}_v'] = v.copy()
    np.savez_compressed(os.path.join(HERE, 'pd.npz'), **out)
    print('pd.npz', len(out))


def gen_pdplain():
    """primal_dual (primal_dual.py:12-87), the un-optimised functional form: psiH / psi RETURN arrays, prox is a
    callable; operators built from the reference's own Psi and prox_21m."""
    out = {}
    rng = np.random.default_rng(461)
    nb, nx, ny, P, Q = 2, 32, 24, 64, 48
    bases = ['self', 'db1', 'db3']
    nlevel = 2
    psfhat = psd_psfhat(rng, nb, P, Q)
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
    conv = partial(psf_convolve_cube, xpad, xhat, xout, psfhat, Q)
    truth = np.zeros((nb, nx, ny))
    truth[:, 10, 12] = 1.0
    truth[:, 20, 8] = 0.5
    truth[:, 15:18, 16:19] = 0.2
    data = conv(truth).copy() + 1e-4 * rng.standard_normal(truth.shape)
    psiop = Psi(nb, nx, ny, bases, nlevel, 1)
    nbasis = len(bases)
    l1weight = 0.5 + rng.random((nbasis, psiop.Nymax, psiop.Nxmax))

    def psiH(x):                                  # analysis: image -> coefficients (new array)
        a = np.zeros((nb, nbasis, psiop.Nymax, psiop.Nxmax))
        psiop.dot(x, a)
        return a

    def psi(a):                                   # synthesis: coefficients -> image (new array)
        x = np.zeros((nb, nx, ny))
        psiop.hdot(a, x)
        return x

    def prox(v, sig):
        return prox_21m(v, sig, weight=l1weight)

    def grad(x):
        return conv(x) - data
    hessnorm = 1.05 * power_method(conv, (nb, nx, ny), b0=rng.standard_normal((nb, nx, ny)),
                                   tol=1e-4, maxit=100, verbosity=0)[0]
    lam = 2e-3
    out.update(psfhat=psfhat, data=data, hessnorm=hessnorm, lam=lam, bases=np.array(bases), nlevel=nlevel, Q=Q,
               l1weight=l1weight)
    for tag, pos, kw in (('pos1', 1, dict(tol=0.0, maxit=8, minit=2)), ('pos0', 0, dict(tol=0.0, maxit=5, minit=1)),
                         ('pos2', 2, dict(tol=0.0, maxit=6, minit=1)), ('tol', 1, dict(tol=5e-2, maxit=60, minit=3))):
        x, v = primal_dual(np.zeros((nb, nx, ny)), np.zeros((nb, nbasis, psiop.Nymax, psiop.Nxmax)), lam, psi, psiH,
                           hessnorm, prox, grad, nu=nbasis, positivity=pos, report_freq=1000, gamma=1.0, verbosity=0, **kw)
        out[f'{tag}_x'] = x.copy()
        out[f'{tag}_v'] = v.copy()
    np.savez_compressed(os.path.join(HERE, 'pdplain.npz'), **out)
    print('pdplain.npz', len(out))


class _Var:
    """Stand-in for an xarray variable: .values / .dtype / .shape is all hessian.py:161-221 touches."""
    def __init__(self, a):
        self.values, self.dtype, self.shape = a, a.dtype, a.shape


class _DS(dict):
    """Stand-in for the per-band xarray dataset (attribute access + `'MODEL' in ds`)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def gen_dist():
    """The per-band stateful operator class hessian_psf_slice (hessian.py:161-251) driven by
    pcg_dist (pcg.py:363-420) -- both only referenced from commented-out worker code, kept in the
    coverage table."""
    out = {}
    rng = np.random.default_rng(470)
    nx, ny, P, Q = 32, 32, 64, 64
    psfhat = psd_psfhat(rng, 1, P, Q)[0]
    psf = sfft.irfftn(psfhat, s=(P, Q))
    truth = np.zeros((nx, ny))
    truth[10, 12] = 1.0
    truth[20, 8] = 0.5
    xpad, xhat, xout = scratch(psfhat, Q, (nx, ny))
    beam = 0.7 + 0.3 * rng.random((nx, ny))
    wsumb, wsum = 3.0, 7.5
    dirty = wsum * (beam * psf_convolve_slice(xpad, xhat, xout, psfhat, Q, beam * truth)
                    + 1e-3 * rng.standard_normal((nx, ny)))
    for tag, extra in (('dirty', {}), ('resid', {'RESIDUAL': _Var(0.5 * dirty + 0.01 * rng.standard_normal((nx, ny))),
                                                  'MODEL': _Var(0.1 * truth)})):
        ds = _DS(DIRTY=_Var(dirty.copy()), PSFHAT=_Var(wsum * psfhat), PSF=_Var(wsum * psf),
                 BEAM=_Var(beam.copy()), WSUM=_Var(np.array([wsumb])), UVW=_Var(np.zeros((1, 3))),
                 WEIGHT=_Var(np.zeros((1, 1))), VIS_MASK=_Var(np.zeros((1, 1), dtype=np.uint8)),
                 FREQ=_Var(np.ones(1)), bandid=3, **extra)
        sigmainv = 1e-3
        A = hessian_psf_slice(ds, 2, 10, 1, sigmainv, 1.0, False, 1e-7, True)
        A.set_wsum(wsum)
        probe = rng.standard_normal((nx, ny))
        out[f'{tag}_Ax'] = A(probe).copy()
        out[f'{tag}_probe'] = probe
        for name, (maxit, minit, tol) in (('a', (30, 5, 1e-6)), ('b', (8, 8, 0.0))):
            out[f'{tag}_x_{name}'] = pcg_dist(A, maxit, minit, tol, sigmainv).copy()
        out[f'{tag}_model'] = A.model.copy()
        out[f'{tag}_residual'] = A.residual.copy()
        out[f'{tag}_dual_shape'] = np.array(A.dual.shape)
        if 'RESIDUAL' in extra:
            out['resid_in'] = extra['RESIDUAL'].values
            out['model_in'] = extra['MODEL'].values
    out.update(dirty=dirty, psfhat=wsum * psfhat, psf=wsum * psf, beam=beam, wsumb=wsumb, wsum=wsum,
               sigmainv=1e-3, Q=Q)
    np.savez_compressed(os.path.join(HERE, 'dist.npz'), **out)
    print('dist.npz', len(out))


def gen_dct():
    """cg_dct (pcg.py:139-239): plain CG over a nested dict {field: {'t..b..': image}}; here two
    fields of different size, two time/band keys each, A = per-leaf PSF Hessian + Tikhonov."""
    out = {}
    rng = np.random.default_rng(480)
    shapes = {'f0': (24, 20), 'f1': (16, 32)}
    keys = ['t0b0', 't0b1']
    sigmainv = 5e-2
    ops, b, x = {}, {}, {}
    for fld, (nx, ny) in shapes.items():
        ops[fld], b[fld], x[fld] = {}, {}, {}
        for i in keys:
            P, Q = 2 * nx, 2 * ny
            psfhat = psd_psfhat(rng, 1, P, Q)[0]
            out[f'{fld}_{i}_psfhat'] = psfhat
            ops[fld][i] = (psfhat, Q, scratch(psfhat, Q, (nx, ny)))
            b[fld][i] = rng.standard_normal((nx, ny))
            x[fld][i] = 0.1 * rng.standard_normal((nx, ny))
            out[f'{fld}_{i}_b'] = b[fld][i].copy()
            out[f'{fld}_{i}_x0'] = x[fld][i].copy()

    def A(v):
        res = {}
        for fld in v.keys():
            res[fld] = {}
            for i in v[fld].keys():
                psfhat, Q, (xpad, xhat, xout) = ops[fld][i]
                res[fld][i] = _hessian_psf_slice(xpad, xhat, xout, psfhat, None, Q, v[fld][i],
                                                 sigmainv=sigmainv)
        return res

    for tag, (tol, maxit) in (('it6', (0.0, 6)), ('tol', (1e-3, 200))):
        x0 = {fld: {i: x[fld][i].copy() for i in keys} for fld in shapes}
        xs, rs = cg_dct(A, b, x0, tol=tol, maxit=maxit, verbosity=0)
        for fld in shapes:
            for i in keys:
                out[f'{tag}_{fld}_{i}_x'] = xs[fld][i].copy()
                out[f'{tag}_{fld}_{i}_r'] = rs[fld][i].copy()
    out['sigmainv'] = sigmainv
    np.savez_compressed(os.path.join(HERE, 'dct.npz'), **out)
    print('dct.npz', len(out))


def gen_clark():
    """Clark CLEAN minor cycle (deconv/clark.py): the sub-minor loop on its own and the full loop with
    the PSF convolution, 2 bands, 32 x 32 image, 64 x 64 PSF (the reference's overlap mask is all true)."""
    out = {}
    rng = np.random.default_rng(490)
    nb, nx, ny, P, Q = 2, 32, 32, 64, 64
    u = (np.arange(P) - P // 2)[:, None]
    v = (np.arange(Q) - Q // 2)[None, :]
    wsums = np.array([0.6, 0.4])
    psf = np.stack([np.exp(-(u ** 2 + v ** 2) / (2 * (1.5 + 0.3 * b) ** 2))
                    * (1 + 0.15 * np.cos(0.7 * u) * np.cos(0.5 * v)) for b in range(nb)])
    psf *= (wsums / psf[:, P // 2, Q // 2])[:, None, None]            # peak = wsum per band
    psfhat = sfft.rfftn(sfft.ifftshift(psf, axes=(1, 2)), axes=(1, 2))
    truth = np.zeros((nb, nx, ny))
    truth[:, 8, 9] = [1.0, 0.9]
    truth[:, 20, 22] = [0.6, 0.7]
    truth[:, 21, 5] = [0.3, 0.25]
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))
    ID = psf_convolve_cube(xpad, xhat, xout, psfhat, Q, truth).copy() + 1e-3 * rng.standard_normal(truth.shape)
    out.update(ID=ID, PSF=psf, PSFHAT=psfhat, wsums=wsums)
    # sub-minor loop alone
    IRsearch = np.sum(ID, axis=0) ** 2
    subth = 0.3 * np.sqrt(IRsearch.max())
    Ip, Iq = np.where(IRsearch > subth ** 2)
    m = subminor(ID[:, Ip, Iq].copy(), psf, Ip, Iq, np.zeros_like(ID), wsums, gamma=0.1, th=subth, maxit=25)
    out.update(sub_Ip=Ip, sub_Iq=Iq, sub_th=subth, sub_model=m.copy())
    for tag, kw in (('a', dict(gamma=0.1, pf=0.05, maxit=6, subpf=0.5, submaxit=40)),
                    ('b', dict(gamma=0.05, pf=0.3, maxit=50, subpf=0.7, submaxit=1000, threshold=0.0))):
        model, status = clark(ID.copy(), psf, psfhat, wsums, verbosity=0, **kw)
        out[f'clark_{tag}_model'] = model.copy()
        out[f'clark_{tag}_status'] = status
    for tag, kw in (('a', dict(gamma=0.1, pf=0.1, maxit=10000)), ('b', dict(gamma=0.2, pf=0.01, maxit=37))):
        model, status = hogbom(ID.copy(), psf, verbosity=0, **kw)
        out[f'hogbom_{tag}_model'] = model.copy()
        out[f'hogbom_{tag}_status'] = status
    np.savez_compressed(os.path.join(HERE, 'clark.npz'), **out)
    print('clark.npz', len(out))


class _RefVar:
    """xarray.DataArray stand-in: what dds2cubes touches (misc.py:665-713)."""

    def __init__(self, a):
        self.data = self.values = a
        self.dtype, self.shape = a.dtype, a.shape


class _RefDS(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def gen_misc():
    """The real pfb/utils/misc.py (loaded by _refstubs.load_misc): norm_diff (:1316-1351),
    l1reweight_func (:1070-1080), dds2cubes (:664-739), freqmul and setup_parametrisation
    (:1366-1423)."""
    misc = _refstubs.load_misc()
    out = {}
    rng = np.random.default_rng(470)
    # norm_diff, 3-D and 2-D, both precisions
    for tag, dt in (('f64', np.float64), ('f32', np.float32)):
        x = rng.standard_normal((3, 10, 12)).astype(dt)
        xp = (x + 0.1 * rng.standard_normal(x.shape)).astype(dt)
        out[f'nd_{tag}_x'], out[f'nd_{tag}_xp'] = x, xp
        out[f'nd_{tag}_3d'] = np.float64(misc.norm_diff(x, xp))
        out[f'nd_{tag}_2d'] = np.float64(misc.norm_diff(x[1], xp[1]))
    out['nd_zero'] = np.float64(misc.norm_diff(np.zeros((4, 4)), np.zeros((4, 4))))
    # l1reweight_func through the reference's own Psi.dot
    nband, nx, ny, bases, nlevel = 2, 32, 24, ['self', 'db1', 'db3'], 2
    psi = Psi(nband, nx, ny, bases, nlevel, 1)
    model = rng.standard_normal((nband, nx, ny))
    model[np.abs(model) < 0.8] = 0.0
    outvar = np.zeros((nband, psi.nbasis, psi.Nymax, psi.Nxmax))
    rms_comps = 0.2 + rng.random((psi.nbasis, psi.Nymax, psi.Nxmax))
    out['rw_model'], out['rw_rms'] = model, rms_comps
    out['rw_meta'] = np.array([nband, nx, ny, nlevel])
    out['rw_bases'] = np.array(bases)
    for alpha in (4, 2):
        out[f'rw_a{alpha}'] = misc.l1reweight_func(psi.dot, outvar, 1.5, rms_comps, model, alpha=alpha)
    # dds2cubes: two datasets share band 0, one in band 1, band 2 stays empty
    nb, dx, dy = 3, 12, 10
    P, Q = 2 * dx, 2 * dy
    ins = {}
    for i, (bandid, w) in enumerate(((0, 2.0), (1, 3.5), (0, 1.5))):
        ins[i] = dict(DIRTY=rng.standard_normal((dx, dy)), BEAM=0.5 + rng.random((dx, dy)),
                      WSUM=np.array([w]), PSF=rng.standard_normal((P, Q)),
                      PSFHAT=rng.standard_normal((P, Q // 2 + 1)) + 1j * rng.standard_normal((P, Q // 2 + 1)),
                      MODEL=rng.standard_normal((dx, dy)), RESIDUAL=rng.standard_normal((dx, dy)),
                      DUAL=rng.standard_normal((2, 14, 13)), bandid=bandid)
        for k, v in ins[i].items():
            out[f'dds{i}_{k}'] = np.asarray(v)

    def build(skip=()):
        return [_RefDS({k: (_RefVar(v.copy()) if k != 'bandid' else v) for k, v in ins[i].items()
                        if k not in skip}) for i in range(3)]
    names = ('dirty', 'model', 'residual', 'psf', 'psfhat', 'mean_beam', 'wsums', 'dual')
    for tag, kw, skip in (('beam', dict(apparent=False), ()), ('app', dict(apparent=True), ()),
                          ('bare', dict(apparent=False), ('RESIDUAL', 'DUAL', 'PSF', 'PSFHAT')),
                          ('nodual', dict(apparent=False, dual=False), ())):
        res = misc.dds2cubes(build(skip), nb, **kw)
        for n, r in zip(names, res):
            if r is not None:
                out[f'cubes_{tag}_{n}'] = np.asarray(r)
    # freqmul + setup_parametrisation
    A = rng.standard_normal((4, 4))
    x = rng.standard_normal((4, 9, 7))
    out['fm_A'], out['fm_x'], out['fm_out'] = A, x, misc.freqmul(A, x)
    freq = np.linspace(0.9e9, 1.7e9, 4)
    out['par_freq'] = freq
    x0 = 0.3 * rng.standard_normal((4, 9, 7))
    v = rng.standard_normal((4, 9, 7))
    out['par_x0'], out['par_v'] = x0, v
    for mode in ('id', 'exp'):
        func, finv, dfunc, dhfunc = misc.setup_parametrisation(mode=mode, minval=1e-5, sigma=0.8,
                                                               freq=freq, lscale=0.5)
        out[f'par_{mode}_func'] = func(x0)
        out[f'par_{mode}_finv'] = finv(func(x0))
        out[f'par_{mode}_dfunc'] = dfunc(x0, v)
        out[f'par_{mode}_dhfunc'] = dhfunc(x0, v)
    np.savez_compressed(os.path.join(HERE, 'misc.npz'), **out)
    print('misc.npz', len(out), 'arrays')


def gen_cg():
    """Plain CG (pcg.py:12-50: no preconditioner, eps = <r, r> un-normalised) on the PSD PSF Hessian + Tikhonov
    of one 2-band cube: fixed iteration counts, a tolerance stop and a warm start."""
    out = {}
    rng = np.random.default_rng(480)
    nb, nx, ny, P, Q = 2, 40, 48, 80, 96
    psfhat = psd_psfhat(rng, nb, P, Q)
    sigmainv = 0.05
    b = rng.standard_normal((nb, nx, ny))
    x0 = 0.1 * rng.standard_normal((nb, nx, ny))
    xpad, xhat, xout = scratch(psfhat, Q, (nb, nx, ny))

    def A(v):
        return psf_convolve_cube(xpad, xhat, xout, psfhat, Q, v, nthreads=1).copy() + sigmainv * v
    out.update(psfhat=psfhat, b=b, x0=x0, sigmainv=np.float64(sigmainv), Q=np.int64(Q))
    for k in (1, 4, 12):
        out[f'k{k}'] = cg(A, b, None, tol=0.0, maxit=k, verbosity=0)
    out['tol'] = cg(A, b, None, tol=1e-6, maxit=500, verbosity=0)
    out['warm_k5'] = cg(A, b, x0, tol=0.0, maxit=5, verbosity=0)
    np.savez_compressed(os.path.join(HERE, 'cg.npz'), **out)
    print('cg.npz', len(out))


if __name__ == '__main__':
    which = sys.argv[1:] or ['conv', 'pcg', 'psi', 'prox', 'pd', 'pdplain', 'dist', 'dct', 'clark', 'misc', 'cg']
    for w in which:
        globals()['gen_' + w]()
