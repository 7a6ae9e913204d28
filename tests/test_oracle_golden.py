"""
Pins the CPU oracle (oracle/) against golden vectors produced by the reference's own
source (tests/golden/make_golden.py) and against known-answer properties
(SURVEY.md 8c).  CPU only.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import fftconv as fc
from oracle import solvers as sv
from oracle import prox as px
from oracle import wavelets as wv
from oracle import daubechies as db

pmp = pytest.mark.parametrize


# ------------------------------------------------------------------ filters
@pmp('K', range(1, 10))
def test_daubechies_orthonormal(K):
    h = db.rec_lo(K)
    F = h.size
    assert F == 2 * K
    for m in range(K):
        s = np.dot(h[:F - 2 * m], h[2 * m:])
        assert abs(s - (1.0 if m == 0 else 0.0)) <= 1e-15
    assert abs(h.sum() - np.sqrt(2)) <= 5e-16
    # K vanishing moments of the high-pass
    dl, dh, rl, rh = db.filter_bank(f'db{K}')
    n = np.arange(F)
    for q in range(K):
        assert abs(np.sum(dh * n ** q)) < 1e-8 * max(1, F ** q)


def test_db2_closed_form():
    s3 = np.sqrt(3)
    assert_allclose(db.rec_lo(2), np.array([1 + s3, 3 + s3, 3 - s3, 1 - s3]) / (4 * np.sqrt(2)),
                    rtol=0, atol=3e-16)


# --------------------------------------------------------------------- conv
def _scratch(psfhat, Q, shape):
    return fc.make_scratch(psfhat, Q, shape, np.float64)


@pmp('c', range(6))
def test_conv_and_hessian_match_reference(golden, c):
    g = golden('conv')
    nx, ny, P, Q = g['cases'][c]
    psfhat, x, beam = g[f'c{c}_psfhat'], g[f'c{c}_x'], g[f'c{c}_beam']
    assert_allclose(fc.psfhat_from_psf(g[f'c{c}_psf']), psfhat, rtol=0, atol=1e-12)
    xpad, xhat, xout = _scratch(psfhat[0], Q, (nx, ny))
    y = fc.psf_convolve_slice(xpad, xhat, xout, psfhat[0], Q, x[0])
    assert y is xout                                        # aliasing contract
    assert_allclose(y, g[f'c{c}_slice'], rtol=0, atol=1e-12)
    xpad, xhat, xout = _scratch(psfhat, Q, x.shape)
    assert_allclose(fc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, x),
                    g[f'c{c}_cube'], rtol=0, atol=1e-12)
    xpad, xhat, xout = _scratch(psfhat[1], Q, (nx, ny))
    assert_allclose(fc._hessian_psf_slice(xpad, xhat, xout, psfhat[1], beam[1], Q, x[1],
                                          sigmainv=0.37, wsum=2.5),
                    g[f'c{c}_h_slice_full'], rtol=0, atol=1e-12)
    assert_allclose(fc._hessian_psf_slice(xpad, xhat, xout, psfhat[1], None, Q, x[1],
                                          sigmainv=0.0),
                    g[f'c{c}_h_slice_bare'], rtol=0, atol=1e-12)
    xpad, xhat, xout = _scratch(psfhat, Q, x.shape)
    assert_allclose(fc.hessian_psf_cube(xpad, xhat, xout, beam, psfhat, Q, x,
                                        sigmainv=1.25, wsum=3.0),
                    g[f'c{c}_h_cube_full'], rtol=0, atol=1e-12)
    assert_allclose(fc.hessian_psf_cube(xpad, xhat, xout, None, psfhat, Q, x,
                                        sigmainv=0.5),
                    g[f'c{c}_h_cube_bare'], rtol=0, atol=1e-12)


def test_conv_is_direct_circular_convolution():
    """Pins the DFT boundary by definition, incl. the
    aliasing case nx_psf < 2 nx."""
    rng = np.random.default_rng(1)
    for (nx, ny, P, Q) in [(6, 5, 12, 10), (6, 5, 8, 6)]:
        psf = rng.standard_normal((P, Q))
        x = rng.standard_normal((nx, ny))
        psfhat = fc.psfhat_from_psf(psf)
        xpad, xhat, xout = _scratch(psfhat, Q, (nx, ny))
        y = fc.psf_convolve_slice(xpad, xhat, xout, psfhat, Q, x)
        assert_allclose(y, fc.direct_circular_convolution(psf, x), rtol=0, atol=1e-13)


# ---------------------------------------------------------------------- pcg
def test_pcg_band_history(golden):
    g = golden('pcg')
    psfhat, b, beam = g['psfhat'], g['b'], g['beam']
    sigmainv, Q = float(g['sigmainv']), int(g['Q'])
    for tag, bm in (('nobeam', None), ('beam', beam)):
        for k in (1, 2, 5, 20):
            for bt in (True, False):
                m = sv.pcg_psf(psfhat, b, np.zeros_like(b), bm, Q, 1, sigmainv,
                               dict(tol=0.0, maxit=k, minit=k, verbosity=0, backtrack=bt))
                ref = g[f'band_{tag}_k{k}_bt{int(bt)}']
                assert_allclose(m, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())
    for key, minit in (('band_tol1e-2', 1), ('band_tol1e-2_minit15', 15)):
        m = sv.pcg_psf(psfhat, b, np.zeros_like(b), None, Q, 1, sigmainv,
                       dict(tol=1e-2, maxit=100, minit=minit, verbosity=0, backtrack=True))
        assert_allclose(m, g[key], rtol=1e-9, atol=1e-11 * np.abs(g[key]).max())


def test_pcg_cube_and_edge_cases(golden):
    g = golden('pcg')
    psfhat, b, beam = g['psfhat'], g['b'], g['beam']
    sigmainv, Q = float(g['sigmainv']), int(g['Q'])
    xpad, xhat, xout = _scratch(psfhat, Q, b.shape)

    def A(v):
        return fc.hessian_psf_cube(xpad, xhat, xout, beam, psfhat, Q, v,
                                   sigmainv=sigmainv, wsum=1.0)
    for k in (1, 3, 10, 25):
        x, r = sv.pcg(A, beam * b, np.zeros_like(b), tol=0.0, maxit=k, minit=k,
                      verbosity=0, return_resid=True)
        assert_allclose(x, g[f'cube_k{k}_x'], rtol=1e-9, atol=1e-11)
        assert_allclose(r, g[f'cube_k{k}_r'], rtol=1e-8, atol=1e-11)
    x0 = g['cube_x0']
    x = sv.pcg(A, beam * b, x0, M=lambda v: v / sigmainv, tol=0.0, maxit=7, minit=7,
               verbosity=0)
    assert_allclose(x, g['cube_M_x0_k7'], rtol=1e-9, atol=1e-11)
    assert bool(g['cube_zero_resid_is_x0'])
    xz = sv.pcg(A, A(x0).copy(), x0, tol=1e-5, maxit=5, minit=1, verbosity=0)
    assert xz is x0


def test_pcg_backtracking_branch(golden):
    """Indefinite operator: the backtracking loop is taken (pcg.py:96-101)."""
    g = golden('pcg')
    ph, bb = g['indef_psfhat'], g['indef_b']
    sigmainv, Q = float(g['sigmainv']), int(g['Q'])
    xpad, xhat, xout = _scratch(ph, Q, bb.shape)

    def A(v):
        return fc._hessian_psf_slice(xpad, xhat, xout, ph, None, Q, v, sigmainv=sigmainv)
    ntaken = 0
    for bt in (True, False):
        for k in (3, 8):
            tr = sv.PCGTrace()
            x = sv.pcg(A, bb, None, tol=0.0, maxit=k, minit=k, verbosity=0, backtrack=bt,
                       trace=tr)
            ref = g[f'indef_k{k}_bt{int(bt)}']
            assert_allclose(x, ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())
            ntaken += sum(tr.nbacktrack)
    assert ntaken > 0
    assert not np.allclose(g['indef_k8_bt1'], g['indef_k8_bt0'])


def test_power_method(golden):
    g = golden('pcg')
    psfhat, Q = g['psfhat'], int(g['Q'])
    b0 = g['pm_b0']
    xpad, xhat, xout = _scratch(psfhat, Q, b0.shape)

    def conv(v):
        return fc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, v)
    beta, bvec = sv.power_method(conv, b0.shape, b0=b0.copy(), tol=1e-3, maxit=40)
    assert_allclose(beta, g['pm_beta'], rtol=1e-12)
    assert_allclose(bvec, g['pm_b'], rtol=0, atol=1e-12)


def test_norm_diff_contract():
    rng = np.random.default_rng(3)
    x, xp = rng.standard_normal((2, 4, 5)), rng.standard_normal((2, 4, 5))
    assert_allclose(sv.norm_diff(x, xp),
                    np.sqrt(np.sum((x - xp) ** 2) / (1e-12 + np.sum(x ** 2))), rtol=1e-15)
    with pytest.raises(ValueError):
        sv.norm_diff(x[0, 0], xp[0, 0])
    assert sv.norm_diff(np.zeros((3, 3)), np.zeros((3, 3))) == 0.0


# ------------------------------------------------------------------ wavelets
def test_psi_matches_reference(golden):
    g = golden('psi')
    for c in range(int(g['ncases'])):
        nband, nx, ny, nlevel, Nymax, Nxmax = g[f'p{c}_meta']
        bases = [str(s) for s in g[f'p{c}_bases']]
        psi = wv.Psi(nband, nx, ny, bases, nlevel, 1)
        assert (psi.Nymax, psi.Nxmax) == (Nymax, Nxmax)
        ref = g[f'p{c}_alpha']
        alpha = np.full(ref.shape, np.nan)
        psi.dot(g[f'p{c}_x'], alpha)
        # identical never-written cells, identical values elsewhere
        assert np.array_equal(np.isnan(alpha), np.isnan(ref))
        m = ~np.isnan(ref)
        assert_allclose(alpha[m], ref[m], rtol=0, atol=1e-13)
        xrec = np.full((nband, nx, ny), np.nan)
        psi.hdot(g[f'p{c}_coef_in'], xrec)
        assert_allclose(xrec, g[f'p{c}_xrec'], rtol=0, atol=1e-12)


@pmp("nx", [128, 250])
@pmp("ny", [64, 78])
@pmp("nband", [1, 3])
@pmp("nlevels", [1, 2])
def test_psi_perfect_reconstruction(nx, ny, nband, nlevels):
    """reference tests/test_psi_operator.py:14-48 (garbage-initialised outputs)."""
    rng = np.random.default_rng(420)
    image = rng.standard_normal((nx, ny))
    nu = 1.0 + 0.1 * np.arange(nband)
    x = image[None] * nu[:, None, None] ** (-0.7)
    bases = ['self', 'db1', 'db2', 'db3', 'db4', 'db5']
    psi = wv.Psi(nband, nx, ny, bases, nlevels, 1)
    alpha = rng.standard_normal((nband, len(bases), psi.Nymax, psi.Nxmax))
    xrec = rng.standard_normal((nband, nx, ny))
    psi.dot(x, alpha)
    psi.hdot(alpha, xrec)
    assert_allclose(len(bases) * x, xrec, rtol=0, atol=1e-12)


@pmp("wavelet", ["db1", "db4", "db5"])
@pmp("shape", [(128, 256), (512, 128)])
@pmp("nlevel", [1, 2, 3])
def test_dwt_idwt_roundtrip(wavelet, shape, nlevel):
    """reference tests/test_wavelets.py:11-82 (the pywt layout half of that test is
    covered by the golden comparison above: pywt is not installed)."""
    rng = np.random.default_rng(5)
    nx, ny = shape
    data = rng.random(shape)
    F = 2 * int(wavelet[-1])
    bk = wv.Bookkeeping(nx, ny, F, nlevel)
    dl, dh, rl, rh = db.filter_bank(wavelet)
    coeffs = np.zeros((bk.Ntoty, bk.Ntotx))
    wv.dwt2d(data, coeffs, bk, dl, dh)
    rec = np.zeros(shape)
    wv.idwt2d(coeffs, rec, bk, rl, rh)
    assert_allclose(rec, data, rtol=0, atol=1e-12)


def test_psi_adjoint():
    rng = np.random.default_rng(9)
    nband, nx, ny = 2, 64, 48
    bases = ['self', 'db2', 'db4']
    psi = wv.Psi(nband, nx, ny, bases, 2, 1)
    x = rng.standard_normal((nband, nx, ny))
    a = np.zeros((nband, 3, psi.Nymax, psi.Nxmax))
    psi.dot(x, a)
    c = rng.standard_normal(a.shape)
    written = np.full(a.shape, np.nan)
    psi.dot(x, written)
    c[np.isnan(written)] = 0.0
    y = np.zeros_like(x)
    psi.hdot(c, y)
    assert_allclose(np.vdot(a, c), np.vdot(x, y), rtol=1e-12)


def test_psi_level_error():
    with pytest.raises(ValueError):
        wv.Psi(1, 16, 16, ['db5'], 3, 1)


# --------------------------------------------------------------------- prox
def test_prox_matches_reference(golden):
    g = golden('prox')
    v, vp, w = g['v'], g['vp'], g['w']
    for i, (lam, sigma) in enumerate(g['grid']):
        res = np.full(v.shape, np.nan)
        px.prox_21m_numba(v, res, lam, sigma=sigma, weight=w)
        assert_allclose(res, g[f'g{i}_prox21m_numba'], rtol=1e-13, atol=1e-15)
        vv = v.copy()
        px.dual_update_numba(vp, vv, lam, sigma=sigma, weight=w)
        assert_allclose(vv, g[f'g{i}_dual_update_numba'], rtol=1e-13, atol=1e-14)
        assert_allclose(px.prox_21m(v, lam, weight=w), g[f'g{i}_prox21m'], rtol=1e-13, atol=1e-15)
        assert_allclose(px.prox_21(v, lam, weight=w), g[f'g{i}_prox21'], rtol=1e-13, atol=1e-15)
        # the band-l2-norm numba variants of prox/prox_21.py (arrays with the coefficient plane flattened)
        nb_, nbas = v.shape[:2]
        v3, vp3, w3 = v.reshape(nb_, nbas, -1), vp.reshape(nb_, nbas, -1), w.reshape(nbas, -1)
        r3 = np.full(v3.shape, np.nan)
        px.prox_21_numba(v3, r3, lam, sigma=sigma, weight=w3)
        assert_allclose(r3, g[f'g{i}_prox21_numba'], rtol=1e-13, atol=1e-15)
        vv3 = v3.copy()
        px.dual_update_numba_l2(vp3, vv3, lam, sigma=sigma, weight=w3)
        assert_allclose(vv3, g[f'g{i}_dual_update_numba_l2'], rtol=1e-13, atol=1e-14)
        # reference tests/test_psi_operator.py:126-147
        assert_allclose(px.prox_21m(v / sigma, lam / sigma, weight=w), res, atol=1e-8)


def test_dual_update_identity():
    """reference tests/test_psi_operator.py:150-193: dual_update_numba ==
    vtilde - sigma*prox_21m(vtilde/sigma, lam/sigma)."""
    rng = np.random.default_rng(11)
    nband, nx, ny = 3, 120, 150
    bases = ['self', 'db1', 'db3']
    psi = wv.Psi(nband, nx, ny, bases, 2, 1)
    w = rng.random((3, psi.Nymax, psi.Nxmax))
    x = rng.standard_normal((nband, nx, ny))
    for lam, sigma in [(1.0, 75.0), (1e-1, 1.0), (1e-3, 1e-3)]:
        v = np.zeros((nband, 3, psi.Nymax, psi.Nxmax))
        psi.dot(rng.standard_normal(x.shape), v)
        vp = v.copy()
        res1 = px.dual_update(v, x, psi.dot, lam, sigma=sigma, weight=w)
        psi.dot(x, v)
        px.dual_update_numba(vp, v, lam, sigma=sigma, weight=w)
        assert_allclose(1 + res1, 1 + v, rtol=0, atol=1e-9)


# -------------------------------------------------------------- primal dual
def test_primal_dual_trajectory(golden):
    g = golden('pd')
    psfhat, Q = g['psfhat'], int(g['Q'])
    nb, P, _ = psfhat.shape
    nx = ny = P // 2
    bases = [str(s) for s in g['bases']]
    psi = wv.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    xpad, xhat, xout = _scratch(psfhat, Q, (nb, nx, ny))
    data = g['data']

    def grad21(x):
        return fc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, x) - data
    l1w = np.ones((len(bases), psi.Nymax, psi.Nxmax))
    for tag, pos, maxit in (('pos1_it10', 1, 10), ('pos0_it4', 0, 4), ('pos2_it6', 2, 6)):
        x0 = np.zeros((nb, nx, ny))
        v0 = np.zeros((nb, len(bases), psi.Nymax, psi.Nxmax))
        x, v = sv.primal_dual_optimised(x0, v0, float(g['lam']), psi.hdot, psi.dot,
                                        float(g['hessnorm']), None, l1w, None, grad21,
                                        nu=len(bases), tol=0.0, maxit=maxit, positivity=pos)
        assert_allclose(x, g[f'{tag}_x'], rtol=1e-9, atol=1e-12)
        assert_allclose(v, g[f'{tag}_v'], rtol=1e-9, atol=1e-12)


# ------------------------------------------- per-band stateful operator + pcg_dist (a5, a8)
class _Var:
    def __init__(self, a):
        self.values, self.dtype, self.shape = a, a.dtype, a.shape


class _DS(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def dist_dataset(g, tag, wrap=lambda a: a):
    extra = {}
    if tag == 'resid':
        extra = {'RESIDUAL': _Var(wrap(g['resid_in'])), 'MODEL': _Var(wrap(g['model_in']))}
    return _DS(DIRTY=_Var(wrap(g['dirty'])), PSFHAT=_Var(wrap(g['psfhat'])), PSF=_Var(wrap(g['psf'])),
               BEAM=_Var(wrap(g['beam'])), WSUM=_Var(np.array([float(g['wsumb'])])), bandid=3, **extra)


@pytest.mark.parametrize('tag', ['dirty', 'resid'])
def test_hessian_psf_slice_class_and_pcg_dist(golden, tag):
    """hessian.py:161-251 + pcg.py:363-420 against the reference's own outputs."""
    g = golden('dist')
    A = fc.hessian_psf_slice(dist_dataset(g, tag), 2, 10, 1, float(g['sigmainv']))
    A.set_wsum(float(g['wsum']))
    assert_allclose(A(g[f'{tag}_probe']), g[f'{tag}_Ax'], rtol=1e-11, atol=1e-13)
    assert_allclose(A.model, g[f'{tag}_model'])
    assert_allclose(A.residual, g[f'{tag}_residual'])
    assert tuple(A.dual.shape) == tuple(g[f'{tag}_dual_shape'])
    for name, (maxit, minit, tol) in (('a', (30, 5, 1e-6)), ('b', (8, 8, 0.0))):
        x = sv.pcg_dist(A, maxit, minit, tol, float(g['sigmainv']))
        assert_allclose(x, g[f'{tag}_x_{name}'], rtol=1e-8, atol=1e-11)


def make_dds(rng, nband=3, nx=12, ny=10, with_resid=True, with_dual=True, wrap=lambda a: a):
    """Three datasets: two share band 0, one sits in band 1, band 2 stays empty."""
    dds = []
    for bandid, w in ((0, 2.0), (1, 3.5), (0, 1.5)):
        P, Q = 2 * nx, 2 * ny
        d = dict(DIRTY=_Var(wrap(rng.standard_normal((nx, ny)))), BEAM=_Var(wrap(0.5 + rng.random((nx, ny)))),
                 WSUM=_Var(np.array([w])), PSF=_Var(wrap(rng.standard_normal((P, Q)))),
                 PSFHAT=_Var(wrap(rng.standard_normal((P, Q // 2 + 1)) + 1j * rng.standard_normal((P, Q // 2 + 1)))),
                 MODEL=_Var(wrap(rng.standard_normal((nx, ny)))), bandid=bandid)
        if with_resid:
            d['RESIDUAL'] = _Var(wrap(rng.standard_normal((nx, ny))))
        if with_dual:
            d['DUAL'] = _Var(wrap(rng.standard_normal((2, 14, 13))))
        dds.append(_DS(**d))
    return dds


@pytest.mark.parametrize('apparent', [False, True])
def test_dds2cubes_oracle_against_definition(apparent):
    """misc.py:664-739 against the formulas written out by hand (the reference's own outputs:
    test_dds2cubes_pinned below)."""
    rng = np.random.default_rng(5)
    dds = make_dds(rng)
    dirty, model, resid, psf, psfhat, mbeam, wsums, dual = sv.dds2cubes(dds, 3, apparent=apparent)
    a, b, c = dds
    wsum = 2.0 + 3.5 + 1.5
    wgt = (lambda ds: 1.0) if apparent else (lambda ds: ds.BEAM.values)
    assert_allclose(wsums, [3.5, 3.5, 0.0])
    assert_allclose(dirty[0], (a.DIRTY.values * wgt(a) + c.DIRTY.values * wgt(c)) / wsum)
    assert_allclose(dirty[1], b.DIRTY.values * wgt(b) / wsum)
    assert_allclose(resid[0], (a.RESIDUAL.values * wgt(a) + c.RESIDUAL.values * wgt(c)) / wsum)
    assert_allclose(psf[0], (a.PSF.values + c.PSF.values) / wsum)
    assert_allclose(psfhat[1], b.PSFHAT.values / wsum)
    assert_allclose(model[0], c.MODEL.values)               # the band's LAST dataset wins
    assert_allclose(dual[0], c.DUAL.values)
    assert_allclose(mbeam[0], (a.BEAM.values * 2.0 + c.BEAM.values * 1.5) / 3.5)
    assert not dirty[2].any() and not mbeam[2].any()
    out = sv.dds2cubes(make_dds(rng, with_resid=False, with_dual=False), 3, dual=True)
    assert out[2] is None and out[7] is None


def dct_problem(g, wrap=lambda a: a):
    shapes = {'f0': (24, 20), 'f1': (16, 32)}
    keys = ['t0b0', 't0b1']
    b = {f: {i: wrap(g[f'{f}_{i}_b']) for i in keys} for f in shapes}
    x0 = {f: {i: wrap(g[f'{f}_{i}_x0'].copy()) for i in keys} for f in shapes}
    return shapes, keys, b, x0


def test_cg_dct_golden(golden):
    """pcg.py:139-239 against the reference's own iterates (tests/golden/dct.npz)."""
    g = golden('dct')
    shapes, keys, b, _ = dct_problem(g)
    sigmainv = float(g['sigmainv'])
    scr = {f: {i: _scratch(g[f'{f}_{i}_psfhat'], 2 * shapes[f][1], shapes[f]) for i in keys} for f in shapes}

    def A(v):
        return {f: {i: fc._hessian_psf_slice(*scr[f][i], g[f'{f}_{i}_psfhat'], None, 2 * shapes[f][1], v[f][i],
                                             sigmainv=sigmainv) for i in v[f]} for f in v}
    for tag, (tol, maxit) in (('it6', (0.0, 6)), ('tol', (1e-3, 200))):
        x0 = dct_problem(g)[3]
        xs, rs = sv.cg_dct(A, b, x0, tol=tol, maxit=maxit, verbosity=0)
        assert xs is x0
        for f in shapes:
            for i in keys:
                assert_allclose(xs[f][i], g[f'{tag}_{f}_{i}_x'], rtol=1e-8, atol=1e-10)
                assert_allclose(rs[f][i], g[f'{tag}_{f}_{i}_r'], rtol=1e-7, atol=1e-9)


# ----------------------------------------------------------------- Clark CLEAN (8f3)
def test_clark_golden(golden):
    """deconv/clark.py:29-177 against the reference's own outputs (tests/golden/clark.npz)."""
    from oracle import clark as ock
    g = golden('clark')
    ID, PSF, PSFHAT, wsums = g['ID'], g['PSF'], g['PSFHAT'], g['wsums']
    Ip, Iq = g['sub_Ip'], g['sub_Iq']
    m, k = ock.subminor(ID[:, Ip, Iq].copy(), PSF, Ip, Iq, np.zeros_like(ID), wsums, gamma=0.1,
                        th=float(g['sub_th']), maxit=25)
    assert k > 3
    assert_allclose(m, g['sub_model'], rtol=1e-12, atol=1e-15)
    for tag, kw in (('a', dict(gamma=0.1, pf=0.05, maxit=6, subpf=0.5, submaxit=40)),
                    ('b', dict(gamma=0.05, pf=0.3, maxit=50, subpf=0.7, submaxit=1000, threshold=0.0))):
        model, status = ock.clark(ID.copy(), PSF, PSFHAT, wsums, **kw)
        assert status == int(g[f'clark_{tag}_status'])
        assert_allclose(model, g[f'clark_{tag}_model'], rtol=1e-9, atol=1e-13)


def test_hogbom_golden(golden):
    """deconv/hogbom.py:8-74 against the reference's own outputs (tests/golden/clark.npz)."""
    from oracle import clark as ock
    g = golden('clark')
    for tag, kw in (('a', dict(gamma=0.1, pf=0.1, maxit=10000)), ('b', dict(gamma=0.2, pf=0.01, maxit=37))):
        model, status, _ = ock.hogbom(g['ID'].copy(), g['PSF'], **kw)
        assert status == int(g[f'hogbom_{tag}_status'])
        assert_allclose(model, g[f'hogbom_{tag}_model'], rtol=1e-11, atol=1e-14)


# ------------------------------------------------------- misc.py (tests/golden/misc.npz)
def misc_dds(g, wrap=lambda a: a, skip=()):
    """The three datasets of golden/misc.npz as in-memory datasets (two share band 0)."""
    dds = []
    for i in range(3):
        d = {k: _Var(wrap(np.array(g[f'dds{i}_{k}']))) for k in
             ('DIRTY', 'BEAM', 'PSF', 'PSFHAT', 'MODEL', 'RESIDUAL', 'DUAL') if k not in skip}
        d['WSUM'] = _Var(np.array(g[f'dds{i}_WSUM']))
        dds.append(_DS(bandid=int(g[f'dds{i}_bandid']), **d))
    return dds


CUBE_NAMES = ('dirty', 'model', 'residual', 'psf', 'psfhat', 'mean_beam', 'wsums', 'dual')
CUBE_CASES = (('beam', dict(apparent=False), ()), ('app', dict(apparent=True), ()),
              ('bare', dict(apparent=False), ('RESIDUAL', 'DUAL', 'PSF', 'PSFHAT')),
              ('nodual', dict(apparent=False, dual=False), ()))


def test_norm_diff_pinned(golden):
    """misc.py:1316-1351 run through the reference's own numba overload body."""
    g = golden('misc')
    for tag in ('f64', 'f32'):
        x, xp = g[f'nd_{tag}_x'], g[f'nd_{tag}_xp']
        assert_allclose(sv.norm_diff(x, xp), g[f'nd_{tag}_3d'], rtol=1e-13 if tag == 'f64' else 1e-6)
        assert_allclose(sv.norm_diff(x[1], xp[1]), g[f'nd_{tag}_2d'], rtol=1e-13 if tag == 'f64' else 1e-6)
    assert g['nd_zero'] == 0.0


def test_l1reweight_pinned(golden):
    """misc.py:1070-1080 with the reference's Psi.dot as psiH."""
    g = golden('misc')
    nband, nx, ny, nlevel = (int(v) for v in g['rw_meta'])
    psi = wv.Psi(nband, nx, ny, [str(b) for b in g['rw_bases']], nlevel)
    outvar = np.zeros((nband, psi.nbasis, psi.Nymax, psi.Nxmax))
    for alpha in (4, 2):
        got = sv.l1reweight_func(psi.dot, outvar, 1.5, g['rw_rms'], g['rw_model'], alpha=alpha)
        assert_allclose(got, g[f'rw_a{alpha}'], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize('case', CUBE_CASES, ids=[c[0] for c in CUBE_CASES])
def test_dds2cubes_pinned(golden, case):
    """misc.py:664-739 (the reference's lines run eagerly: dask.array zeros/stack/compute mapped onto
    numpy by tests/golden/_refstubs.py)."""
    g = golden('misc')
    tag, kw, skip = case
    got = sv.dds2cubes(misc_dds(g, skip=skip), 3, **kw)
    for n, r in zip(CUBE_NAMES, got):
        key = f'cubes_{tag}_{n}'
        if key in g.files:
            assert r.shape == g[key].shape and r.dtype == g[key].dtype
            assert_allclose(r, g[key], rtol=1e-14, atol=0)
        else:
            assert r is None


def test_freqmul_parametrisation_pinned(golden):
    """misc.py:1366-1423."""
    g = golden('misc')
    assert_allclose(sv.freqmul(g['fm_A'], g['fm_x']), g['fm_out'], rtol=1e-13, atol=1e-14)
    for mode in ('id', 'exp'):
        func, finv, dfunc, dhfunc = sv.setup_parametrisation(mode=mode, minval=1e-5, sigma=0.8,
                                                            freq=g['par_freq'], lscale=0.5)
        x0, v = g['par_x0'], g['par_v']
        assert_allclose(func(x0), g[f'par_{mode}_func'], rtol=1e-12)
        assert_allclose(finv(func(x0)), g[f'par_{mode}_finv'], rtol=1e-7, atol=1e-9)
        assert_allclose(dfunc(x0, v), g[f'par_{mode}_dfunc'], rtol=1e-12, atol=1e-13)
        assert_allclose(dhfunc(x0, v), g[f'par_{mode}_dhfunc'], rtol=1e-12, atol=1e-13)


def test_plain_cg_pinned(golden):
    """pcg.py:12-50 (cg): fixed iteration counts, tolerance stop, warm start -- the reference's outputs."""
    g = golden('cg')
    psfhat, b, Q, sigmainv = g['psfhat'], g['b'], int(g['Q']), float(g['sigmainv'])
    xpad, xhat, xout = fc.make_scratch(psfhat, Q, b.shape, np.float64)

    def A(v):
        return fc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, v).copy() + sigmainv * v
    for k in (1, 4, 12):
        assert_allclose(sv.cg(A, b, None, tol=0.0, maxit=k, verbosity=0), g[f'k{k}'], rtol=1e-10, atol=1e-12)
    assert_allclose(sv.cg(A, b, None, tol=1e-6, maxit=500, verbosity=0), g['tol'], rtol=1e-7, atol=1e-9)
    assert_allclose(sv.cg(A, b, g['x0'], tol=0.0, maxit=5, verbosity=0), g['warm_k5'], rtol=1e-10, atol=1e-12)


def test_primal_dual_unoptimised_golden(golden):
    """oracle/solvers.primal_dual (primal_dual.py:12-87, the functional form) against trajectories the reference's
    own function produced (tests/golden/pdplain.npz): positivity 0 / 1 / 2 with minit forcing, and a live tolerance."""
    from oracle import solvers as osv, fftconv as ofc, wavelets as owv, prox as opx
    g = golden('pdplain')
    psfhat, Q = g['psfhat'], int(g['Q'])
    nb, P, _ = psfhat.shape
    nx, ny = P // 2, Q // 2
    bases = [str(b) for b in g['bases']]
    ps = owv.Psi(nb, nx, ny, bases, int(g['nlevel']))
    nbasis = len(bases)
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, (nb, nx, ny), np.float64)
    data, w = g['data'], g['l1weight']

    def psiH(x):
        a = np.zeros((nb, nbasis, ps.Nymax, ps.Nxmax))
        ps.dot(x, a)
        return a

    def psi(a):
        x = np.zeros((nb, nx, ny))
        ps.hdot(a, x)
        return x
    for tag, pos, kw in (('pos1', 1, dict(tol=0.0, maxit=8, minit=2)), ('pos0', 0, dict(tol=0.0, maxit=5, minit=1)),
                         ('pos2', 2, dict(tol=0.0, maxit=6, minit=1)), ('tol', 1, dict(tol=5e-2, maxit=60, minit=3))):
        x, v = osv.primal_dual(np.zeros((nb, nx, ny)), np.zeros((nb, nbasis, ps.Nymax, ps.Nxmax)), float(g['lam']), psi,
                               psiH, float(g['hessnorm']), lambda u, s: opx.prox_21m(u, s, weight=w),
                               lambda u: ofc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, u) - data, nu=nbasis,
                               positivity=pos, verbosity=0, **kw)
        assert np.abs(x - g[f'{tag}_x']).max() <= 1e-13 * np.abs(g[f'{tag}_x']).max(), tag
        assert np.abs(v - g[f'{tag}_v']).max() <= 1e-13 * np.abs(g[f'{tag}_v']).max(), tag
