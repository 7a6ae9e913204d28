"""
GPU parity tests for the wavelet dictionary (psi / psi^H), the l21 prox / dual update and
the primal-dual backward step, against golden vectors from the reference source
(tests/golden/{psi,prox,pd}.npz) and the CPU oracle.  Mirrors the reference's own
tests/test_psi_operator.py and tests/test_wavelets.py.

Tolerances: fp64 1e-12 (psi coefficients vs reference: 1e-13 observed), fp32 1e-5.
"""
from functools import partial

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import wavelets as owv          # noqa: E402  (checker only)
from oracle import prox as opx              # noqa: E402
from oracle import solvers as osv           # noqa: E402
from oracle import fftconv as ofc           # noqa: E402

pmp = pytest.mark.parametrize


@pytest.fixture(scope='module')
def amd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.operators.psi import Psi
    from pfb_clean_amd.operators import psf
    from pfb_clean_amd.prox import prox_21m, prox_21
    from pfb_clean_amd.opt import primal_dual, power_method
    from pfb_clean_amd.utils import misc

    class NS:
        pass
    ns = NS()
    ns.Psi, ns.psf, ns.p21m, ns.p21, ns.pd, ns.pm, ns.misc = Psi, psf, prox_21m, prox_21, primal_dual, power_method, misc
    return ns


def maxerr(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max()


@pmp('rdt', [np.float64, np.float32])
def test_psi_matches_reference_golden(amd, golden, rdt):
    g = golden('psi')
    tol = 1e-12 if rdt == np.float64 else 2e-5
    for c in range(int(g['ncases'])):
        nband, nx, ny, nlevel, Nymax, Nxmax = (int(v) for v in g[f'p{c}_meta'])
        bases = [str(s) for s in g[f'p{c}_bases']]
        psi = amd.Psi(nband, nx, ny, bases, nlevel, 1)
        assert (psi.Nymax, psi.Nxmax, psi.nbasis) == (Nymax, Nxmax, len(bases))
        ref = g[f'p{c}_alpha']
        # sentinel-filled output: cells the reference never writes must stay untouched
        alpha = np.full(ref.shape, 7.25, dtype=rdt)
        psi.dot(g[f'p{c}_x'].astype(rdt), alpha)
        written = ~np.isnan(ref)
        assert np.all(alpha[~written] == 7.25), c
        scale = np.abs(ref[written]).max()
        assert maxerr(alpha[written], ref[written]) < tol * scale, c
        xrec = np.full((nband, nx, ny), -3.5, dtype=rdt)
        psi.hdot(g[f'p{c}_coef_in'].astype(rdt), xrec)
        rr = g[f'p{c}_xrec']
        assert maxerr(xrec, rr) < tol * np.abs(rr).max(), c


@pmp("nx", [128, 250])
@pmp("ny", [64, 78])
@pmp("nband", [1, 3, 6])
@pmp("nlevels", [1, 2])
def test_psi(amd, nx, ny, nband, nlevels):
    """reference tests/test_psi_operator.py:14-48: hdot(dot(x)) == nbasis * x to 1e-12 with
    randomly populated outputs."""
    np.random.seed(420)
    image = np.random.randn(nx, ny)
    nu = 1.0 + 0.1 * np.arange(nband)
    x = image[None, 0:nx, 0:ny] * nu[:, None, None] ** (-0.7)
    bases = ['self', 'db1', 'db2', 'db3', 'db4', 'db5']
    nbasis = len(bases)
    psi = amd.Psi(nband, nx, ny, bases, nlevels, 1)
    alpha = np.random.randn(nband, nbasis, psi.Nymax, psi.Nxmax)
    xrec = np.random.randn(nband, nx, ny)
    psi.dot(x, alpha)
    psi.hdot(alpha, xrec)
    np.testing.assert_array_almost_equal(nbasis * x, xrec, decimal=12)


@pmp("wavelet", ["db1", "db4", "db5", "db9"])
@pmp("shape", [(128, 256), (512, 128)])
@pmp("nlevel", [1, 2, 3])
def test_dwt_idwt_roundtrip_and_layout(amd, wavelet, shape, nlevel):
    """reference tests/test_wavelets.py:11-109: reconstruction + packed layout (the
    layout is checked against the oracle, itself pinned to the reference)."""
    rng = np.random.default_rng(5)
    nx, ny = shape
    data = rng.random((1, nx, ny))
    from pfb_clean_amd.wavelets import dwt_max_level
    if nlevel > dwt_max_level(min(nx, ny), wavelet):
        with pytest.raises(ValueError):
            amd.Psi(1, nx, ny, [wavelet], nlevel, 1)
        return
    psi = amd.Psi(1, nx, ny, [wavelet], nlevel, 1)
    ref = owv.Psi(1, nx, ny, [wavelet], nlevel, 1)
    assert (psi.Nymax, psi.Nxmax) == (ref.Nymax, ref.Nxmax)
    a = np.zeros((1, 1, psi.Nymax, psi.Nxmax))
    b = np.zeros_like(a)
    psi.dot(data, a)
    ref.dot(data, b)
    assert maxerr(a, b) < 1e-13
    rec = np.zeros_like(data)
    psi.hdot(a, rec)
    assert maxerr(rec, data) < 1e-12


@pmp("wavelet", ["db1", "db4", "db5"])
@pmp("shape", [(128, 256), (512, 128)])
@pmp("nlevel", [1, 2, 3])
def test_standalone_dwt2d_idwt2d_with_the_reference_argument_lists(amd, wavelet, shape, nlevel):
    """pfb.wavelets.dwt2d / idwt2d as the reference's own test calls them (tests/test_wavelets.py:11-109: same
    wavelets, shapes and level counts; wavelets.py:175-213, 261-315 for the argument lists): scratch buffers and
    bookkeeping handed in positionally, outputs written in place, reconstruction to 1e-12 and the packed layout equal
    to the oracle's (which is pinned to the reference's by tests/golden/psi.npz)."""
    from pfb_clean_amd.wavelets import dwt2d, idwt2d, level_sizes, filter_bank, coeff_size, signal_size
    rng = np.random.default_rng(12)
    nx, ny = shape
    dec_lo, dec_hi, rec_lo, rec_hi = filter_bank(wavelet)
    F = dec_lo.size
    sx, sy, spx, spy, ix, iy, ntx, nty = level_sizes(nx, ny, F, nlevel)
    bk = owv.Bookkeeping(nx, ny, F, nlevel)
    assert (ntx, nty) == (bk.Ntotx, bk.Ntoty) and list(sx) == bk.sx and list(spy) == bk.spy and ix == bk.ix and iy == bk.iy
    assert sx[0] == coeff_size(nx, F) and spx[0] == signal_size(sx[0], F)
    data = rng.random((nx, ny))
    alpha2 = np.full((nty, ntx), 3.5)                     # cells outside the level blocks must keep this value
    cbuff, cbuffT = np.zeros((ntx, nty)), np.zeros((nty, ntx))
    keep = data.copy()
    out = dwt2d(data, alpha2, cbuff, cbuffT, ix, iy, sx, sy, dec_lo, dec_hi, nlevel)
    assert out is alpha2 and np.array_equal(data, keep)
    want = np.full((nty, ntx), 3.5)
    owv.dwt2d(data, want, bk, dec_lo, dec_hi)
    assert maxerr(alpha2, want) < 1e-13
    xrec2 = np.full((nx, ny), -1.0)
    coeffs_scratch = np.zeros((nty, ntx))
    before = alpha2.copy()
    idwt2d(alpha2, xrec2, coeffs_scratch, cbuff, cbuffT, ix, iy, sx, sy, spx, spy, rec_lo, rec_hi, nlevel)
    np.testing.assert_array_almost_equal(data, xrec2, decimal=12)
    assert np.array_equal(alpha2, before)                 # the input coefficients are not modified
    # device tensors, fp32
    d32 = torch.from_numpy(data.astype(np.float32)).cuda()
    a32 = torch.zeros((nty, ntx), dtype=torch.float32, device='cuda')
    assert dwt2d(d32, a32, None, None, ix, iy, sx, sy, dec_lo, dec_hi, nlevel) is a32
    w0 = np.zeros((nty, ntx))
    owv.dwt2d(data, w0, bk, dec_lo, dec_hi)
    assert maxerr(a32.cpu().numpy(), w0) < 5e-6 * np.abs(w0).max()
    r32 = torch.empty_like(d32)
    idwt2d(a32, r32, None, None, None, ix, iy, sx, sy, spx, spy, rec_lo, rec_hi, nlevel)
    assert maxerr(r32.cpu().numpy(), data) < 2e-5
    # bookkeeping that does not belong to this transform is refused, not followed
    with pytest.raises(ValueError):
        dwt2d(data, np.zeros((nty + 1, ntx)), cbuff, cbuffT, ix, iy, sx, sy, dec_lo, dec_hi, nlevel)
    with pytest.raises(ValueError):
        dwt2d(data, alpha2, cbuff, cbuffT, ix, iy, tuple(v + 1 for v in sx), sy, dec_lo, dec_hi, nlevel)


def test_primal_dual_with_a_callers_own_affine_synthesis(amd):
    """`psiH` is an arbitrary callable in primal_dual_optimised's signature (primal_dual.py:91-101).  The linear-synthesis
    shortcut (2 psi^H(v) - psi^H(vp)) is only valid for a LINEAR synthesis that overwrites its output, so it is taken
    for this package's Psi.hdot alone; any other callable must see the reference's statement order vp = 2 v - vp;
    psiH(vp, xout) (primal_dual.py:137-138).  Here the synthesis is AFFINE (adds an offset, clips): the shortcut would
    give different iterates; the oracle runs the reference's statements with the same callable."""
    from pfb_clean_amd.opt.primal_dual import primal_dual_optimised
    rng = np.random.default_rng(21)
    nb, nx, ny = 2, 48, 40
    bases = ['self', 'db2']
    psi = amd.Psi(nb, nx, ny, bases, 2, 1)
    po = owv.Psi(nb, nx, ny, bases, 2, 1)
    nbasis = len(bases)
    data = rng.standard_normal((nb, nx, ny))
    x0 = 0.1 * rng.random((nb, nx, ny))
    a = np.zeros((nb, nbasis, po.Nymax, po.Nxmax))
    po.dot(x0, a)
    v0 = 0.05 * rng.standard_normal(a.shape) * (a != 0)
    w = np.ones((nbasis, po.Nymax, po.Nxmax))

    def syn_ref(vv, xo):
        po.hdot(vv, xo)
        xo += 0.01
        np.clip(xo, -0.8, 0.8, out=xo)

    def syn_gpu(vv, xo):
        psi.hdot(vv, xo)
        xo += 0.01
        xo.clamp_(-0.8, 0.8)
    xo_, vo_ = x0.copy(), v0.copy()
    osv.primal_dual_optimised(xo_, vo_, 0.02, syn_ref, po.dot, 1.0, None, w, None, lambda t: 0.3 * t - data,
                              nu=nbasis, tol=0.0, maxit=6, positivity=1, verbosity=0)
    dgpu = torch.from_numpy(data).cuda()
    xg, vg = torch.from_numpy(x0).cuda(), torch.from_numpy(v0).cuda()
    xg, vg = primal_dual_optimised(xg, vg, 0.02, syn_gpu, psi.dot, 1.0, None, torch.from_numpy(w).cuda(), None,
                                   lambda t: 0.3 * t - dgpu, nu=nbasis, tol=0.0, maxit=6, positivity=1, verbosity=0)
    assert maxerr(xg.cpu().numpy(), xo_) < 1e-12 * np.abs(xo_).max()
    assert maxerr(vg.cpu().numpy(), vo_) < 1e-12 * max(np.abs(vo_).max(), 1.0)
    # and the shortcut is really not equivalent here: with Psi.hdot handed in directly the iterates differ
    xl, vl = primal_dual_optimised(torch.from_numpy(x0).cuda(), torch.from_numpy(v0).cuda(), 0.02, psi.hdot, psi.dot, 1.0,
                                   None, torch.from_numpy(w).cuda(), None, lambda t: 0.3 * t - dgpu, nu=nbasis, tol=0.0,
                                   maxit=6, positivity=1, verbosity=0)
    assert maxerr(xl.cpu().numpy(), xo_) > 1e-6


def test_psi_band_maker_single_band_object(amd):
    """psi_band_maker (psi.py:17-123) / psi_band.dot / hdot (psi.py:187-256): the per-band object of the reference, 2-D
    image and (nbasis, Nymax, Nxmax) coefficients, in place, numpy and device tensors; equals one band of Psi."""
    from pfb_clean_amd.operators.psi import psi_band_maker
    rng = np.random.default_rng(31)
    nx, ny, bases, nlevel = 128, 64, ['self', 'db1', 'db3'], 2
    pb = psi_band_maker(nx, ny, bases, nlevel)
    ref = owv.Psi(1, nx, ny, bases, nlevel, 1)
    assert (pb.Nxmax, pb.Nymax, pb.nbasis, pb.Nx, pb.Ny) == (ref.Nxmax, ref.Nymax, 3, nx, ny)
    x = rng.standard_normal((nx, ny))
    a = np.full((3, pb.Nymax, pb.Nxmax), 2.5)
    want = np.full((1, 3, pb.Nymax, pb.Nxmax), 2.5)
    assert pb.dot(x, a) is a
    ref.dot(x[None], want)
    assert maxerr(a, want[0]) < 1e-13
    xo = np.empty((nx, ny))
    pb.hdot(a * (want[0] != 2.5), xo)
    back = np.zeros((1, nx, ny))
    ref.hdot(want * (want != 2.5), back)
    assert maxerr(xo, back[0]) < 1e-12
    xt = torch.from_numpy(x).cuda()
    at = torch.zeros((3, pb.Nymax, pb.Nxmax), dtype=torch.float64, device='cuda')
    pb.dot(xt, at)
    assert maxerr(at.cpu().numpy() * (want[0] != 2.5), want[0] * (want[0] != 2.5)) < 1e-13
    with pytest.raises(ValueError):
        psi_band_maker(16, 16, ['db5'], 3)
    with pytest.raises(ValueError):
        pb.dot(x[None], a)


def test_psi_device_tensors_adjoint_and_errors(amd):
    rng = np.random.default_rng(9)
    nband, nx, ny = 2, 96, 80
    bases = ['self', 'db2', 'db4']
    psi = amd.Psi(nband, nx, ny, bases, 2, 1)
    x = torch.from_numpy(rng.standard_normal((nband, nx, ny))).cuda()
    a = torch.zeros((nband, 3, psi.Nymax, psi.Nxmax), dtype=torch.float64, device='cuda')
    out = psi.dot(x, a)
    assert out is a and a.is_cuda
    mark = torch.full_like(a, float('nan'))
    psi.dot(x, mark)
    c = torch.from_numpy(rng.standard_normal(tuple(a.shape))).cuda()
    c[torch.isnan(mark)] = 0.0
    y = torch.empty_like(x)
    psi.hdot(c, y)
    lhs = torch.sum(a * c).item()
    rhs = torch.sum(x * y).item()
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)
    with pytest.raises(ValueError):
        amd.Psi(1, 16, 16, ['db5'], 3, 1)
    with pytest.raises(ValueError):
        psi.dot(x[:1], a)


@pmp('rdt', [np.float64, np.float32])
def test_prox_golden(amd, golden, rdt):
    g = golden('prox')
    v, vp, w = g['v'].astype(rdt), g['vp'].astype(rdt), g['w'].astype(rdt)
    rtol = 1e-12 if rdt == np.float64 else 2e-5
    for i, (lam, sigma) in enumerate(g['grid']):
        res = np.full(v.shape, np.nan, dtype=rdt)
        amd.p21m.prox_21m_numba(v, res, lam, sigma=sigma, weight=w)
        ref = g[f'g{i}_prox21m_numba']
        assert maxerr(res, ref) <= rtol * max(np.abs(ref).max(), 1e-30), (i, 'prox')
        vv = v.copy()
        amd.p21m.dual_update_numba(vp, vv, lam, sigma=sigma, weight=w)
        ref = g[f'g{i}_dual_update_numba']
        assert maxerr(vv, ref) <= rtol * np.abs(ref).max(), (i, 'dual')
        assert maxerr(amd.p21m.prox_21m(v, lam, weight=w), g[f'g{i}_prox21m']) <= rtol * 10
        assert maxerr(amd.p21.prox_21(v, lam, weight=w), g[f'g{i}_prox21']) <= rtol * 10
        # band-l2-norm variants (prox/prox_21.py:23-48, 62-88), the reference's flattened (nband, nbasis, ntot) shapes
        nb_, nbas = v.shape[:2]
        v3, vp3, w3 = v.reshape(nb_, nbas, -1), vp.reshape(nb_, nbas, -1), w.reshape(nbas, -1)
        r3 = np.full(v3.shape, np.nan, dtype=rdt)
        assert amd.p21.prox_21_numba(v3, r3, lam, sigma=sigma, weight=w3) is r3
        ref = g[f'g{i}_prox21_numba']
        assert maxerr(r3, ref) <= rtol * max(np.abs(ref).max(), 1e-30), (i, 'prox21_numba')
        vv3 = v3.copy()
        amd.p21.dual_update_numba(vp3, vv3, lam, sigma=sigma, weight=w3)
        ref = g[f'g{i}_dual_update_numba_l2']
        assert maxerr(vv3, ref) <= rtol * np.abs(ref).max(), (i, 'dual l2')


@pmp("nband", [1, 3, 6])
@pmp("lam,sigma", [(1.0, 75.0), (1e-1, 1.0), (1e-3, 1e-3)])
def test_dual_update_identity(amd, nband, lam, sigma):
    """reference tests/test_psi_operator.py:150-193."""
    rng = np.random.default_rng(11)
    nx, ny = 120, 150
    bases = ['self', 'db1', 'db2', 'db3', 'db4', 'db5']
    psi = amd.Psi(nband, nx, ny, bases, 2, 1)
    w = rng.random((len(bases), psi.Nymax, psi.Nxmax))
    x = rng.standard_normal((nband, nx, ny))
    v = np.zeros((nband, len(bases), psi.Nymax, psi.Nxmax))
    psi.dot(rng.standard_normal(x.shape), v)
    vp = v.copy()
    res1 = amd.p21m.dual_update(v, x, psi.dot, lam, sigma=sigma, weight=w)
    res_ref = opx.dual_update(vp.copy(), x, owv.Psi(nband, nx, ny, bases, 2, 1).dot, lam, sigma=sigma, weight=w)
    psi.dot(x, v)
    amd.p21m.dual_update_numba(vp, v, lam, sigma=sigma, weight=w)
    np.testing.assert_array_almost_equal(1 + res_ref, 1 + v, decimal=9)
    np.testing.assert_array_almost_equal(1 + res1, 1 + v, decimal=9)


def test_primal_dual_trajectory_golden(amd, golden):
    g = golden('pd')
    psfhat, Q = g['psfhat'], int(g['Q'])
    nb, P, _ = psfhat.shape
    nx = ny = P // 2
    bases = [str(s) for s in g['bases']]
    psi = amd.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    data = g['data']
    conv = partial(amd.psf.psf_convolve_cube, None, None, None, psfhat, Q)

    def grad21(x):
        return conv(x) - data
    l1w = np.ones((len(bases), psi.Nymax, psi.Nxmax))
    for tag, pos, maxit in (('pos1_it10', 1, 10), ('pos0_it4', 0, 4), ('pos2_it6', 2, 6)):
        x0 = np.zeros((nb, nx, ny))
        v0 = np.zeros((nb, len(bases), psi.Nymax, psi.Nxmax))
        x, v = amd.pd.primal_dual_optimised(x0, v0, float(g['lam']), psi.hdot, psi.dot,
                                            float(g['hessnorm']), None, l1w, None, grad21,
                                            nu=len(bases), tol=0.0, maxit=maxit, positivity=pos,
                                            verbosity=0)
        assert x is x0 and v is v0                       # in place, like the reference
        assert maxerr(x, g[f'{tag}_x']) < 1e-9 * max(np.abs(g[f'{tag}_x']).max(), 1e-30), tag
        assert maxerr(v, g[f'{tag}_v']) < 1e-9 * np.abs(g[f'{tag}_v']).max(), tag


def test_primal_dual_device_resident_with_reweighting(amd, golden):
    """Tensor-in/tensor-out run with an l1 reweighter (spotless.py:243-285 wiring) against
    the oracle's primal_dual_optimised on the same inputs."""
    g = golden('pd')
    psfhat, Q = g['psfhat'], int(g['Q'])
    nb, P, _ = psfhat.shape
    nx = ny = P // 2
    bases = [str(s) for s in g['bases']]
    nbasis = len(bases)
    data = g['data']
    lam, L = float(g['lam']), float(g['hessnorm'])
    # oracle
    opsi = owv.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, (nb, nx, ny), np.float64)

    def ograd(x):
        return ofc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, x) - data
    rms = np.full((nbasis, opsi.Nymax, opsi.Nxmax), 1e-3)
    ooutvar = np.zeros((nb, nbasis, opsi.Nymax, opsi.Nxmax))
    orew = partial(osv.l1reweight_func, opsi.dot, ooutvar, 1.0, rms, alpha=2)
    xo, vo = osv.primal_dual_optimised(np.zeros((nb, nx, ny)), np.zeros_like(ooutvar), lam, opsi.hdot,
                                       opsi.dot, L, None, np.ones_like(rms), orew, ograd, nu=nbasis,
                                       tol=5e-2, maxit=40, positivity=1, maxreweight=3)
    # device
    dev = torch.device('cuda')
    psi = amd.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    conv = partial(amd.psf.psf_convolve_cube, None, None, None, torch.from_numpy(psfhat).to(dev), Q)
    datad = torch.from_numpy(data).to(dev)

    def grad(x):
        return conv(x) - datad
    outvar = torch.zeros((nb, nbasis, psi.Nymax, psi.Nxmax), dtype=torch.float64, device=dev)
    rew = partial(amd.misc.l1reweight_func, psi.dot, outvar, 1.0, torch.from_numpy(rms).to(dev), alpha=2)
    x = torch.zeros((nb, nx, ny), dtype=torch.float64, device=dev)
    v = torch.zeros_like(outvar)
    x2, v2 = amd.pd.primal_dual_optimised(x, v, lam, psi.hdot, psi.dot, L, None,
                                          torch.ones_like(outvar[0]), rew, grad, nu=nbasis, tol=5e-2,
                                          maxit=40, positivity=1, maxreweight=3, verbosity=0)
    assert x2 is x and x2.is_cuda
    assert maxerr(x2.cpu().numpy(), xo) < 1e-9 * np.abs(xo).max()
    assert maxerr(v2.cpu().numpy(), vo) < 1e-9 * np.abs(vo).max()


def test_primal_dual_nonzero_margins_keep_reference_semantics(amd, golden):
    """The packed coefficient layout has margins psi never writes.  The reference drags whatever the
    caller put there through `vtilde = vp + sigma v` and `vp = v.copy()`; the buffer rotation of
    opt/primal_dual.py is only used when they are zero -- with garbage in the margins the result
    (margins included) must still equal the oracle's."""
    g = golden('pd')
    psfhat, Q = g['psfhat'], int(g['Q'])
    nb, P, _ = psfhat.shape
    nx = ny = P // 2
    bases = [str(s) for s in g['bases']]
    nbasis = len(bases)
    data = g['data']
    rng = np.random.default_rng(77)
    opsi = owv.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    v0 = rng.standard_normal((nb, nbasis, opsi.Nymax, opsi.Nxmax))
    x0 = rng.standard_normal((nb, nx, ny))
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, (nb, nx, ny), np.float64)
    l1w = 0.5 + rng.random((nbasis, opsi.Nymax, opsi.Nxmax))
    xo, vo = osv.primal_dual_optimised(x0.copy(), v0.copy(), float(g['lam']), opsi.hdot, opsi.dot,
                                       float(g['hessnorm']), None, l1w,
                                       None, lambda x: ofc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, x) - data,
                                       nu=nbasis, tol=0.0, maxit=7, positivity=1)
    psi = amd.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    conv = partial(amd.psf.psf_convolve_cube, None, None, None, psfhat, Q)
    x, v = amd.pd.primal_dual_optimised(x0.copy(), v0.copy(), float(g['lam']), psi.hdot, psi.dot,
                                        float(g['hessnorm']), None, l1w, None,
                                        lambda t: conv(t) - data, nu=nbasis, tol=0.0, maxit=7,
                                        positivity=1, verbosity=0)
    assert maxerr(x, xo) < 1e-9 * np.abs(xo).max()
    assert maxerr(v, vo) < 1e-9 * np.abs(vo).max()


@pytest.mark.parametrize('rdt', [np.float64, np.float32])
def test_psi_batched_and_fused_kernels_match_per_basis_kernels(amd, rdt, monkeypatch):
    """Default: one launch per level for all wavelet bases (k_dwt_batched / k_idwt_batched) and one
    fused finest synthesis level (k_idwt_finest_fused).  PFB_PSI_FUSED=0 runs the per-basis kernels
    instead; both must give the same coefficients / image (incl. db5+, the FMAX = 18 variants)."""
    rng = np.random.default_rng(17)
    nband, nx, ny = 3, 200, 168
    for bases, nlevel in ((['self', 'db1', 'db2', 'db3', 'db4'], 3), (['db6', 'self', 'db2'], 2), (['db3'], 1)):
        psi = amd.Psi(nband, nx, ny, bases, nlevel, 1, dtype=torch.float64 if rdt == np.float64 else torch.float32)
        x = torch.from_numpy(rng.standard_normal((nband, nx, ny)).astype(rdt)).cuda()
        shape = (nband, len(bases), psi.Nymax, psi.Nxmax)
        a1 = torch.zeros(shape, dtype=x.dtype, device='cuda')
        a0 = torch.zeros_like(a1)
        psi.dot(x, a1)
        c = torch.from_numpy(rng.standard_normal(shape).astype(rdt)).cuda()
        y1 = torch.empty_like(x)
        psi.hdot(c, y1)
        monkeypatch.setenv('PFB_PSI_FUSED', '0')
        psi.dot(x, a0)
        y0 = torch.empty_like(x)
        psi.hdot(c, y0)
        monkeypatch.delenv('PFB_PSI_FUSED')
        assert torch.equal(a0, a1)
        tol = 1e-14 if rdt == np.float64 else 1e-6
        assert (y0 - y1).abs().max().item() <= tol * y0.abs().max().item()


# ---- the reference's own prox tests (tests/test_psi_operator.py:50-148), same statements on the GPU
@pmp("prox", ['prox_21', 'prox_21m'])
@pmp("nx,ny", [(120, 64), (240, 150)])
@pmp("nband", [1, 3])
@pmp("nlevels", [1, 2])
def test_prox_with_zero_step_is_the_identity(amd, prox, nx, ny, nband, nlevels):
    """test_prox21 / test_prox21m: dot -> prox(., 0, w) -> hdot gives nbasis * x (decimal = 12)."""
    rng = np.random.default_rng(420)
    image = rng.standard_normal((nx, ny))
    nu = 1.0 + 0.1 * np.arange(nband)
    x = image[None] * nu[:, None, None] ** (-0.7)
    bases = ['self', 'db1', 'db2', 'db3', 'db4', 'db5']
    nbasis = len(bases)
    psi = amd.Psi(nband, nx, ny, bases, nlevels, 1)
    w21 = rng.random((nbasis, psi.Nymax, psi.Nxmax))
    alpha = np.zeros((nband, nbasis, psi.Nymax, psi.Nxmax))
    xrec = np.zeros((nband, nx, ny))
    psi.dot(x, alpha)
    y = (amd.p21.prox_21 if prox == 'prox_21' else amd.p21m.prox_21m)(alpha, 0.0, w21)
    psi.hdot(y, xrec)
    np.testing.assert_array_almost_equal(nbasis * x, xrec, decimal=12)


@pmp("nymax,nxmax", [(1234, 134), (240, 896)])
@pmp("nbasis", [1, 5])
@pmp("nband", [1, 6])
@pmp("lam,sigma", [(1.0, 75.0), (1e-1, 1.0), (1e-3, 1e-3)])
def test_prox21m_numba_matches_prox21m(amd, nband, nbasis, nymax, nxmax, lam, sigma):
    """test_prox21m_numba: the in-place kernel equals prox_21m even when the output holds random
    numbers initially, also with the sigma scaling."""
    rng = np.random.default_rng(420)
    v = rng.standard_normal((nband, nbasis, nymax, nxmax))
    vout = rng.standard_normal((nband, nbasis, nymax, nxmax))
    w = rng.random((nbasis, nymax, nxmax))
    res = amd.p21m.prox_21m(v, lam, weight=w)
    amd.p21m.prox_21m_numba(v, vout, lam, weight=w)
    np.testing.assert_array_almost_equal(res, vout, decimal=12)
    res = amd.p21m.prox_21m(v / sigma, lam / sigma, weight=w)
    amd.p21m.prox_21m_numba(v, vout, lam, sigma=sigma, weight=w)
    np.testing.assert_array_almost_equal(res, vout, decimal=8)


@pytest.mark.parametrize('case', [(2, 256, 192, ['self', 'db6', 'db7', 'db8', 'db9'], 2),
                                  (1, 300, 260, ['db9', 'db2', 'db7'], 3),
                                  (1, 514, 390, ['self', 'db1', 'db6', 'db9'], 2)],
                         ids=['db6-9', 'db9-mixed-3lev', 'wide-mixed'])
@pytest.mark.parametrize('rdt', [np.float64, np.float32])
def test_long_filters_against_oracle(amd, case, rdt):
    """Filters beyond the reference tests' db5 (F = 12 .. 18): the FMAX = 18 instantiations of the batched
    level kernels and the wide-tile staging paths (tile rows wider than a wavefront by up to 16 columns)."""
    nb, nx, ny, bases, nl = case
    rng = np.random.default_rng(3)
    x = rng.standard_normal((nb, nx, ny)).astype(rdt)
    po = owv.Psi(nb, nx, ny, bases, nl)
    a_ref = np.zeros((nb, po.nbasis, po.Nymax, po.Nxmax))
    po.dot(x.astype(np.float64), a_ref)
    psi = amd.Psi(nb, nx, ny, bases, nl, 1)
    a = np.zeros(a_ref.shape, dtype=rdt)
    psi.dot(x, a)
    c = rng.standard_normal(a_ref.shape).astype(rdt)
    xo_ref = np.zeros((nb, nx, ny))
    po.hdot(c.astype(np.float64), xo_ref)
    xo = np.full((nb, nx, ny), np.nan, dtype=rdt)
    psi.hdot(c, xo)
    tol = 1e-12 if rdt == np.float64 else 2e-5
    assert np.abs(a - a_ref).max() < tol * np.abs(a_ref).max()
    assert np.abs(xo - xo_ref).max() < tol * np.abs(xo_ref).max()


@pytest.mark.parametrize('kind', ['numpy', 'tensor'])
def test_primal_dual_unoptimised_golden(amd, golden, kind):
    """primal_dual (primal_dual.py:12-87, the functional form workers/fwdbwd.py:367 names) against the reference's own
    trajectories (tests/golden/pdplain.npz): operators built from this package's Psi / prox_21m / psf_convolve_cube,
    numpy in -> numpy out and tensors staying on the device."""
    from pfb_clean_amd.opt.primal_dual import primal_dual
    from pfb_clean_amd.prox.prox_21m import prox_21m
    g = golden('pdplain')
    psfhat, Q = g['psfhat'], int(g['Q'])
    nb, P, _ = psfhat.shape
    nx, ny = P // 2, Q // 2
    bases = [str(b) for b in g['bases']]
    ps = amd.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
    nbasis = len(bases)
    dev = torch.device('cuda')
    t = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)) if kind == 'tensor' else (lambda a: a)
    data, w, ph = t(g['data']), t(g['l1weight']), t(psfhat)
    zeros = (lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)) if kind == 'tensor' else (lambda *s: np.zeros(s))

    def psiH(x):
        a = zeros(nb, nbasis, ps.Nymax, ps.Nxmax)
        ps.dot(x, a)
        return a

    def psi(a):
        x = zeros(nb, nx, ny)
        ps.hdot(a, x)
        return x

    def grad(x):
        return amd.psf.psf_convolve_cube(None, None, None, ph, Q, x) - data
    for tag, pos, kw in (('pos1', 1, dict(tol=0.0, maxit=8, minit=2)), ('pos0', 0, dict(tol=0.0, maxit=5, minit=1)),
                         ('pos2', 2, dict(tol=0.0, maxit=6, minit=1)), ('tol', 1, dict(tol=5e-2, maxit=60, minit=3))):
        x0, v0 = zeros(nb, nx, ny), zeros(nb, nbasis, ps.Nymax, ps.Nxmax)
        x, v = primal_dual(x0, v0, float(g['lam']), psi, psiH, float(g['hessnorm']),
                           lambda u, s: prox_21m(u, s, weight=w), grad, nu=nbasis, positivity=pos, verbosity=0, **kw)
        xh = x.cpu().numpy() if kind == 'tensor' else x
        vh = v.cpu().numpy() if kind == 'tensor' else v
        assert (kind == 'tensor') == isinstance(x, torch.Tensor)
        assert maxerr(xh, g[f'{tag}_x']) < 1e-9 * np.abs(g[f'{tag}_x']).max(), tag
        assert maxerr(vh, g[f'{tag}_v']) < 1e-9 * np.abs(g[f'{tag}_v']).max(), tag
