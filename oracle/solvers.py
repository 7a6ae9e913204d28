"""
CPU restatement of the iterative solvers on the hot path.

Test infrastructure (see oracle/__init__.py).  Follows
  pfb/opt/pcg.py:53-136            pcg
  pfb/opt/pcg.py:243-291           _pcg_psf_impl (per-band PSF PCG)
  pfb/utils/misc.py:1316-1351      norm_diff
  pfb/opt/power_method.py:11-49    power_method
  pfb/opt/primal_dual.py:91-180    primal_dual_optimised
  pfb/utils/misc.py:1070-1080      l1reweight_func
  pfb/opt/pcg.py:363-420           pcg_dist (per-band variant, disabled in the live workers)
  pfb/opt/pcg.py:139-239           cg_dct (CG over a nested dict of images, unused by the live workers)
  pfb/utils/misc.py:664-739        dds2cubes (cube assembly)
  pfb/utils/misc.py:1366-1423      freqmul, setup_parametrisation
All of misc.py's functions here are pinned by tests/golden/misc.npz, produced by the reference's real
module (tests/golden/_refstubs.py:load_misc).

The recurrences are written out in full so the
semantics that matter for parity are visible: residual sign r = A x - b, the
stopping rule, the backtracking loop that costs no extra matvec, the break on an
all-zero search direction that does NOT increment k.
"""
import numpy as np
from . import fftconv as _fc
from . import prox as _prox


# --------------------------------------------------------------------- misc
def dds2cubes(dds, nband, apparent=False, dual=True, modelname='MODEL'):
    """misc.py:664-739 on in-memory datasets (each `ds` exposes DIRTY, BEAM, WSUM and optionally
    RESIDUAL, PSF, PSFHAT, DUAL, <modelname> with `.values`, plus `bandid`; `name in ds` works).
    Several datasets may share a band (they are SUMMED: dirty/residual weighted by the beam unless
    `apparent`, psf/psfhat plain); model and dual are taken from the LAST dataset of a band;
    mean_beam = sum(beam*wsum)/wsums[b]; dirty, residual, psf, psfhat are divided by the TOTAL wsum."""
    d0 = dds[0]
    rt = d0.DIRTY.values.dtype
    ct = np.result_type(rt, np.complex64)
    nx, ny = d0.DIRTY.values.shape
    dirty = np.zeros((nband, nx, ny), rt)
    model = np.zeros((nband, nx, ny), rt)
    residual = np.zeros((nband, nx, ny), rt) if 'RESIDUAL' in d0 else None
    wsums = np.zeros(nband, rt)
    psf = psfhat = None
    if 'PSF' in d0:
        psf = np.zeros((nband,) + d0.PSF.values.shape, rt)
        psfhat = np.zeros((nband,) + d0.PSFHAT.values.shape, ct)
    mean_beam = np.zeros((nband, nx, ny), rt)
    dualc = np.zeros((nband,) + d0.DUAL.values.shape, rt) if (dual and 'DUAL' in d0) else None
    for ds in dds:
        b = ds.bandid
        beam = ds.BEAM.values
        dirty[b] += ds.DIRTY.values if apparent else ds.DIRTY.values * beam
        if 'RESIDUAL' in ds:
            residual[b] += ds.RESIDUAL.values if apparent else ds.RESIDUAL.values * beam
        if 'PSF' in ds:
            psf[b] += ds.PSF.values
            psfhat[b] += ds.PSFHAT.values
        if modelname in ds:
            model[b] = ds[modelname].values
        if dual and 'DUAL' in ds:
            dualc[b] = ds.DUAL.values
        mean_beam[b] += beam * ds.WSUM.values[0]
        wsums[b] += ds.WSUM.values[0]
    wsum = wsums.sum()
    dirty /= wsum
    if residual is not None:
        residual /= wsum
    if psf is not None:
        psf /= wsum
        psfhat /= wsum
    for b in range(nband):
        if wsums[b]:
            mean_beam[b] /= wsums[b]
    return dirty, model, residual, psf, psfhat, mean_beam, wsums, dualc


def norm_diff(x, xp):
    """misc.py:1326-1351: sqrt(sum((x-xp)^2) / (1e-12 + sum(x^2))), float64
    accumulators, 2-D or 3-D input only."""
    if x.ndim not in (2, 3):
        raise ValueError("norm_diff is only implemented for 2D or 3D arrays")
    d = x.astype(np.float64) - xp.astype(np.float64)
    num = float(np.sum(d * d))
    den = 1e-12 + float(np.sum(x.astype(np.float64) ** 2))
    return np.sqrt(num / den)


def l1reweight_func(psiH, outvar, rmsfactor, rms_comps, model, alpha=4):
    """misc.py:1070-1080.  `psiH` is the analysis operator Psi.dot here."""
    psiH(model, outvar)
    mcomps = np.abs(np.sum(outvar, axis=0))
    return (1 + rmsfactor) / (1 + mcomps ** alpha / rms_comps ** alpha)


def freqmul(A, x):
    """misc.py:1366-1375: out[k] = sum_l A[k, l] x[l] over an (nband, nx, ny) cube."""
    return np.einsum('kl,lij->kij', A, x)


def setup_parametrisation(mode='id', minval=1e-5, sigma=1.0, freq=None, lscale=1.0):
    """misc.py:1378-1423: x = f(s), K = sigma^2 exp(-(nu_i - nu_j)^2 / (2 lscale^2)) = L L^T over the
    normalised band frequencies; returns (func, finv, dfunc, dhfunc)."""
    from scipy.linalg import solve_triangular
    nu = freq / np.mean(freq)
    nband = nu.size
    K = sigma ** 2 * np.exp(-(nu[:, None] - nu[None, :]) ** 2 / (2 * lscale ** 2))
    L = np.linalg.cholesky(K + 1e-10 * np.eye(nband))
    LH = L.T

    def solve(x):
        return solve_triangular(L, x.reshape(nband, -1), lower=True).reshape(x.shape)
    if mode == 'id':
        return ((lambda x: freqmul(L, x)), solve,
                (lambda x0, v: freqmul(L, v)), (lambda x0, v: freqmul(LH, v)))
    if mode == 'exp':
        return ((lambda x: np.exp(freqmul(L, x))),
                (lambda x: np.log(np.maximum(np.abs(solve(x)), minval))),
                (lambda x0, v: np.exp(freqmul(L, x0)) * freqmul(L, v)),
                (lambda x0, v: freqmul(LH, v * np.exp(freqmul(L, x0)))))
    raise ValueError(f"Unknown mode - {mode}")


# ---------------------------------------------------------------------- pcg
class PCGTrace:
    """Optional recorder of the iterate history (used for golden vectors)."""

    def __init__(self, keep=()):
        self.keep = set(keep)
        self.x_at = {}
        self.eps = []
        self.alpha = []
        self.nbacktrack = []
        self.k_exit = None
        self.status = None


def pcg(A, b, x0=None, M=None, tol=1e-5, maxit=500, minit=100, verbosity=1,
        report_freq=10, backtrack=True, return_resid=False, trace=None):
    """pcg.py:53-136."""
    if x0 is None:
        x0 = np.zeros(b.shape, dtype=b.dtype)
    if M is None:
        def M(v):
            return v

    r = A(x0) - b                       # pcg.py:71   residual convention A x - b
    y = M(r)
    if not np.any(y):                   # pcg.py:73-75  early exit returns x0 ONLY
        if trace is not None:
            trace.status = 'zero-residual'
            trace.k_exit = 0
        return x0
    p = -y
    rnorm = np.vdot(r, y)
    k = 0
    x = x0
    eps = 1.0
    status = None
    while (eps > tol or k < minit) and k < maxit:   # stall_count is inert
        xp = x.copy()
        rp = r.copy()
        Ap = A(p)                       # the one matvec per iteration
        rnorm = np.vdot(r, y)
        alpha = rnorm / np.vdot(p, Ap)
        x = xp + alpha * p
        r = rp + alpha * Ap
        y = M(r)
        rnorm_next = np.vdot(r, y)
        nbt = 0
        while rnorm_next > rnorm and backtrack:     # pcg.py:96-101
            alpha *= 0.75
            x = xp + alpha * p
            r = rp + alpha * Ap
            y = M(r)
            rnorm_next = np.vdot(r, y)
            nbt += 1
        beta = rnorm_next / rnorm
        p = beta * p - y
        if not np.any(p):               # pcg.py:106-107: break BEFORE k += 1
            status = 'breakdown'
            break
        rnorm = rnorm_next
        k += 1
        eps = norm_diff(x, xp)
        if trace is not None:
            trace.eps.append(float(eps))
            trace.alpha.append(float(alpha))
            trace.nbacktrack.append(nbt)
            if k in trace.keep:
                trace.x_at[k] = x.copy()

    if trace is not None:
        trace.k_exit = k
        trace.status = status or ('maxit' if k >= maxit else 'converged')
    if not return_resid:
        return x
    return x, r


def cg(A, b, x0=None, tol=1e-5, maxit=500, verbosity=1, report_freq=10):
    """pcg.py:12-50: plain CG, eps = <r,r> (unnormalised), in-place x/r updates."""
    x = np.zeros(b.shape, dtype=b.dtype) if x0 is None else x0.copy()
    r = A(x) - b
    p = -r
    rnorm = np.vdot(r, r)
    eps, k = rnorm, 0
    while eps > tol and k < maxit:
        Ap = A(p)
        alpha = rnorm / np.vdot(p, Ap)
        x += alpha * p
        r += alpha * Ap
        rnorm_next = np.vdot(r, r)
        beta = rnorm_next / rnorm
        p = beta * p - r
        rnorm = rnorm_next
        eps = rnorm
        k += 1
    return x


def cg_dct(A, b, x, tol=1e-5, maxit=500, verbosity=1, report_freq=10):
    """pcg.py:139-239: plain CG (no preconditioner, no backtracking) whose unknown is a nested
    dict {field: {'t..b..': image}}; A maps such a dict to one of the same structure.
    r = A x - b, eps = <r, r> summed over all leaves, rule `eps > tol and k < maxit`; x's leaves
    are updated IN PLACE (`a[field][i] += alpha*b[field][i]`).  Returns (x, r)."""
    def vdot(u, v):
        return sum(np.vdot(u[f][i], v[f][i]) for f in u for i in u[f])
    Ax = A(x)
    r = {f: {i: Ax[f][i] - b[f][i] for i in Ax[f]} for f in Ax}
    p = {f: {i: -r[f][i] for i in r[f]} for f in r}
    rnorm = vdot(r, r)
    eps, k = rnorm, 0
    while eps > tol and k < maxit:
        Ap = A(p)
        alpha = rnorm / vdot(p, Ap)
        for f in x:
            for i in x[f]:
                x[f][i] += alpha * p[f][i]
                r[f][i] += alpha * Ap[f][i]
        rnorm_next = vdot(r, r)
        beta = rnorm_next / rnorm
        for f in p:
            for i in p[f]:
                p[f][i] = beta * p[f][i] - r[f][i]
        rnorm = rnorm_next
        eps = rnorm
        k += 1
    return x, r


def pcg_dist(A, maxit, minit, tol, sigmainv):
    """pcg.py:363-420.  Differs from pcg(): b = A.residual/A.wsum (A.dirty if there is no
    residual), x0 = 0, M = x/sigmainv always, eps = rnorm/eps0 with eps0 the INITIAL <r,y>
    (1.0 if that is NaN or 0), a stall counter (|eps - epsp| < 1e-3 tol, five times) joins the
    stopping rule, rnorm is re-evaluated from the current r, y at the top of each iteration."""
    b = (A.residual if hasattr(A, 'residual') else A.dirty) / A.wsum
    x = np.zeros_like(b)
    r = A(x) - b
    y = r / sigmainv
    p = -y
    rnorm = np.vdot(r, y)
    eps0 = 1.0 if (np.isnan(rnorm) or rnorm == 0.0) else rnorm
    k, eps, stall = 0, 1.0, 0
    while (eps > tol or k < minit) and k < maxit and stall < 5:
        xp, rp, epsp = x.copy(), r.copy(), eps
        Ap = A(p)
        rnorm = np.vdot(r, y)
        alpha = rnorm / np.vdot(p, Ap)
        x = xp + alpha * p
        r = rp + alpha * Ap
        y = r / sigmainv
        rnorm_next = np.vdot(r, y)
        while rnorm_next > rnorm:
            alpha *= 0.75
            x = xp + alpha * p
            r = rp + alpha * Ap
            y = r / sigmainv
            rnorm_next = np.vdot(r, y)
        beta = rnorm_next / rnorm
        p = beta * p - y
        if not np.any(p):
            break
        rnorm = rnorm_next
        k += 1
        eps = rnorm / eps0
        if np.abs(eps - epsp) < 1e-3 * tol:
            stall += 1
    return x


def pcg_psf(psfhat, b, x0, beam, lastsize, nthreads, sigmainv, cgopts):
    """pcg.py:243-360 without the dask wrapping: independent PCG per band with
    A = _hessian_psf_slice(psfhat[k], beam[k], sigmainv), M = x/sigmainv."""
    nband, nx, ny = b.shape
    if beam is not None:
        if beam.ndim == 2:
            beam = beam[None]
        if beam.shape[0] == 1:
            beam = np.tile(beam, (nband, 1, 1))
        elif beam.shape[0] != nband:
            raise ValueError('Beam has incorrect shape')
    model = np.zeros((nband, nx, ny), dtype=b.dtype)
    if sigmainv > 0:
        def M(v):
            return v / sigmainv
    else:
        M = None
    for k in range(nband):
        xpad, xhat, xout = _fc.make_scratch(psfhat[k], lastsize, (nx, ny), b.dtype)
        bk = None if beam is None else beam[k]

        def A(v, xpad=xpad, xhat=xhat, xout=xout, k=k, bk=bk):
            return _fc._hessian_psf_slice(xpad, xhat, xout, psfhat[k], bk, lastsize,
                                          v, nthreads=nthreads, sigmainv=sigmainv)
        model[k] = pcg(A, b[k], x0[k], M=M, **cgopts)
    return model


# ------------------------------------------------------------- power method
def power_method(A, imsize, b0=None, tol=1e-5, maxit=250, verbosity=1,
                 report_freq=25):
    """power_method.py:11-49.  A may return an aliased buffer that is normalised
    in place (power_method.py:29-36); b0=None draws an unseeded randn start."""
    if b0 is None:
        b = np.random.randn(*imsize)
        b /= np.linalg.norm(b)
    else:
        b = b0 / np.linalg.norm(b0)
    beta = 1.0
    eps = 1.0
    k = 0
    bp = b.copy()
    while eps > tol and k < maxit:
        b = A(bp)
        bnorm = np.linalg.norm(b)
        betap = beta
        beta = np.vdot(bp, b) / np.vdot(bp, bp)
        b /= bnorm
        eps = np.linalg.norm(beta - betap) / betap
        k += 1
        bp[...] = b[...]
    return beta, b


# -------------------------------------------------------------- primal dual
def primal_dual_optimised(x, v, lam, psiH, psi, L, prox, l1weight, reweighter, grad,
                          nu=1.0, sigma=None, mask=None, tol=1e-5, maxit=1000,
                          positivity=1, report_freq=10, gamma=1.0, verbosity=1,
                          maxreweight=50, trace=None):
    """primal_dual.py:91-180.

    Naming trap kept from the reference: the 4th positional `psiH` receives the
    SYNTHESIS operator (Psi.hdot) and the 5th `psi` the ANALYSIS operator (Psi.dot)
    at the call site (workers/spotless.py:268-269).  `prox` is accepted and unused.
    Where the reference drops into pdb (x all zero -> eps = 1.0, NaN eps) we just
    carry on with the same values.
    """
    xp = x.copy()
    vp = v.copy()
    xout = np.zeros_like(x)        # NB: x and v are updated IN PLACE (out=x, psi(xp, v))
    if sigma is None:
        sigma = L / (2.0 * gamma) / nu
    tau = 0.9 / (L / (2.0 * gamma) + sigma * nu ** 2)

    eps = 1.0
    numreweight = 0
    k_exit = maxit
    for k in range(maxit):
        psi(xp, v)                                        # analysis into v
        _prox.dual_update_numba(vp, v, lam, sigma=sigma, weight=l1weight)
        vp[...] = 2.0 * v - vp                            # primal_dual.py:137
        psiH(vp, xout)                                    # synthesis into xout
        xout += grad(xp)
        x[...] = xp - tau * xout                          # primal_dual.py:140
        if positivity == 1:
            x[x < 0.0] = 0.0
        elif positivity == 2:
            msk = np.any(x <= 0, axis=0)
            x[:, msk] = 0.0
        if x.any():
            eps = norm_diff(x, xp)
        else:
            eps = 1.0
        if trace is not None:
            trace.append(float(eps))
        if eps < tol:
            if reweighter is not None and numreweight < maxreweight:
                l1weight = reweighter(x)
                numreweight += 1
            else:
                k_exit = k
                break
        xp[...] = x[...]
        vp[...] = v[...]
    return x, v


def primal_dual(x, v, lam, psi, psiH, L, prox, grad, nu=1.0, sigma=None, mask=None, tol=1e-5, maxit=1000, minit=10,
                positivity=1, report_freq=10, gamma=1.0, verbosity=1):
    """primal_dual.py:12-87 -- the un-optimised functional form: psiH(x) RETURNS the analysis coefficients, psi(v)
    the synthesised image, prox(v, sigma) is a callable; stopping rule `(eps > tol or k < minit) and k < maxit`,
    eps = |x - xp| / |x| (no 1e-12 guard here).  Where the reference drops into pdb (NaN / inf eps) we carry on."""
    xp = x.copy()
    vp = v.copy()
    if sigma is None:
        sigma = L / (2.0 * gamma) / nu
    tau = 0.9 / (L / (2.0 * gamma) + sigma * nu ** 2)
    eps = 1.0
    k = 0
    while (eps > tol or k < minit) and k < maxit:
        vtilde = v + sigma * psiH(xp)
        v = vtilde - sigma * prox(vtilde / sigma, lam / sigma)
        x = xp - tau * (psi(2 * v - vp) + grad(xp))
        if positivity == 1:
            x[x < 0.0] = 0.0
        elif positivity == 2:
            msk = np.any(x <= 0, axis=0)
            x[:, msk] = 0.0
        eps = np.linalg.norm(x - xp) / np.linalg.norm(x)
        xp[...] = x[...]
        vp[...] = v[...]
        k += 1
    return x, v
