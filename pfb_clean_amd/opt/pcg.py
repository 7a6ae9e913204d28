"""
Preconditioned conjugate gradients on MI355X -- drop-in for pfb/opt/pcg.py.

    pcg(A, b, x0=None, M=None, tol=1e-5, maxit=500, minit=100, verbosity=1,
        report_freq=10, backtrack=True, return_resid=False)          pcg.py:53-136
    pcg_psf(psfhat, b, x0, beam, lastsize, nthreads, sigmainv, cgopts, compute=True)
                                                                      pcg.py:243-360

Two execution paths with identical semantics (residual r = A x - b, stopping rule
`(eps > tol or k < minit) and k < maxit`, backtracking without extra matvecs,
break-before-increment on an all-zero direction, "Initial residual is zero" returns x0):

 * FUSED: when `A` is a HessianPsf (or a functools.partial of this package's
   hessian_psf_cube / _hessian_psf_slice, which is exactly what the reference workers
   build -- fluxmop.py:160-174, pcg.py:276-284) and `M` is None or a DivPrecond, the
   whole solve runs inside libpfb_hip.so (pfb_pcg_solve): 3 convolution kernels + 3
   fused vector kernels per iteration, device-resident scalars.
 * GENERIC: any Python callables A, M.  Vector arithmetic and reductions still run in
   the HIP kernels (pfb_dot, pfb_axpby, pfb_norm_diff_sums, pfb_any_nonzero); A and M
   are called with the same array kind (numpy / tensor) the caller passed for `b`.
"""
import ctypes as C
import functools
import math
import os
import sys
import threading

import torch

from .. import _lib, _dev
from ..operators import hessian as _hess
from ..operators.hessian import HessianPsf


class DivPrecond:
    """M(x) = x / div -- the diagonal preconditioner of pcg.py:264-267."""

    def __init__(self, div):
        self.div = float(div)

    def __call__(self, x):
        return x / self.div


def _detect_div_precond(M, b):
    """The workers write the preconditioner as a closure, `lambda x: x / sigmainv` (pcg.py:264-267,
    fluxmop.py).  A closure cannot be inspected, but it can be ASKED: M is applied once to b itself and
    to -2 b; if both answers are b / c and -2 b / c for one scalar c (elementwise, to rounding), M is the
    scalar division and the solve stays on the fused path.  Anything else (a diagonal with varying
    entries, an operator, something non-linear) fails the elementwise test and keeps the generic path."""
    try:
        bd = _dev.to_dev(b)
        host = (lambda t: t.cpu().numpy()) if _dev.is_numpy(b) else (lambda t: t)
        y1 = _dev.to_dev(M(host(bd)), bd.dtype)
        if y1.shape != bd.shape:
            return None
        flat_b, flat_y = bd.reshape(-1), y1.reshape(-1)
        k = int(torch.argmax(flat_b.abs()).item())
        if flat_b[k].item() == 0.0 or flat_y[k].item() == 0.0:
            return None
        c = flat_b[k].item() / flat_y[k].item()
        eps = 16 * torch.finfo(bd.dtype).eps
        scale = flat_b.abs().max().item() / abs(c)
        if (flat_y - flat_b / c).abs().max().item() > eps * scale:
            return None
        y2 = _dev.to_dev(M(host(-2.0 * bd)), bd.dtype).reshape(-1)
        if (y2 + 2.0 * flat_b / c).abs().max().item() > 2 * eps * scale:
            return None
        return DivPrecond(c)
    except Exception:
        return None


def _log(msg, verbosity, level=1):
    if verbosity >= level:
        print(msg, file=sys.stderr)


def _as_hessian(A, b):
    """Recognise the operator objects / partials the fused driver can run."""
    if isinstance(A, HessianPsf):
        return A
    if isinstance(A, functools.partial) and not A.keywords.get('_nofuse', False):
        kw = dict(A.keywords)
        args = A.args
        nx, ny = b.shape[-2:]
        try:
            if A.func is _hess.hessian_psf_cube and len(args) == 6:
                _, _, _, beam, psfhat, lastsize = args
            elif A.func is _hess._hessian_psf_slice and len(args) == 6:
                _, _, _, psfhat, beam, lastsize = args
            else:
                return None
        except ValueError:
            return None
        return HessianPsf(psfhat, nx, ny, lastsize, beam=beam,
                          sigmainv=kw.get('sigmainv', 1), wsum=kw.get('wsum', None))
    return None


def _backtrack_mode(backtrack):
    """False -> 0; True -> 2 (predictive line search, same decisions as pcg.py:96-101 from
    three fused scalars) unless PFB_PCG_EXACT_BACKTRACK=1 or backtrack == 'exact' -> 1 (the
    reference loop verbatim, one extra vector pass per rejected step)."""
    if not backtrack:
        return 0
    if backtrack == 'exact' or os.environ.get('PFB_PCG_EXACT_BACKTRACK', '0') == '1':
        return 1
    return 2


class _Work:
    """Device scratch for pfb_pcg_solve, cached per (plan, nb, stream).  id(plan) can be re-used
    by a NEW plan object of a different size once the old one is garbage collected, so the cached
    buffer is only handed out if it has exactly the size this plan asks for."""
    _cache = {}

    @classmethod
    def get(cls, plan, nb):
        # per host thread: the reference may drive per-band solves from several dask threads
        # (pcg.py:346-356).  Solves on ONE plan are serialised by plan.lock (the plan's spectrum workspace and
        # dot partials are single-owner); the vectors of a solve still live in a per-thread scratch so that a
        # thread's result buffers are not overwritten by the next thread's solve
        key = (id(plan), nb, _dev.stream(), threading.get_ident())
        nbytes = _lib.load().pfb_pcg_work_bytes(plan.handle, nb)
        w = cls._cache.get(key)
        if w is None or w.numel() != nbytes or w.device != plan.device:
            if len(cls._cache) > 16:
                cls._cache.clear()
            w = cls._cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=plan.device)
        return w


def pcg_fused(A, b, x0=None, mdiv=0.0, tol=1e-5, maxit=500, minit=100, backtrack=True,
              return_resid=False, group=None, distributed=False):
    """Run pfb_pcg_solve.  b, x0: GPU tensors (nb, nx, ny) | (nx, ny).  Returns
    (x, r|None, PcgResult).  distributed=True: the bands of this call are one rank's
    shard of a cube solve; every inner product is all-reduced over `group`
    (pfb_clean_amd.dist.AllReduceHook, RCCL)."""
    lib = _lib.load()
    plan = A.plan
    squeeze = b.ndim == 2
    b3 = (b[None] if squeeze else b).contiguous()
    nb = b3.shape[0]
    if nb != A.nb:
        raise ValueError(f"b has {nb} bands, operator has {A.nb}")
    if b3.dtype != plan.rdtype:
        raise TypeError(f"b is {b3.dtype}, operator is {plan.rdtype}")
    x = torch.zeros_like(b3) if x0 is None else (x0[None] if squeeze else x0).contiguous().clone()
    beam = A.beam
    if plan.embed is not None:
        # Embedded plan (arbitrary size on the power-of-two kernels, operators/psf.py): solve in the
        # zero-padded domain with a beam that is ZERO outside the image.  There A' x' = sigmainv x',
        # b' = 0 and x0' = 0, so r, y, p stay exactly zero outside and every inner product, step
        # length and stopping decision equals the un-padded solve's.
        b3, x = plan._pad(b3, nb), plan._pad(x, nb)
        beam = plan._pad(torch.ones((nb, plan.nx, plan.ny), dtype=plan.rdtype, device=b3.device)
                         if beam is None else beam, nb)
    r = torch.empty_like(b3) if return_resid else None
    work = _Work.get(plan, nb)
    res = _lib.PcgResult()
    cb_ctx = None
    native = None
    if distributed:
        from ..dist import AllReduceHook, native_comm
        native = native_comm(group, b3.device) if b3.is_cuda else None
    if not distributed:
        cb = _lib.ALLREDUCE_FN(0)
    elif native is not None:             # RCCL from C on the solver's stream: nothing on the host per iteration
        cb, cb_ctx = native.fn, native.ctx
    else:
        allreduce = AllReduceHook(work, group)

        def _hook(ctx, buf, count, stream):
            try:
                allreduce(buf or 0, count)      # count == 0: probe, count < 0: abort (include/pfb_hip.h)
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                if count > 0:
                    print(f"pfb_clean_amd: allreduce hook failed: {e!r}", file=sys.stderr)
                return 1
        cb = _lib.ALLREDUCE_FN(_hook)
    with plan.lock:            # a plan is single-owner (include/pfb_hip.h): one solve at a time per plan
        plan._enter_stream()
        _lib.check(lib.pfb_pcg_solve(plan.handle, A.band0, nb, _dev.ptr(b3), _dev.ptr(x), _dev.ptr(r),
                                     _dev.ptr(beam), A.wsum if A.wsum is not None else 0.0,
                                     A.sigmainv, float(mdiv), float(tol), int(maxit), int(minit),
                                     _backtrack_mode(backtrack), _dev.ptr(work), cb, cb_ctx, C.byref(res),
                                     _dev.stream()))
    if distributed:                  # bench.py reports which exchange ran and what the hook costs the host
        res.exchange = 'rccl-native' if native is not None else 'torch-hook'
        res.hook_calls, res.hook_host_s = (0, 0.0) if native is not None else (allreduce.calls, allreduce.host_s)
    if plan.embed is not None:
        x = x[:, :plan.nx, :plan.ny].contiguous()
        r = None if r is None else r[:, :plan.nx, :plan.ny].contiguous()
    if squeeze:
        x = x[0]
        r = None if r is None else r[0]
    return x, r, res


def _report(res_status, k, eps, verbosity):
    if res_status == 'maxit':
        _log(f"Max iters reached. eps = {eps:.3e}", verbosity)
    elif res_status in ('converged', 'breakdown'):
        _log(f"Success, converged after {k} iterations", verbosity)


def pcg(A, b, x0=None, M=None, tol=1e-5, maxit=500, minit=100, verbosity=1,
        report_freq=10, backtrack=True, return_resid=False):
    """pfb/opt/pcg.py:53-136."""
    H = _as_hessian(A, b)
    if H is not None and M is not None and not isinstance(M, DivPrecond):
        M = _detect_div_precond(M, b) or M
    fusable_M = M is None or isinstance(M, DivPrecond)
    as_numpy = _dev.is_numpy(b)
    if H is not None and fusable_M:
        bd = _dev.to_dev(b)
        x0d = None if x0 is None else _dev.to_dev(x0, bd.dtype)
        x, r, res = pcg_fused(H, bd, x0d, mdiv=M.div if M is not None else 0.0, tol=tol,
                              maxit=maxit, minit=minit, backtrack=backtrack,
                              return_resid=return_resid)
        status = _lib.PCG_STATUS[res.status]
        if status == 'zero-residual':
            _log("Initial residual is zero", verbosity)
            # the reference returns x0 itself, and ONLY x0, even with return_resid
            return x0 if x0 is not None else (x.cpu().numpy() if as_numpy else x)
        _report(status, res.iters, res.eps, verbosity)
        if as_numpy:
            x = x.cpu().numpy()
            r = None if r is None else r.cpu().numpy()
        return (x, r) if return_resid else x
    return _pcg_generic(A, b, x0, M, tol, maxit, minit, verbosity, report_freq, backtrack,
                        return_resid)


def _pcg_generic(A, b, x0, M, tol, maxit, minit, verbosity, report_freq, backtrack,
                 return_resid):
    lib = _lib.load()
    as_numpy = _dev.is_numpy(b)
    bd = _dev.to_dev(b).contiguous()
    dt = bd.dtype
    code = _dev.code(dt)
    n = bd.numel()
    st = _dev.stream
    ws, out = _dev.scratch()

    def host(t):
        return t.cpu().numpy() if as_numpy else t

    def dev(a):
        return _dev.to_dev(a, dt).contiguous()

    def callA(v):
        return dev(A(host(v))).clone() if not as_numpy else dev(A(host(v)))

    def callM(v):
        return v if M is None else dev(M(host(v)))

    def dot(u, v):
        _lib.check(lib.pfb_dot(code, _dev.ptr(u), _dev.ptr(v), n, _dev.ptr(out), _dev.ptr(ws), st()))
        return out[0].item()

    def anynz(u):
        _lib.check(lib.pfb_any_nonzero(code, _dev.ptr(u), n, _dev.ptr(out), _dev.ptr(ws), st()))
        return out[0].item() != 0.0

    def axpby(a, u, bb, v):       # v = a*u + bb*v
        _lib.check(lib.pfb_axpby(code, float(a), _dev.ptr(u), float(bb), _dev.ptr(v), n, st()))

    x0_in = x0
    x = torch.zeros_like(bd) if x0 is None else dev(x0).clone()
    r = callA(x)
    axpby(-1.0, bd, 1.0, r)                          # r = A(x0) - b
    y = callM(r)
    if not anynz(y):
        _log("Initial residual is zero", verbosity)
        return x0_in if x0_in is not None else host(x)
    p = y.clone()
    axpby(0.0, y, -1.0, p)                           # p = -y
    k = 0
    eps = 1.0
    xp = torch.empty_like(x)
    rp = torch.empty_like(r)
    broke = False
    while (eps > tol or k < minit) and k < maxit:
        xp.copy_(x)
        rp.copy_(r)
        Ap = callA(p)
        rnorm = dot(r, y)
        alpha = rnorm / dot(p, Ap)
        x.copy_(xp); axpby(alpha, p, 1.0, x)
        r.copy_(rp); axpby(alpha, Ap, 1.0, r)
        y = callM(r)
        rnorm_next = dot(r, y)
        while rnorm_next > rnorm and backtrack:
            alpha *= 0.75
            x.copy_(xp); axpby(alpha, p, 1.0, x)
            r.copy_(rp); axpby(alpha, Ap, 1.0, r)
            y = callM(r)
            rnorm_next = dot(r, y)
        beta = rnorm_next / rnorm
        axpby(-1.0, y, beta, p)                      # p = beta*p - y
        if not anynz(p):
            broke = True
            break
        k += 1
        _lib.check(lib.pfb_norm_diff_sums(code, _dev.ptr(x), _dev.ptr(xp), n, _dev.ptr(out),
                                          _dev.ptr(ws), st()))
        num, den = out[:2].tolist()
        eps = math.sqrt(num / (1e-12 + den))
        if not k % report_freq and verbosity > 1:
            _log(f"At iteration {k} epsx = {eps:.3e}", verbosity, 2)
    _report('maxit' if (k >= maxit and not broke) else 'converged', k, eps, verbosity)
    if not return_resid:
        return host(x)
    return host(x), host(r)


def cg_dct(A, b, x, tol=1e-5, maxit=500, verbosity=1, report_freq=10):
    """pfb/opt/pcg.py:139-239 -- plain CG whose unknown is a nested dict
    {field: {'t..b..': image}} (distinct fields, no preconditioner, no backtracking; unused by
    the live workers).  r = A x - b, eps = <r, r> over all leaves, rule
    `eps > tol and k < maxit`.  Leaves live on the GPU for the whole solve; `A` is called with a
    dict of GPU tensors when the caller's leaves are tensors, of numpy arrays otherwise, and like
    the reference the caller's `x` leaves are updated in place.  Returns (x, r)."""
    lib = _lib.load()
    as_numpy = _dev.is_numpy(next(iter(next(iter(b.values())).values())))
    ws, out = _dev.scratch()
    st = _dev.stream

    def todev(d):
        return {f: {i: _dev.to_dev(d[f][i]).contiguous() for i in d[f]} for f in d}

    def callA(d):
        if not as_numpy:
            return todev(A(d))
        return todev(A({f: {i: d[f][i].cpu().numpy() for i in d[f]} for f in d}))

    def vdot(u, v):
        acc = 0.0
        for f in u:
            for i in u[f]:
                t = u[f][i]
                _lib.check(lib.pfb_dot(_dev.code(t.dtype), _dev.ptr(t), _dev.ptr(v[f][i]), t.numel(),
                                       _dev.ptr(out), _dev.ptr(ws), st()))
                acc += out[0].item()
        return acc

    def axpby(a, u, bb, v):       # v = a*u + bb*v, leaf by leaf
        for f in v:
            for i in v[f]:
                t = v[f][i]
                _lib.check(lib.pfb_axpby(_dev.code(t.dtype), float(a), _dev.ptr(u[f][i]), float(bb),
                                         _dev.ptr(t), t.numel(), st()))

    xd = todev(x)
    if not as_numpy:                 # tensors already on the device are updated in place
        xd = {f: {i: (x[f][i] if (x[f][i].is_cuda and x[f][i].is_contiguous()) else xd[f][i]) for i in x[f]}
              for f in x}
    bd = todev(b)
    r = callA(xd)
    r = {f: {i: r[f][i].clone() for i in r[f]} for f in r}
    axpby(-1.0, bd, 1.0, r)                                   # r = A x - b
    p = {f: {i: -r[f][i] for i in r[f]} for f in r}
    rnorm = vdot(r, r)
    eps, k = rnorm, 0
    while eps > tol and k < maxit:
        Ap = callA(p)
        alpha = rnorm / vdot(p, Ap)
        axpby(alpha, p, 1.0, xd)                              # x += alpha p
        axpby(alpha, Ap, 1.0, r)                              # r += alpha Ap
        rnorm_next = vdot(r, r)
        beta = rnorm_next / rnorm
        axpby(-1.0, r, beta, p)                               # p = beta p - r
        rnorm = rnorm_next
        eps = rnorm
        k += 1
        if not k % report_freq and verbosity > 1:
            _log(f"At iteration {k} eps = {eps:.3e}", verbosity, 2)
    if verbosity:
        _log(f"Max iters reached. eps = {eps:.3e}" if k >= maxit
             else f"Success, converged after {k} iterations", verbosity)
    for f in x:                                               # in-place semantics of the reference
        for i in x[f]:
            if as_numpy:
                x[f][i][...] = xd[f][i].cpu().numpy()
            elif x[f][i] is not xd[f][i]:
                x[f][i].copy_(xd[f][i])
    if as_numpy:
        r = {f: {i: r[f][i].cpu().numpy() for i in r[f]} for f in r}
    return x, r


def pcg_dist(A, maxit, minit, tol, sigmainv):
    """pfb/opt/pcg.py:363-420 -- the per-band PCG variant of the (commented-out) distributed
    spotless: b = A.residual/A.wsum (A.dirty without a residual), x0 = 0, M = x/sigmainv,
    eps = rnorm/eps0 with eps0 the INITIAL <r, y>, stall counter in the stopping rule.  A is a
    hessian_psf_slice (or anything with those attributes, callable on GPU tensors).  All vectors
    stay on the GPU; the matvec is the fused convolution, the recurrences are pfb_axpby /
    pfb_dot / pfb_any_nonzero; the scalars the while-conditions need come back per iteration
    (this path keeps the reference's exact backtracking loop)."""
    lib = _lib.load()
    src = A.residual if hasattr(A, 'residual') else A.dirty
    as_numpy = _dev.is_numpy(src)
    b = _dev.to_dev(src).contiguous() / float(A.wsum)
    dt = b.dtype
    code = _dev.code(dt)
    n = b.numel()
    st = _dev.stream
    ws, out = _dev.scratch()

    def dot(u, v):
        _lib.check(lib.pfb_dot(code, _dev.ptr(u), _dev.ptr(v), n, _dev.ptr(out), _dev.ptr(ws), st()))
        return out[0].item()

    def anynz(u):
        _lib.check(lib.pfb_any_nonzero(code, _dev.ptr(u), n, _dev.ptr(out), _dev.ptr(ws), st()))
        return out[0].item() != 0.0

    def axpby(a, u, bb, v):       # v = a*u + bb*v
        _lib.check(lib.pfb_axpby(code, float(a), _dev.ptr(u), float(bb), _dev.ptr(v), n, st()))

    def callA(v):
        return _dev.to_dev(A(v), dt).contiguous().clone()

    x = torch.zeros_like(b)
    r = callA(x)
    axpby(-1.0, b, 1.0, r)                           # r = A(x) - b
    y = r / sigmainv
    p = -y
    rnorm = dot(r, y)
    eps0 = 1.0 if (math.isnan(rnorm) or rnorm == 0.0) else rnorm
    k, eps, stall = 0, 1.0, 0
    xp, rp = torch.empty_like(x), torch.empty_like(r)
    while (eps > tol or k < minit) and k < maxit and stall < 5:
        xp.copy_(x)
        rp.copy_(r)
        epsp = eps
        Ap = callA(p)
        rnorm = dot(r, y)
        alpha = rnorm / dot(p, Ap)
        while True:
            x.copy_(xp); axpby(alpha, p, 1.0, x)
            r.copy_(rp); axpby(alpha, Ap, 1.0, r)
            y = r / sigmainv
            rnorm_next = dot(r, y)
            if not rnorm_next > rnorm:
                break
            alpha *= 0.75
        beta = rnorm_next / rnorm
        axpby(-1.0, y, beta, p)                      # p = beta*p - y
        if not anynz(p):
            break
        rnorm = rnorm_next
        k += 1
        eps = rnorm / eps0
        if abs(eps - epsp) < 1e-3 * tol:
            stall += 1
    print(f'Band={getattr(A, "bandid", "?")}, iters{k}, eps={eps}', file=sys.stderr)
    return x.cpu().numpy() if as_numpy else x


def cg(A, b, x0=None, tol=1e-5, maxit=500, verbosity=1, report_freq=10):
    """pfb/opt/pcg.py:12-50 -- the plain CG variant (no preconditioner, no backtracking,
    stopping rule `eps = <r,r> > tol`, unused by the live workers).  Vector work and
    reductions in the HIP kernels; A is called with the array kind of `b`."""
    lib = _lib.load()
    as_numpy = _dev.is_numpy(b)
    bd = _dev.to_dev(b).contiguous()
    dt, n = bd.dtype, bd.numel()
    code = _dev.code(dt)
    ws, out = _dev.scratch()

    def host(t):
        return t.cpu().numpy() if as_numpy else t

    def dot(u, v):
        _lib.check(lib.pfb_dot(code, _dev.ptr(u), _dev.ptr(v), n, _dev.ptr(out), _dev.ptr(ws), _dev.stream()))
        return out[0].item()

    def axpby(a, u, bb, v):
        _lib.check(lib.pfb_axpby(code, float(a), _dev.ptr(u), float(bb), _dev.ptr(v), n, _dev.stream()))

    x = torch.zeros_like(bd) if x0 is None else _dev.to_dev(x0, dt).contiguous().clone()
    r = _dev.to_dev(A(host(x)), dt).contiguous().clone()
    axpby(-1.0, bd, 1.0, r)                          # r = A(x) - b
    p = r.clone()
    axpby(0.0, r, -1.0, p)                           # p = -r
    rnorm = dot(r, r)
    eps, k = rnorm, 0
    while eps > tol and k < maxit:
        Ap = _dev.to_dev(A(host(p)), dt).contiguous()
        alpha = rnorm / dot(p, Ap)
        axpby(alpha, p, 1.0, x)
        axpby(alpha, Ap, 1.0, r)
        rnorm_next = dot(r, r)
        beta = rnorm_next / rnorm
        axpby(-1.0, r, beta, p)                      # p = beta*p - r
        rnorm = rnorm_next
        eps = rnorm
        k += 1
    if k >= maxit and verbosity:
        print(f"Max iters reached eps = {eps}", file=sys.stderr)
    return host(x)


def pcg_psf(psfhat, b, x0, beam, lastsize, nthreads, sigmainv, cgopts, compute=True):
    """pfb/opt/pcg.py:310-360 (+ _pcg_psf_impl :243-291): independent PCG per band with
    A = _hessian_psf_slice(psfhat[k], beam[k], sigmainv) and M = x/sigmainv when
    sigmainv > 0.  The reference's dask blockwise-over-bands becomes a loop of fused
    device solves on one plan (bands are independent: shard them over GPUs with
    pfb_clean_amd.dist.shard_bands for multi-GPU)."""
    as_numpy = _dev.is_numpy(b)
    bd = _dev.to_dev(b)
    nband, nx, ny = bd.shape
    if psfhat.shape[0] != nband:
        raise ValueError("psfhat and b disagree on the number of bands")
    beamd = None
    if beam is not None:
        beamd = _dev.to_dev(beam, bd.dtype)
        if beamd.ndim == 2:
            beamd = beamd[None]
        if beamd.shape[0] == 1:
            beamd = beamd.expand(nband, -1, -1)
        elif beamd.shape[0] != nband:
            raise ValueError('Beam has incorrect shape')
    x0d = torch.zeros_like(bd) if x0 is None else _dev.to_dev(x0, bd.dtype)
    from ..operators.psf import plan_for
    plan = plan_for(psfhat, nx, ny, lastsize)
    model = torch.zeros_like(bd)
    opts = dict(cgopts)
    verbosity = opts.pop('verbosity', 1)
    opts.pop('report_freq', None)
    mdiv = sigmainv if sigmainv > 0 else 0.0
    for k in range(nband):
        A = HessianPsf(plan, nx, ny, lastsize, beam=None if beamd is None else beamd[k:k + 1],
                       sigmainv=sigmainv, band0=k, nb=1)
        x, _, res = pcg_fused(A, bd[k:k + 1], x0d[k:k + 1], mdiv=mdiv, **opts)
        status = _lib.PCG_STATUS[res.status]
        if status == 'zero-residual':
            _log("Initial residual is zero", verbosity)
            model[k] = x0d[k]
        else:
            _report(status, res.iters, res.eps, verbosity)
            model[k] = x[0]
    return model.cpu().numpy() if as_numpy else model
