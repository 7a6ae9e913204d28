"""
Primal-dual (Condat-Vu) backward step on MI355X -- drop-in for
pfb/opt/primal_dual.py:91-180 (primal_dual_optimised).

Per iteration (reference statement -> device work):
    psi(xp, v)                         pfb_psi_dot        (analysis, all bands/bases)
    dual_update_numba(vp, v, ...)      pfb_dual_update
    vp = 2 v - vp ; psiH(vp, xout)     pfb_psi_hdot(v) -> s, and xout = 2 s - s_prev inside the primal update:
                                       the synthesis is LINEAR, psi^H(2 v - vp) = 2 psi^H(v) - psi^H(vp), and psi^H(vp)
                                       is the previous iteration's psi^H(v) -- the cube 2 v - vp (primal_dual.py:137)
                                       is never written nor read (-0.68 GB of 2.1 GB per iteration at config #4)
    xout += grad(xp)                   caller's callable (one PSF convolution); a PsfGradient object (below) hands
                                       over conv(x) and the data separately and the subtraction is fused as well
    x = xp - tau xout ; positivity     pfb_pd_primal_update2 (+ norm_diff sums + any(x))
    eps = norm_diff(x, xp)             from the same kernel's sums
Everything stays on the GPU; three scalars per iteration come back for the stopping rule.

Naming trap kept from the reference: the 4th positional `psiH` receives the SYNTHESIS
operator (Psi.hdot), the 5th `psi` the ANALYSIS operator (Psi.dot) at the call site
(workers/spotless.py:268-269); `prox` is accepted and unused.  Like the reference the
iteration updates `x` and `v` IN PLACE and returns them.  Where the reference drops into
pdb (x all zero, NaN eps) this returns with the state as is and a warning.

Extension `group`: a torch.distributed process group over which the BAND axis is sharded
(SURVEY 8e).  x, v, grad then hold this rank's bands; the band sum inside the dual update is
all-reduced as one (nbasis, nymax, nxmax) plane per iteration, the norm_diff sums and the
any(x) flag as three scalars, and positivity=2 (a pixel is zeroed in EVERY band if it is
non-positive in ANY band) all-reduces its mask.  Every rank takes the same decisions.
"""
import math
import sys

import torch

from .. import _lib, _dev
from ..prox.prox_21m import dual_update_numba


class PsfGradient:
    """grad(x) = psf_convolve_cube(x) - data, the gradient workers/spotless.py:259-260 hands to
    primal_dual_optimised, as an object: callable like the closure it replaces, and recognised by
    primal_dual_optimised, which then takes conv(x) and `data` separately and subtracts inside the primal update
    (one pass over the cube less).  plan: a PsfConvPlan (or psfhat + nx, ny, lastsize through plan_for)."""

    def __init__(self, plan, data):
        self.plan = plan
        self.data = _dev.to_dev(data, plan.rdtype).contiguous()
        self._out = torch.empty_like(self.data)

    def conv(self, x):
        return self.plan.apply(x, out=self._out)

    def __call__(self, x):
        xd = _dev.to_dev(x, self.plan.rdtype)
        res = self.conv(xd) - self.data
        return res.cpu().numpy() if _dev.is_numpy(x) else res


def primal_dual_optimised(x, v, lam, psiH, psi, L, prox, l1weight, reweighter, grad,
                          nu=1.0, sigma=None, mask=None, tol=1e-5, maxit=1000, positivity=1,
                          report_freq=10, gamma=1.0, verbosity=1, maxreweight=50, group=None):
    lib = _lib.load()
    if group is not None:
        import torch.distributed as dist
        pg = None if group is True else group
    as_numpy = _dev.is_numpy(x)
    xd = _dev.to_dev(x).contiguous()
    vd = _dev.to_dev(v, xd.dtype).contiguous()
    dt = xd.dtype
    code = _dev.code(dt)
    nband = xd.shape[0]
    npix = xd[0].numel()
    xp = xd.clone()
    vp = vd.clone()
    # The LINEAR-synthesis form (2 psi^H(v) - psi^H(vp), one extra synthesis before the loop, outputs overwritten) is
    # only taken when `psiH` is this package's Psi.hdot, which is exactly linear and overwrites its output.  Any other
    # callable -- a masked, clipped or otherwise affine synthesis is legal in this signature -- gets the reference's
    # statement order verbatim: vp = 2 v - vp; psiH(vp, xout) (primal_dual.py:137-138).
    linear_syn = getattr(psiH, '__self__', None).__class__.__name__ == 'Psi' and getattr(psiH, '__name__', '') == 'hdot' \
        and getattr(psiH, '__module__', '').startswith('pfb_clean_amd')
    # s_new = psi^H(v) of this iteration, s_old = psi^H(vp) = psi^H(v) of the previous one (linear synthesis)
    s_new = torch.zeros_like(xd)
    s_old = torch.zeros_like(xd)
    if linear_syn:
        psiH(vp, s_old)
    fused_grad = linear_syn and isinstance(grad, PsfGradient) and not as_numpy and grad.data.shape == xd.shape \
        and grad.data.dtype == dt
    w = _dev.to_dev(l1weight, dt).contiguous()
    ws, out = _dev.scratch()

    if sigma is None:
        sigma = L / (2.0 * gamma) / nu
    tau = 0.9 / (L / (2.0 * gamma) + sigma * nu ** 2)

    def host(t):
        return t.cpu().numpy() if as_numpy else t

    # Buffer rotation instead of the reference's end-of-iteration copies `xp = x.copy()`,
    # `vp = v.copy()` (primal_dual.py:176-177): the freshly written x / v simply become the next
    # iteration's xp / vp.  For v this is exact only if (a) `psi` overwrites the whole coefficient
    # support (ours does) and (b) the margins of the packed layout, which psi never writes, are zero
    # in the caller's v (then they stay zero in both buffers); otherwise the copy is kept.
    rotate_v = False
    if getattr(psi, '__self__', None).__class__.__name__ == 'Psi' and getattr(psi, '__name__', '') == 'dot':
        mark = torch.full_like(vd, float('nan'))
        psi(xp, mark)
        rotate_v = not bool(torch.any(vd[torch.isnan(mark)] != 0).item())
        del mark

    eps = 1.0
    numreweight = 0
    k = 0
    xn, vn = xd, vd                               # where the newest iterate lives
    # The host looks at three scalars per iteration (stopping rule, :150-166).  With rotating buffers on the device the
    # NEXT iteration's analysis psi(x_new) -- which depends on nothing the host decides -- is enqueued BEFORE that look,
    # and the look waits for an event recorded right behind the scalars' copy instead of for the whole stream: the
    # device runs the analysis while the host reads, decides and enqueues the rest.  A converged (or failed) solve has
    # then run one analysis too many, into the coefficient buffer that is not returned.
    lookahead = rotate_v and group is None and not as_numpy and xd.is_cuda
    if lookahead:
        host_out = torch.empty(3, dtype=torch.float64).pin_memory()
        ev = torch.cuda.Event()
    prefetched = False
    for k in range(maxit):
        if k > 0:                                 # :176-177 of the previous iteration
            xp, xn = xn, xp
            if rotate_v:
                vp, vn = vn, vp
            else:
                vp.copy_(vn)
        if k > 0:
            s_old, s_new = s_new, s_old
        if not prefetched:
            psi(xp, vn)                                                  # :135
        prefetched = False
        dual_update_numba(vp, vn, lam, sigma=sigma, weight=w, group=group)   # :136
        pos_arg = 0 if (group is not None and positivity == 2) else int(positivity)
        if linear_syn:
            psiH(vn, s_new)                                              # :137-138 as 2 psiH(v) - psiH(vp)
            if fused_grad:
                gd, gsub = grad.conv(xp), grad.data                      # :139, `- data` inside the update
            else:
                gd, gsub = _dev.to_dev(grad(host(xp)), dt).contiguous(), None
            _lib.check(lib.pfb_pd_primal_update2(code, _dev.ptr(xp), _dev.ptr(s_new), _dev.ptr(s_old), _dev.ptr(gd),
                                                 _dev.ptr(gsub), float(tau), pos_arg, nband, npix,
                                                 _dev.ptr(xn), _dev.ptr(out), _dev.ptr(ws),
                                                 _dev.stream()))         # :140-146
        else:
            # the reference's own statements: vp = 2 v - vp (vp is dead afterwards: re-set at the top of the next
            # iteration), psiH(vp, xout) with whatever the caller's synthesis does to its arguments, xout += grad(xp)
            _lib.check(lib.pfb_axpby(code, 2.0, _dev.ptr(vn), -1.0, _dev.ptr(vp), vp.numel(), _dev.stream()))   # :137
            if as_numpy:
                sh = s_new.cpu().numpy()
                psiH(host(vp), sh)
                s_new.copy_(torch.from_numpy(sh))
            else:
                psiH(vp, s_new)                                          # :138
            gd = _dev.to_dev(grad(host(xp)), dt).contiguous()            # :139
            _lib.check(lib.pfb_pd_primal_update(code, _dev.ptr(xp), _dev.ptr(s_new), _dev.ptr(gd), float(tau), pos_arg,
                                                nband, npix, _dev.ptr(xn), _dev.ptr(out), _dev.ptr(ws),
                                                _dev.stream()))          # :140-146
        if group is not None:
            if positivity == 2:
                bad = (xn <= 0).any(dim=0).to(torch.uint8)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=pg)
                xn.mul_((1 - bad).to(dt)[None])
                _lib.check(lib.pfb_norm_diff_sums(code, _dev.ptr(xn), _dev.ptr(xp), xn.numel(),
                                                  _dev.ptr(out), _dev.ptr(ws), _dev.stream()))
                out[2] = (xn != 0).any().to(out.dtype)
            dist.all_reduce(out[:3], op=dist.ReduceOp.SUM, group=pg)
        if lookahead:
            host_out.copy_(out[:3], non_blocking=True)
            ev.record()
            if k + 1 < maxit:
                psi(xn, vp)                   # next iteration's :135 (its xp is this xn, its vn this vp buffer)
                prefetched = True
            ev.synchronize()
            num, den, anyx = host_out.tolist()
        else:
            num, den, anyx = out[:3].tolist()
        if anyx:
            eps = math.sqrt(num / (1e-12 + den))
        else:
            print("primal_dual: x is identically zero (the reference stops in pdb here)",
                  file=sys.stderr)
            eps = 1.0
        if eps < tol:
            if reweighter is not None and numreweight < maxreweight:
                w = _dev.to_dev(reweighter(host(xn)), dt).contiguous()
                numreweight += 1
            else:
                if numreweight >= maxreweight and verbosity:
                    print("Maximum reweighting steps reached", file=sys.stderr)
                break
        if math.isnan(eps) or math.isinf(eps):
            print("primal_dual: non-finite eps (the reference stops in pdb here)", file=sys.stderr)
            break
        if not k % report_freq and verbosity > 1:
            print(f"At iteration {k} eps = {eps:.3e}", file=sys.stderr)

    if verbosity:
        if k == maxit - 1:
            print(f"Max iters reached. eps = {eps:.3e}", file=sys.stderr)
        else:
            print(f"Success, converged after {k} iterations", file=sys.stderr)

    if as_numpy:
        x[...] = xn.cpu().numpy()
        v[...] = vn.cpu().numpy()
        return x, v
    if xn is not x:
        x.copy_(xn)
    if vn is not v:
        v.copy_(vn)
    return x, v


def primal_dual(x, v, lam, psi, psiH, L, prox, grad, nu=1.0, sigma=None, mask=None, tol=1e-5, maxit=1000,
                minit=10, positivity=1, report_freq=10, gamma=1.0, verbosity=1):
    """pfb/opt/primal_dual.py:12-87 -- the un-optimised, functional form of the same iteration (the one
    workers/fwdbwd.py:367 names): `psiH(x)` RETURNS the analysis coefficients, `psi(v)` RETURNS the synthesised image,
    `prox(v, sigma)` is any callable (prox_21m of this package keeps it on the device), `grad(x)` the smooth term.
    Same positional order as the reference (note psi / psiH swapped w.r.t. primal_dual_optimised), stopping rule
    `(eps > tol or k < minit) and k < maxit`, eps = |x - xp| / |x| (no 1e-12 guard in this variant).
    The combinations between the operator calls run in the library's vector kernels (pfb_axpby) and the primal update
    with positivity and the two norms in pfb_pd_primal_update; the callables see GPU tensors when x is a tensor, numpy
    arrays when x is numpy (drop-in: every call then crosses PCIe).  Returns NEW x, v like the reference."""
    lib = _lib.load()
    as_numpy = _dev.is_numpy(x)
    xd = _dev.to_dev(x).contiguous()
    dt = xd.dtype
    code = _dev.code(dt)
    vd = _dev.to_dev(v, dt).contiguous()
    nband, npix = xd.shape[0], xd[0].numel()
    ws, out = _dev.scratch()
    if sigma is None:
        sigma = L / (2.0 * gamma) / nu
    tau = 0.9 / (L / (2.0 * gamma) + sigma * nu ** 2)

    def host(t):
        return t.cpu().numpy() if as_numpy else t

    def dev(a, private=False):                    # result of a caller's operator -> contiguous device tensor
        t = _dev.to_dev(a, dt).contiguous()       # private: we write into it -- never into a buffer the caller may own
        return t.clone() if (private and isinstance(a, torch.Tensor)) or t.data_ptr() in keep else t

    def axpby(a, u, b, w):                        # w = a*u + b*w
        _lib.check(lib.pfb_axpby(code, float(a), _dev.ptr(u), float(b), _dev.ptr(w), w.numel(), _dev.stream()))

    xp = xd.clone()
    vp = vd.clone()
    vcur = vd.clone()
    xnew = torch.empty_like(xd)
    keep = set()
    eps, k = 1.0, 0
    while (eps > tol or k < minit) and k < maxit:
        keep = {xp.data_ptr(), vp.data_ptr(), vcur.data_ptr()}
        vt = dev(psiH(host(xp)), private=True)                    # psiH(xp)
        axpby(1.0, vcur, sigma, vt)                               # vtilde = v + sigma psiH(xp)            :49
        arg = vt.clone()
        axpby(0.0, vt, 1.0 / sigma, arg)                          # vtilde / sigma
        pr = dev(prox(host(arg), lam / sigma))
        axpby(-sigma, pr, 1.0, vt)                                # v = vtilde - sigma prox(vtilde/sigma)  :52
        vcur, vt = vt, vcur
        axpby(2.0, vcur, -1.0, vp)                                # vp <- 2 v - vp (vp is re-set below)    :55
        so = dev(psi(host(vp)))
        g = dev(grad(host(xp)))
        _lib.check(lib.pfb_pd_primal_update(code, _dev.ptr(xp), _dev.ptr(so), _dev.ptr(g), float(tau),
                                            int(positivity), nband, npix, _dev.ptr(xnew), _dev.ptr(out),
                                            _dev.ptr(ws), _dev.stream()))                         # :55-60
        num, den, _ = out[:3].tolist()
        eps = math.sqrt(num) / math.sqrt(den) if den > 0 else float('nan')                        # :63
        xp.copy_(xnew)
        vp.copy_(vcur)
        if math.isnan(eps) or math.isinf(eps):
            print("primal_dual: non-finite eps (the reference stops in pdb here)", file=sys.stderr)
            break
        if not k % report_freq and verbosity > 1:
            print(f"At iteration {k} eps = {eps:.3e}", file=sys.stderr)
        k += 1
    if verbosity:
        if k == maxit:
            print(f"Max iters reached. eps = {eps:.3e}", file=sys.stderr)
        else:
            print(f"Success, converged after {k} iterations", file=sys.stderr)
    if as_numpy:
        return xp.cpu().numpy(), vcur.cpu().numpy()
    return xp.clone(), vcur
