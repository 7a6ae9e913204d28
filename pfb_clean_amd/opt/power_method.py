"""
Power method for the spectral norm of the (PSF-approximated) Hessian -- drop-in for
pfb/opt/power_method.py:11-49.  One matvec per iteration and ONE host look at the three inner products
<b,b>, <bp,b>, <bp,bp> (they are left in device scalars and read back together); when `A` is this package's
HessianPsf (or the functools.partial of hessian_psf_cube / _hessian_psf_slice the workers build) <bp,b> and
<b,b> come fused out of the convolution's epilogue (pfb_psfconv_apply_dots) instead of two more passes.
`A` may return an aliased buffer (it is normalised into `bp` straight away, power_method.py:29-36).  With b0=None the start vector is drawn with numpy's global
RNG exactly like the reference (np.random.randn, fp64); `dtype` (numpy or torch dtype,
an extension) casts it and, when a torch dtype, keeps the iteration on device tensors.
A is called with the same array kind (numpy / GPU tensor) as the start vector.

Extension `group` (the reference's power_method_dist, power_method.py:70-116, sums the same three numbers
over its dask workers): a torch.distributed process group over which the BAND axis is sharded.  `A`, `imsize` / `b0`
then describe this rank's bands; <bp,b>, <bp,bp>, <b,b> are all-reduced (three scalars per iteration, one collective)
and every rank returns the same beta together with its own bands of the eigenvector.
"""
import math
import sys

import numpy as np
import torch

from .. import _lib, _dev


def power_method(A, imsize, b0=None, tol=1e-5, maxit=250, verbosity=1, report_freq=25,
                 dtype=None, group=None):
    lib = _lib.load()
    if group is not None:
        import torch.distributed as dist
        pg = None if group is True else group

    def allsum(vals):                   # sum of a short list of host scalars over the band shards
        if group is None:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=bd.device if dist.get_backend(pg) != 'gloo' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
        return t.tolist()

    if b0 is None:
        b0 = np.random.randn(*imsize)
        if isinstance(dtype, torch.dtype):
            as_numpy = False
            bd = _dev.to_dev(b0, dtype).contiguous()
        else:
            as_numpy = True
            bd = _dev.to_dev(b0 if dtype is None else b0.astype(dtype)).contiguous()
    else:
        as_numpy = _dev.is_numpy(b0)
        bd = _dev.to_dev(b0).contiguous().clone()
    code = _dev.code(bd.dtype)
    n = bd.numel()
    ws, out = _dev.scratch()

    def dot_into(u, v, slot):           # out[slot] = <u, v>, no host synchronisation
        _lib.check(lib.pfb_dot(code, _dev.ptr(u), _dev.ptr(v), n, _dev.ptr(out) + 8 * slot, _dev.ptr(ws),
                               _dev.stream()))

    def dot(u, v):
        dot_into(u, v, 0)
        return out[0].item()

    # the fused path: A as the library's own operator on device tensors
    H = None
    if not as_numpy:
        from .pcg import _as_hessian
        H = _as_hessian(A, bd)
        if H is not None and (H.plan.rdtype != bd.dtype or H.plan.embed is not None or
                              tuple(bd.shape[-2:]) != (H.nx, H.ny) or (bd.ndim == 3 and bd.shape[0] != H.nb)
                              or (bd.ndim == 2 and H.nb != 1)):
            H = None
    bout = torch.empty_like(bd) if H is not None else None

    def scale(v, s):
        _lib.check(lib.pfb_axpby(code, 0.0, _dev.ptr(v), float(s), _dev.ptr(v), n, _dev.stream()))

    scale(bd, 1.0 / math.sqrt(allsum([dot(bd, bd)])[0]))
    bp = bd.clone()
    b = bd
    beta, eps, k = 1.0, 1.0, 0
    while eps > tol and k < maxit:
        if H is not None:
            b3 = bp if bp.ndim == 3 else bp[None]
            with H.plan.lock:
                H.plan._enter_stream()
                _lib.check(lib.pfb_psfconv_apply_dots(
                    H.plan.handle, H.band0, H.nb, _dev.ptr(b3), _dev.ptr(H.beam),
                    H.wsum if H.wsum is not None else 0.0, H.sigmainv, _dev.ptr(bout), _dev.ptr(b3), None,
                    _dev.ptr(out), _dev.stream()))                 # out[0] = <bp, b>, out[2] = <b, b>
            b = bout
            dot_into(bp, bp, 1)
            pb, pp, bb = allsum(out[:3].tolist())
        else:
            res = A(bp.cpu().numpy()) if as_numpy else A(bp)
            b = _dev.to_dev(res, bp.dtype).contiguous()
            dot_into(bp, b, 0)
            dot_into(bp, bp, 1)
            dot_into(b, b, 2)
            pb, pp, bb = allsum(out[:3].tolist())                  # ONE host look per iteration
        bnorm = math.sqrt(bb)
        betap = beta
        beta = pb / pp
        bp.copy_(b)
        scale(bp, 1.0 / bnorm)                   # b /= bnorm ; bp[...] = b
        eps = abs(beta - betap) / betap
        k += 1
        if not k % report_freq and verbosity > 1:
            print(f"At iteration {k} eps = {eps:.3e}", file=sys.stderr)
    if verbosity:
        if k == maxit:
            print(f"Maximum iterations reached. eps = {eps:.3e}, beta = {beta:.3e}", file=sys.stderr)
        else:
            print(f"Success, converged after {k} iterations. beta = {beta:.3e}", file=sys.stderr)
    if as_numpy:
        return beta, bp.cpu().numpy()
    return beta, bp
