"""
Power method for the spectral norm of the (PSF-approximated) Hessian -- drop-in for
pfb/opt/power_method.py:11-49.  One matvec + two fused device reductions per
iteration; `A` may return an aliased buffer (it is normalised into `bp` straight away,
power_method.py:29-36).  With b0=None the start vector is drawn with numpy's global
RNG exactly like the reference (np.random.randn, fp64); `dtype` (numpy or torch dtype,
an extension) casts it and, when a torch dtype, keeps the iteration on device tensors.
A is called with the same array kind (numpy / GPU tensor) as the start vector.
"""
import math
import sys

import numpy as np
import torch

from .. import _lib, _dev


def power_method(A, imsize, b0=None, tol=1e-5, maxit=250, verbosity=1, report_freq=25,
                 dtype=None):
    lib = _lib.load()
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

    def dot(u, v):
        _lib.check(lib.pfb_dot(code, _dev.ptr(u), _dev.ptr(v), n, _dev.ptr(out), _dev.ptr(ws),
                               _dev.stream()))
        return out[0].item()

    def scale(v, s):
        _lib.check(lib.pfb_axpby(code, 0.0, _dev.ptr(v), float(s), _dev.ptr(v), n, _dev.stream()))

    scale(bd, 1.0 / math.sqrt(dot(bd, bd)))
    bp = bd.clone()
    b = bd
    beta, eps, k = 1.0, 1.0, 0
    while eps > tol and k < maxit:
        res = A(bp.cpu().numpy()) if as_numpy else A(bp)
        b = _dev.to_dev(res, bp.dtype).contiguous()
        bnorm = math.sqrt(dot(b, b))
        betap = beta
        beta = dot(bp, b) / dot(bp, bp)
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
