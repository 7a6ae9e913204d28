"""Randomised size sweep: convolution + Hessian (beam, wsum, sigmainv) of random (nx, ny | nx_psf, ny_psf)
problems against the CPU oracle, through whatever path the plan picks (fast / embedded / coverage).
Development aid: python tools/stress_sizes.py [ncases] [seed]"""
import sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fftconv as ofc
from pfb_clean_amd.operators import psf as P_, hessian as H_


def smooth(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n == 1


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
done = 0
while done < ncases:
    nx, ny = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    P = int(rng.integers(nx, 2 * nx + 4))
    Q = int(rng.integers(ny, 2 * ny + 4))
    Q += Q % 2
    if not (smooth(P) and smooth(Q // 2)):
        continue
    nb = int(rng.integers(1, 4))
    rdt = np.float64 if rng.random() < 0.5 else np.float32
    cdt = np.complex128 if rdt == np.float64 else np.complex64
    psfhat = ofc.psfhat_from_psf(rng.standard_normal((nb, P, Q)))
    x = rng.standard_normal((nb, nx, ny))
    beam = 0.5 + rng.random((nb, nx, ny))
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, x.shape, np.float64)
    ref_c = ofc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, x).copy()
    ref_h = ofc.hessian_psf_cube(xpad, xhat, xout, beam, psfhat, Q, x, sigmainv=0.7, wsum=2.5)
    P_.clear_plan_cache()
    plan = P_.plan_for(psfhat.astype(cdt), nx, ny, Q)
    kind = 'embed' if plan.embed else ('fast' if plan.fast_path else 'generic')
    got_c = P_.psf_convolve_cube(None, None, None, psfhat.astype(cdt), Q, x.astype(rdt))
    got_h = H_.hessian_psf_cube(None, None, None, beam.astype(rdt), psfhat.astype(cdt), Q, x.astype(rdt),
                                sigmainv=0.7, wsum=2.5)
    e = max(np.abs(got_c - ref_c).max() / np.abs(ref_c).max(), np.abs(got_h - ref_h).max() / np.abs(ref_h).max())
    key = (kind, rdt.__name__)
    worst[key] = max(worst.get(key, 0.0), float(e))
    tol = 1e-12 if rdt == np.float64 else 2e-5
    flag = '' if e < tol else '   <<<<<< FAIL'
    print(f"{done:3d} ({nx:3d},{ny:3d}|{P:3d},{Q:3d}) nb={nb} {rdt.__name__:8s} {kind:8s} relerr {e:.2e}{flag}", flush=True)
    done += 1
print("worst per (path, dtype):", worst)
