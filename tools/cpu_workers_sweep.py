"""How does the CPU oracle's matvec scale with scipy.fft workers on the GPU box's host? (development aid
for choosing the cpu_baseline thread count of bench.py)"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import fftconv as ofc
n = 4096
rng = np.random.default_rng(0)
psfhat = (rng.random((2 * n, n + 1)) + 0j).astype(np.complex64)
x = rng.standard_normal((n, n)).astype(np.float32)
xpad, xhat, xout = ofc.make_scratch(psfhat, 2 * n, x.shape, x.dtype)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for w in (8, 16, 32, 64, 128, 256):
    ofc._hessian_psf_slice(xpad, xhat, xout, psfhat, None, 2 * n, x, nthreads=w, sigmainv=np.float32(0.1))
    t0 = time.perf_counter()
    for _ in range(3):
        ofc._hessian_psf_slice(xpad, xhat, xout, psfhat, None, 2 * n, x, nthreads=w, sigmainv=np.float32(0.1))
    print(f"workers {w:4d}: {(time.perf_counter() - t0) / 3 * 1e3:8.1f} ms per 4096^2 band matvec", flush=True)
