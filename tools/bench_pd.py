"""Timing of the BASELINE config #4 pieces: 2048^2 x 4 bands, bases db1..db4 (+self), 3 levels
(development aid; prints per-call milliseconds of psi.dot, psi.hdot, dual_update, conv and one
primal-dual iteration)."""
import sys, time
from functools import partial
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.psf import PsfConvPlan
from pfb_clean_amd.operators.psi import Psi
from pfb_clean_amd.prox.prox_21m import dual_update_numba
from pfb_clean_amd.opt.primal_dual import primal_dual_optimised

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else torch.float32
cdt = torch.complex128 if dt == torch.float64 else torch.complex64
bases = ['self', 'db1', 'db2', 'db3', 'db4']
dev = torch.device('cuda')
psi = Psi(nb, n, n, bases, 3, 1, dtype=dt)
x = torch.randn((nb, n, n), dtype=dt, device=dev)
a = torch.zeros((nb, len(bases), psi.Nymax, psi.Nxmax), dtype=dt, device=dev)
vp = torch.randn_like(a)
w = torch.rand_like(a[0])
psfhat = (torch.rand((nb, 2 * n, n + 1), dtype=dt, device=dev) / nb).to(cdt)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
out = torch.empty_like(x)

def timeit(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

s = 4 if dt == torch.float32 else 8
coef_bytes = a.numel() * s
img_bytes = x.numel() * s
t = timeit(lambda: psi.dot(x, a));  print(f"psi.dot   {t:8.3f} ms   ({(img_bytes*len(bases) + coef_bytes)/t/1e6:7.1f} GB/s alg)")
t = timeit(lambda: psi.hdot(a, out)); print(f"psi.hdot  {t:8.3f} ms   ({(img_bytes + coef_bytes)/t/1e6:7.1f} GB/s alg)")
t = timeit(lambda: dual_update_numba(vp, a, 1e-3, sigma=0.5, weight=w, vp_out=vp)); print(f"dual_upd  {t:8.3f} ms   ({(4*coef_bytes + w.numel()*s)/t/1e6:7.1f} GB/s alg)")
t = timeit(lambda: plan.apply(x, out=out)); print(f"conv      {t:8.3f} ms")
data = plan.apply(x).clone()
conv = lambda v: plan.apply(v, out=out)
grad = lambda v: conv(v) - data
xx = torch.zeros_like(x); vv = torch.zeros_like(a)
primal_dual_optimised(xx, vv, 1e-3, psi.hdot, psi.dot, 1.0, None, w, None, grad, nu=len(bases), tol=0.0, maxit=2, positivity=1, verbosity=0)   # warm-up (first torch op of a kind costs ~15 ms)
xx.zero_(); vv.zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
primal_dual_optimised(xx, vv, 1e-3, psi.hdot, psi.dot, 1.0, None, w, None, grad, nu=len(bases), tol=0.0, maxit=20, positivity=1, verbosity=0)
torch.cuda.synchronize(); print(f"PD iteration (20 its) {(time.perf_counter()-t0)/20*1e3:8.3f} ms each   shape {tuple(a.shape)} {dt}")
