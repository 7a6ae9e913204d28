"""The PCG bookkeeping in the inverse row kernel's tail (PFB_PCG_TAIL, pcg_state.hpp) must be INVISIBLE: run this once
with PFB_PCG_TAIL=0 and once with 1 and compare the printed digests -- the iterates of every solve must be bitwise equal.
Many short solves at several sizes: the ticket / cross-XCD visibility of the partials is what is being stressed.

    python tools/check_tail.py [repeats]
"""
import hashlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.hessian import HessianPsf   # noqa: E402
from pfb_clean_amd.opt.pcg import pcg_fused              # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda')
for n, nb, dt in ((256, 2, torch.float64), (1024, 1, torch.float32), (512, 3, torch.float32), (2048, 2, torch.float32),
                  (4096, 1, torch.float32), (1024, 2, torch.float64)):
    cdt = torch.complex64 if dt == torch.float32 else torch.complex128
    g = torch.Generator(device=dev).manual_seed(n + nb)
    psfhat = (torch.rand((nb, 2 * n, n + 1), generator=g, device=dev, dtype=dt) / nb + 0.05).to(cdt)
    A = HessianPsf(psfhat, n, n, 2 * n, sigmainv=0.3)
    h = hashlib.sha256()
    for r in range(reps):
        b = torch.randn((nb, n, n), generator=g, device=dev, dtype=dt)
        x, _, res = pcg_fused(A, b, None, mdiv=0.3, tol=0.0, maxit=7, minit=7)
        h.update(x.cpu().numpy().tobytes())
        h.update(repr((res.iters, res.backtracks, res.eps, res.rnorm)).encode())
    print(n, nb, str(dt)[6:], h.hexdigest()[:24], flush=True)
    del A
