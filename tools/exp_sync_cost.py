import sys, time, torch
sys.path.insert(0, '.')
import bench as B
from pfb_clean_amd.operators.psf import PsfConvPlan
from pfb_clean_amd.operators.hessian import HessianPsf
from pfb_clean_amd.opt.pcg import pcg_fused
dev = torch.device('cuda')
for n, nb in ((4096, 8), (4096, 1), (1024, 1)):
    psfhat = torch.stack([B.synth_band(k, nb, n, n, torch.float32, dev) for k in range(nb)])
    plan = PsfConvPlan(psfhat, n, n, 2 * n)
    model = torch.stack([B.synth_model(k, n, n, torch.float32, dev) for k in range(nb)])
    b = plan.apply(model).clone()
    sig = 1e-3 * b.abs().max().item()
    A = HessianPsf(plan, n, n, 2 * n, sigmainv=sig)
    for minit in (40, 5):
        pcg_fused(A, b, None, mdiv=sig, tol=0.0, maxit=5, minit=5, backtrack=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, _, res = pcg_fused(A, b, None, mdiv=sig, tol=0.0, maxit=40, minit=minit, backtrack=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"n={n} nb={nb} minit={minit}: {dt/40*1e3:.4f} ms per iteration ({res.iters} its)")
