"""fp64 / large-size functional + timing probe (development aid)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.psf import PsfConvPlan
n = int(sys.argv[1]); nb = int(sys.argv[2]); dt = torch.float64 if sys.argv[3] == 'f64' else torch.float32
cdt = torch.complex128 if dt == torch.float64 else torch.complex64
dev = torch.device('cuda')
psfhat = torch.rand((nb, 2 * n, n + 1), dtype=dt, device=dev).to(cdt)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
x = torch.randn((nb, n, n), dtype=dt, device=dev)
out = torch.empty_like(x)
plan.set_profiling(True)
for _ in range(3):
    plan.apply(x, out=out, sigmainv=0.1)
torch.cuda.synchronize()
ms, k = plan.get_profile()
# reference via torch.fft on the same device (bring-up cross-check only)
xp = torch.zeros((nb, 2 * n, 2 * n), dtype=dt, device=dev); xp[:, :n, :n] = x
ref = torch.fft.irfft2(torch.fft.rfft2(xp) * psfhat, s=(2 * n, 2 * n))[:, :n, :n] + 0.1 * x
err = ((out - ref).abs().max() / ref.abs().max()).item()
print(f"n={n} nb={nb} {dt} fast={plan.fast_path}: stage ms {[round(m / k, 3) for m in ms]} relerr vs torch.fft {err:.2e}")
