"""Run a few PSF-convolution applies (development aid for rocprofv3 --pmc passes)."""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.psf import PsfConvPlan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device('cuda')
psfhat = torch.rand((nb, 2 * n, n + 1), dtype=torch.float32, device=dev).to(torch.complex64)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
x = torch.randn((nb, n, n), dtype=torch.float32, device=dev)
out = torch.empty_like(x)
dot = torch.zeros(1, dtype=torch.float64, device=dev)
plan.set_profiling(True)
for _ in range(reps):
    plan.apply(x, out=out, sigmainv=0.1, dot_with=x, dot_out=dot)
torch.cuda.synchronize()
ms, k = plan.get_profile()
print("stage ms per apply:", [round(m / k, 4) for m in ms], "n", n, "nb", nb)
