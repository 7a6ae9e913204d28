"""Experiment: can an HBM-bound streaming kernel (stand-in for the CG update of half the bands) run UNDER the
convolution of the other half when the persistent convolution kernels leave some CUs free (PFB_CU_LIMIT)?"""
import os, sys
import torch
sys.path.insert(0, '.')
from pfb_clean_amd.operators.psf import PsfConvPlan

n, nb = 4096, 4
dev = torch.device('cuda')
psfhat = torch.rand((nb, 2 * n, n + 1), dtype=torch.float32, device=dev).to(torch.complex64)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
x = torch.randn((nb, n, n), dtype=torch.float32, device=dev)
out = torch.empty_like(x)
a, b, c, d = (torch.randn((nb, n, n), dtype=torch.float32, device=dev) for _ in range(4))
o1, o2, o3 = (torch.empty_like(a) for _ in range(3))
s2 = torch.cuda.Stream()


def stream_work():          # ~7 passes over a 4-band vector, like k_pcg_update_dir
    torch.add(a, b, alpha=0.5, out=o1)
    torch.add(c, d, alpha=0.5, out=o2)
    torch.mul(a, 2.0, out=o3)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        stream_work()
    plan.apply(x, out=out)
    torch.cuda.current_stream().wait_stream(s2)


print("CU limit", os.environ.get('PFB_CU_LIMIT', 'none'))
print("conv(4 bands) alone      %.3f ms" % timeit(lambda: plan.apply(x, out=out)))
print("streaming alone          %.3f ms" % timeit(stream_work))
print("both, two streams        %.3f ms" % timeit(both))
