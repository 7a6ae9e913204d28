"""Ad-hoc timing of the convolution stages (development aid, not the judged bench)."""
import sys, time
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.psf import PsfConvPlan

def run(n, dtype, nb=1, reps=10):
    dev = torch.device('cuda')
    cdt = torch.complex64 if dtype == torch.float32 else torch.complex128
    P = Q = 2 * n
    psfhat = torch.rand((nb, P, Q // 2 + 1), dtype=dtype, device=dev).to(cdt)
    plan = PsfConvPlan(psfhat, n, n, Q)
    x = torch.randn((nb, n, n), dtype=dtype, device=dev)
    out = torch.empty_like(x)
    for _ in range(3):
        plan.apply(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.apply(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    s = 4 if dtype == torch.float32 else 8
    balg = nb * s * (2 * n * n + 2 * P * (Q // 2 + 1))
    print(f"n={n} {dtype} nb={nb} fast={plan.fast_path} vb={plan.vb}: {ms:.3f} ms/apply, "
          f"{balg / ms / 1e6:.1f} GB/s algorithmic ({balg/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)

if __name__ == '__main__':
    for n in (256, 1024, 2048, 4096):
        run(n, torch.float32)
    run(1024, torch.float64)
    run(2048, torch.float64)
    run(1024, torch.float32, nb=8)
