"""Experiment: 8-band launches vs band-serial launches (is T kept in the 256 MB Infinity Cache?)."""
import sys
import torch
sys.path.insert(0, '.')
from pfb_clean_amd.operators.psf import PsfConvPlan

def timeit(f, reps=10):
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = 8
dev = torch.device('cuda')
P = Q = 2 * n
psfhat = torch.rand((nb, P, Q // 2 + 1), dtype=torch.float32, device=dev).to(torch.complex64)
x = torch.randn((nb, n, n), dtype=torch.float32, device=dev)
out = torch.empty_like(x)
full = PsfConvPlan(psfhat, n, n, Q)
print("8-band launches: %.3f ms" % timeit(lambda: full.apply(x, out=out)), flush=True)
for grp in (1, 2, 4):
    plans = [PsfConvPlan(psfhat[b:b + grp], n, n, Q) for b in range(0, nb, grp)]
    def f():
        for i, p in enumerate(plans):
            p.apply(x[i * grp:(i + 1) * grp], out=out[i * grp:(i + 1) * grp])
    print("groups of %d band(s): %.3f ms" % (grp, timeit(f)), flush=True)
    del plans
