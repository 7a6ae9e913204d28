"""Two-level column kernel (k_col_pow2x, PFB_COL_X) against the one-level kernels on the same inputs, one process:
the switch is read per plan.  Also the PSFHAT producer into the class-major layout (from_psf + the layout handed back).

    python tools/check_colx.py [sizes ...]      e.g.  2048 4096 8192
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.psf import PsfConvPlan   # noqa: E402


def one(n, dt, nb=2):
    cdt = torch.complex64 if dt == torch.float32 else torch.complex128
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(7)
    psfhat = (torch.rand((nb, 2 * n, n + 1), generator=g, device=dev, dtype=dt) - 0.3).to(cdt)
    psfhat = psfhat + 1j * (torch.rand((nb, 2 * n, n + 1), generator=g, device=dev, dtype=dt) - 0.5).to(cdt)
    x = torch.randn((nb, n, n), generator=g, device=dev, dtype=dt)
    outs = {}
    for sw in ('0', '1'):
        os.environ['PFB_COL_X'] = sw
        plan = PsfConvPlan(psfhat, n, n, 2 * n)
        outs[sw] = plan.apply(x, sigmainv=0.1).clone()
        outs[sw + 'b'] = plan.apply(x[1:], sigmainv=0.0, band0=1).clone()
        plan.close()
    ref = outs['0'].abs().max().item()
    e1 = (outs['0'] - outs['1']).abs().max().item() / ref
    e2 = (outs['0b'] - outs['1b']).abs().max().item() / ref
    # producer: psf -> psfhat through the plan's own kernels, class-major layout, handed back in the reference layout
    psf = torch.randn((1, 2 * n, 2 * n), generator=g, device=dev, dtype=dt)
    os.environ['PFB_COL_X'] = '1'
    plan, ph = PsfConvPlan.from_psf(psf, n, n, want_psfhat=True)
    want = torch.fft.rfft2(torch.fft.ifftshift(psf.double(), dim=(1, 2)))
    e3 = (ph.to(torch.complex128) - want).abs().max().item() / want.abs().max().item()
    y1 = plan.apply(x[:1]).clone()
    plan.close()
    os.environ['PFB_COL_X'] = '0'
    plan0 = PsfConvPlan(ph, n, n, 2 * n)
    y0 = plan0.apply(x[:1]).clone()
    plan0.close()
    e4 = (y0 - y1).abs().max().item() / y0.abs().max().item()
    tol = 2e-5 if dt == torch.float32 else 1e-12
    ok = max(e1, e2, e3, e4) < tol
    print(f"{n}^2 {str(dt)[6:]}: apply {e1:.2e}  band-subrange {e2:.2e}  producer {e3:.2e}  producer->apply {e4:.2e}  "
          f"{'OK' if ok else 'FAIL'}", flush=True)
    return ok


if __name__ == '__main__':
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
    good = True
    for n in sizes:
        for dt in (torch.float32, torch.float64):
            good &= one(n, dt)
    sys.exit(0 if good else 1)
