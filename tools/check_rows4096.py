"""Persistent row kernels for 4096-point rows (ny = 8192; fp32: 4-row tiles, fp64: 2-row 512-thread tiles; small twiddle
tables) against the plain
kernels (PFB_FWD_PERSIST=0 / PFB_INV_PERSIST=0) on the same inputs, every epilogue mode.  The switches are read per
plan, so one process compares both.
    python tools/check_rows4096.py
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd import _lib, _dev                      # noqa: E402
from pfb_clean_amd.operators.psf import PsfConvPlan      # noqa: E402

lib = _lib.load()
dev = torch.device('cuda')
ok = True
for nx, ny, nb, dt in ((256, 8192, 2, torch.float32), (1024, 8192, 1, torch.float32), (8192, 8192, 1, torch.float32),
                       (256, 8192, 2, torch.float64), (2048, 8192, 1, torch.float64)):
    cdt = torch.complex64 if dt == torch.float32 else torch.complex128
    g = torch.Generator(device=dev).manual_seed(nx + ny)
    psfhat = ((torch.rand((nb, 2 * nx, ny + 1), generator=g, device=dev, dtype=dt) - 0.3)
              + 1j * (torch.rand((nb, 2 * nx, ny + 1), generator=g, device=dev, dtype=dt) - 0.5)).to(cdt) / (nx * ny) ** 0.5
    x = torch.randn((nb, nx, ny), generator=g, device=dev, dtype=dt)
    r = torch.randn((nb, nx, ny), generator=g, device=dev, dtype=dt)
    beam = torch.rand((nb, nx, ny), generator=g, device=dev, dtype=dt)
    res = {}
    for tag, fp, ip in (('plain', '0', '0'), ('persistent', '1', '1')):
        os.environ['PFB_FWD_PERSIST'], os.environ['PFB_INV_PERSIST'] = fp, ip
        plan = PsfConvPlan(psfhat, nx, ny, 2 * ny)
        outs = []
        for bm in (None, beam):
            for mode in (0, 1, 2):
                out = torch.empty_like(x)
                dots = torch.zeros(3, dtype=torch.float64, device=dev)
                if mode == 0:
                    _lib.check(lib.pfb_psfconv_apply(plan._h, 0, nb, _dev.ptr(x), _dev.ptr(bm), 0.0, 0.1, _dev.ptr(out),
                                                     None, None, _dev.stream()))
                else:
                    _lib.check(lib.pfb_psfconv_apply_dots(plan._h, 0, nb, _dev.ptr(x), _dev.ptr(bm), 0.0, 0.1, _dev.ptr(out),
                                                          _dev.ptr(x), _dev.ptr(r) if mode == 2 else None, _dev.ptr(dots),
                                                          _dev.stream()))
                outs.append((out.clone(), dots.clone()))
        res[tag] = outs
        plan.close()
    worst = 0.0
    for (o0, d0), (o1, d1) in zip(res['plain'], res['persistent']):
        e = (o0 - o1).abs().max().item() / o0.abs().max().item()
        ed = ((d0 - d1).abs() / (d0.abs() + 1e-30)).max().item()
        worst = max(worst, e, ed if ed < 1 else 0.0 if d0.abs().max().item() == 0 else ed)
    good = worst < (2e-6 if dt == torch.float32 else 1e-13)
    ok &= good
    print(f"{nx} x {ny} x {nb} {str(dt)[6:]}: max rel difference over 6 modes (conv, fused dots) {worst:.2e} {'OK' if good else 'FAIL'}",
          flush=True)
os.environ.pop('PFB_FWD_PERSIST', None)
os.environ.pop('PFB_INV_PERSIST', None)
sys.exit(0 if ok else 1)
