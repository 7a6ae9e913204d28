"""Per-phase time table of the three persistent convolution kernels from the DIAGNOSTIC build
(make -C pfb_clean_amd/csrc stamp -> libpfb_hip_stamp.so: thread 0 of every workgroup stamps the 100 MHz
wall clock at its phase boundaries for loop trips 2..5).  The stamps perturb the schedule a little (each is
an s_memrealtime + a store); totals are reported next to the un-stamped kernel time for that reason.

    python tools/stamp_report.py [size] [bands] [f32|f64]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['PFB_HIP_LIB'] = os.path.join(ROOT, 'pfb_clean_amd', 'libpfb_hip_stamp.so')
import numpy as np          # noqa: E402
import torch                # noqa: E402
from pfb_clean_amd import _lib, _dev                     # noqa: E402
from pfb_clean_amd.operators.psf import PsfConvPlan      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else torch.float32
cdt = torch.complex128 if dt == torch.float64 else torch.complex64
ITS, NWG, NSLOT = 4, 1024, 16
NAMES = {
    # default forward kernel k_row_fwd_pow2q (parities in sequence); PFB_FWD_SEQ=0 runs k_row_fwd_pow2p, whose seven
    # intervals are: pack + w_M multiply | both transforms | LDS write even + issue next x | even sweep | LDS write odd |
    # odd sweep | final barrier
    0: ('k_row_fwd_pow2q', ['top -> pack (x * beam)', 'FFT (even bins)', 'LDS write even + barriers',
                             'post-process + store even bins', 'z w_M, FFT (odd bins) with the next rows requested per pass',
                             'LDS write odd + post-process + store odd bins', 'final barrier']),
    1: ('k_col_pow2p', ['aw = a w, issue psf_e + next a', 'FFT (even)', 'wait psf_e, multiply', 'issue psf_o, IFFT (even)',
                        'FFT (odd)', 'wait psf_o, multiply', 'IFFT (odd)', 'combine + stores issued']),
    2: ('k_row_inv_pow2p', ['barrier (top)', 'wait y_even, scatter to LDS', 'issue y_odd + barrier', 'build (even)',
                            'IFFT (even)', 'park + barrier', 'wait y_odd, scatter', 'issue x, r, next y_even + barrier',
                            'build (odd)', 'IFFT (odd)', 'wait x, r; epilogue + stores issued']),
}

if n >= 8192 or os.environ.get('PFB_COL_X', '0') not in ('', '0'):
    # two-level column kernel (k_col_pow2x): four rounds (classes 0, 2, 1, 3) of FFT | multiply + IFFT | accumulate + next input
    NAMES[1] = ('k_col_pow2x', ['r0: a_lo + a_hi, FFT', 'r0: wait psf, multiply, IFFT (psf class 2 requested)',
                                'r0 -> r2: keep c0, form (a_lo - a_hi) w^2n', 'r2: FFT', 'r2: multiply, IFFT (class 1 requested)',
                                'r2 -> r1: accumulate, park hi, form both odd-class inputs', 'r1: FFT (next a_hi requested)',
                                'r1: multiply, IFFT (class 3 requested)', 'r1 -> r3: accumulate (park read-modify-write)',
                                'r3: FFT (next a_lo requested)', 'r3: multiply, IFFT (next class 0 requested)',
                                'combine + stores issued'])
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.pfb_debug_set_stamps.argtypes = [C.c_void_p]
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
psfhat = (torch.rand((nb, 2 * n, n + 1), generator=g, device=dev, dtype=dt) / nb).to(cdt)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
x = torch.randn((nb, n, n), generator=g, device=dev, dtype=dt)
r = torch.randn((nb, n, n), generator=g, device=dev, dtype=dt)
out = torch.empty_like(x)
dots = torch.zeros(3, dtype=torch.float64, device=dev)


def apply():
    _lib.check(lib.pfb_psfconv_apply_dots(plan._h, 0, nb, _dev.ptr(x), None, 0.0, 0.1, _dev.ptr(out),
                                          _dev.ptr(x), _dev.ptr(r), _dev.ptr(dots), _dev.stream()))


for _ in range(3):
    apply()
buf = torch.zeros(3 * NWG * ITS * NSLOT, dtype=torch.int64, device=dev)
assert raw.pfb_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
plan.set_profiling(1)
apply()
torch.cuda.synchronize()
ms, napply = plan.get_profile()
raw.pfb_debug_set_stamps(None)
st = buf.cpu().numpy().reshape(3, NWG, ITS, NSLOT).astype(np.float64) * 0.01     # 100 MHz ticks -> us
print(f"# stamp report: {n}^2 x {nb} bands {dt}; kernel times of the stamped launch (HIP events): "
      f"row_fwd {ms[0]:.3f} ms, col {ms[1]:.3f} ms, row_inv {ms[2]:.3f} ms")
for kid, (name, phases) in NAMES.items():
    a = st[kid]
    live = (a[:, :, 0] > 0) & (a[:, :, len(phases)] > 0)
    if not live.any():
        print(f"\n## {name}: no stamps (kernel not used at this size)")
        continue
    print(f"\n## {name}: {int(live.any(axis=1).sum())} workgroups, {int(live.sum())} stamped trips; microseconds per trip")
    print("| phase | mean | p10 | p90 |")
    print("|---|---|---|---|")
    tot = np.zeros(live.sum())
    for k, ph in enumerate(phases):
        d = (a[:, :, k + 1] - a[:, :, k])[live]
        tot += d
        print(f"| {ph} | {d.mean():.2f} | {np.percentile(d, 10):.2f} | {np.percentile(d, 90):.2f} |")
    print(f"| **sum of phases** | {tot.mean():.2f} | {np.percentile(tot, 10):.2f} | {np.percentile(tot, 90):.2f} |")
    # trip to trip (includes whatever sits between the last stamp and the next top)
    nxt = (a[:, 1:, 0] - a[:, :-1, 0])[live[:, 1:] & live[:, :-1]]
    if nxt.size:
        print(f"| trip to trip | {nxt.mean():.2f} | {np.percentile(nxt, 10):.2f} | {np.percentile(nxt, 90):.2f} |")

# per-XCD spread of the trip-to-trip time (workgroup b runs on XCD b % 8 in launch order): a systematic difference between
# XCDs is what a dynamic tile queue could balance
print("## trip-to-trip time by XCD (mean microseconds over the stamped trips of the workgroups b = xcd mod 8)")
print("| kernel | " + " | ".join(f"xcd {x}" for x in range(8)) + " | slowest workgroup / mean |")
print("|---|" + "---|" * 9)
for kid in (0, 1, 2):
    name, _ = NAMES[kid]
    a = st[kid]
    ok = (a[:, :, 0] > 0).all(axis=1)
    idx = np.nonzero(ok)[0]
    if idx.size == 0:
        continue
    t2t = (a[:, 1:, 0] - a[:, :-1, 0]).mean(axis=1)                      # per workgroup
    cells = []
    for x in range(8):
        sel = idx[idx % 8 == x]
        cells.append(f"{t2t[sel].mean():.2f}" if len(sel) else "-")
    print(f"| {name} | " + " | ".join(cells) + f" | {t2t[idx].max() / t2t[idx].mean():.3f} |")
