"""How much EXTRA HBM traffic does the forward row kernel absorb?  (diagnostic build)

Folding the CG update into the forward row kernel (read x, r, p, Ap / write x', r', p' per tile instead of reading p)
only pays if the kernel can stream that traffic in the shadow of its own work.  The diagnostic build lets
k_row_fwd_pow2q stream ONE extra array of the cube's shape (N s bytes per band: 1/7 of what the fused update moves)
through registers, spread over the even-bin sweep, result discarded.  Compared: the kernel's time without / with the
extra stream, and the time a pure streaming kernel needs for the same bytes.

    python tools/exp_fwd_extra_stream.py [size] [bands]
"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['PFB_HIP_LIB'] = os.path.join(ROOT, 'pfb_clean_amd', 'libpfb_hip_stamp.so')
import torch                                             # noqa: E402
from pfb_clean_amd import _lib, _dev                     # noqa: E402
from pfb_clean_amd.operators.psf import PsfConvPlan      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda')
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.pfb_debug_set_extra_stream.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device=dev).manual_seed(1)
psfhat = (torch.rand((nb, 2 * n, n + 1), generator=g, device=dev) / nb).to(torch.complex64)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
x = torch.randn((nb, n, n), generator=g, device=dev)
r = torch.randn((nb, n, n), generator=g, device=dev)
extra = torch.randn((nb, n, n), generator=g, device=dev)
sink = torch.zeros(2, dtype=torch.float64, device=dev)
out = torch.empty_like(x)
dots = torch.zeros(3, dtype=torch.float64, device=dev)


def stage_ms(reps=30):
    for _ in range(3):
        _lib.check(lib.pfb_psfconv_apply_dots(plan._h, 0, nb, _dev.ptr(x), None, 0.0, 0.1, _dev.ptr(out), _dev.ptr(x),
                                              _dev.ptr(r), _dev.ptr(dots), _dev.stream()))
    torch.cuda.synchronize()
    plan.set_profiling(1)
    for _ in range(reps):
        _lib.check(lib.pfb_psfconv_apply_dots(plan._h, 0, nb, _dev.ptr(x), None, 0.0, 0.1, _dev.ptr(out), _dev.ptr(x),
                                              _dev.ptr(r), _dev.ptr(dots), _dev.stream()))
    torch.cuda.synchronize()
    ms, k = plan.get_profile()
    plan.set_profiling(0)
    return [m / k for m in ms]


res = {0: [], 1: []}
for rnd in range(3):
    for on in (0, 1):
        rc = raw.pfb_debug_set_extra_stream(C.c_void_p(extra.data_ptr()) if on else None, C.c_void_p(sink.data_ptr()))
        assert rc == 0, (rc, lib.pfb_last_error())
        res[on].append(stage_ms()[0])
raw.pfb_debug_set_extra_stream(None, None)
# a pure streaming read of the same bytes (pfb_dot of the array with itself reads it once: both operands are the same lines)
ws, o = _dev.scratch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    lib.pfb_dot(0, _dev.ptr(extra), _dev.ptr(extra), extra.numel(), _dev.ptr(o), _dev.ptr(ws), _dev.stream())
e0.record()
for _ in range(20):
    lib.pfb_dot(0, _dev.ptr(extra), _dev.ptr(extra), extra.numel(), _dev.ptr(o), _dev.ptr(ws), _dev.stream())
e1.record()
torch.cuda.synchronize()
t_stream = e0.elapsed_time(e1) / 20
base, ext = statistics.median(res[0]), statistics.median(res[1])
gb = extra.numel() * 4 / 1e9
print(f"# extra-stream experiment, {n}^2 x {nb} fp32 (diagnostic build): row_fwd {base:.4f} ms -> {ext:.4f} ms with "
      f"{gb:.3f} GB streamed in addition (+{ext - base:.4f} ms); a pure streaming read of those bytes takes {t_stream:.4f} ms "
      f"({gb / t_stream * 1e3:.0f} GB/s); absorbed for free: {max(0.0, 1 - (ext - base) / t_stream) * 100:.0f} %")
