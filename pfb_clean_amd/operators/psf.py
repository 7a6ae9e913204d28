"""
PSF convolution on MI355X -- drop-in for pfb/operators/psf.py:11-56.

    psf_convolve_slice(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1)
    psf_convolve_cube (xpad, xhat, xout, psfhat, lastsize, x, nthreads=1)

Same positional order and aliasing contract as the reference: the result is written
into and returned as `xout` (psf.py:29,56) and `x` is not overwritten.  `xpad`/`xhat`
are the reference's host scratch buffers; the HIP path never materialises the padded
image or its spectrum, so they are accepted and ignored (None is fine).  numpy
arguments are staged through the GPU (PCIe both ways per call); torch-ROCm tensors
stay resident -- that is the fast way to drive it.

The psfhat re-layout happens once per distinct psfhat (plan cache keyed on the
buffer identity + a strided fingerprint), mirroring how the workers bind psfhat once
with functools.partial (workers/spotless.py:175-183).
"""
import ctypes as C
import threading
from collections import OrderedDict

import numpy as np
import torch

from .. import _lib, _dev


def _pow2ceil(n):
    p = 1
    while p < n:
        p *= 2
    return p


# what the kernels take (csrc): the fast path serves power-of-two images 64 <= nx <= 8192, 128 <= ny <= NY_FAST_MAX
# with nx_psf = 2 nx, ny_psf = 2 ny; the coverage ("generic") kernels hold one line in two LDS buffers
NX_FAST_MAX = 8192
NY_FAST_MAX = {torch.float32: 16384, torch.float64: 8192}
LINE_MAX = {torch.float32: 10240, torch.float64: 5120}


def _embed_grid(nx, ny, nx_psf, ny_psf, rdtype):
    """(nx2, ny2) of the power-of-two plan an (nx, ny | nx_psf, ny_psf) problem is embedded in, or None: already
    a fast-path size, switched off (PFB_NO_EMBED / PFB_FORCE_GENERIC), odd ny_psf, or beyond the fast kernels.
    Problems the coverage kernels can run (every line fits the LDS) are only embedded when the padded problem has
    at most 3x the pixels; problems they cannot run (nx_psf > 10240 fp32 / 5120 fp64: 6000^2, 7200^2 ... images
    with psf-oversize 2) are embedded whenever the fast path can hold them."""
    import os
    if os.environ.get('PFB_NO_EMBED', '0') not in ('', '0') or os.environ.get('PFB_FORCE_GENERIC', '0') not in ('', '0'):
        return None
    if ny_psf % 2:
        return None
    nx2, ny2 = max(64, _pow2ceil(nx)), max(128, _pow2ceil(ny))
    if (nx2, ny2) == (nx, ny) and (nx_psf, ny_psf) == (2 * nx, 2 * ny):
        return None
    if nx2 > NX_FAST_MAX or ny2 > NY_FAST_MAX[rdtype]:
        return None
    generic_ok = max(nx_psf, ny_psf // 2) <= LINE_MAX[rdtype]
    if generic_ok and nx2 * ny2 > 3 * nx * ny:
        return None
    return nx2, ny2


class PsfConvPlan:
    """Owns the device plan (twiddles, re-laid-out psfhat, spectrum workspace) for one
    psfhat cube.  psfhat: (nband, nx_psf, nyo2) or (nx_psf, nyo2) complex.

    A plan is single-owner in the C-ABI (one stream and one host thread at a time: its spectrum
    workspace, fused-dot partials and profiling slots are per plan).  plan_for() hands the SAME plan to
    every host thread that presents the same psfhat (the reference's dask threads do, pcg.py:346-356), so
    the Python layer serialises: `lock` is held for the enqueue of an apply and for a whole fused solve,
    and a caller on a different stream than the previous user first waits for that stream's work."""

    def __init__(self, psfhat, nx, ny, lastsize):
        lib = _lib.load()
        ph = _dev.to_dev(psfhat)
        if ph.dtype not in _dev.REAL_OF:
            raise TypeError(f"psfhat must be complex64/complex128, got {ph.dtype}")
        if ph.ndim == 2:
            ph = ph[None]
        if ph.ndim != 3:
            raise ValueError("psfhat must be (nx_psf, nyo2) or (nband, nx_psf, nyo2)")
        self.nband, self.nx_psf, self.nyo2 = ph.shape
        self.nx, self.ny, self.lastsize = int(nx), int(ny), int(lastsize)
        if self.nyo2 != self.lastsize // 2 + 1:
            raise ValueError(f"psfhat last axis {self.nyo2} != lastsize//2+1 "
                             f"({self.lastsize // 2 + 1})")
        self.rdtype = _dev.REAL_OF[ph.dtype]
        self.code = _dev.code(self.rdtype)
        self.device = ph.device
        self._lib = lib
        self._h = C.c_void_p()
        self.lock = threading.RLock()
        # Arbitrary sizes on the power-of-two kernels: the same image-space PSF is re-gridded
        # (pfb_psfhat_regrid) onto nx_psf2 = 2 nx2, ny_psf2 = 2 ny2 with nx2, ny2 the next powers
        # of two, images are zero-padded into (nx2, ny2) buffers and results cropped.  Identical
        # results to rounding, 3-6x faster than the line-per-workgroup coverage kernels
        # (measured: 3600^2 x 2 bands 3.2 ms generic vs 0.55 ms for two 4096^2 bands).
        self.embed = None
        grid2 = _embed_grid(self.nx, self.ny, self.nx_psf, self.lastsize, self.rdtype)
        cnx, cny, cpx, cpy = self.nx, self.ny, self.nx_psf, self.lastsize
        if grid2 is not None:
            ph2 = self._regrid(ph.contiguous(), grid2)
            if ph2 is not None:
                self.embed = grid2
                ph = ph2
                cnx, cny, cpx, cpy = grid2[0], grid2[1], 2 * grid2[0], 2 * grid2[1]
        h = C.c_void_p()
        _lib.check(lib.pfb_psfconv_plan_create(cnx, cny, cpx, cpy, self.nband, self.code, C.byref(h)))
        self._h = h
        _lib.check(lib.pfb_psfconv_set_psfhat(h, _dev.ptr(ph.contiguous()), _dev.stream()))
        torch.cuda.current_stream().synchronize()      # ph may be a temporary
        fast, vb, wsb = C.c_int(), C.c_int(), C.c_size_t()
        lib.pfb_psfconv_plan_info(h, C.byref(fast), C.byref(vb), C.byref(wsb))
        self.fast_path, self.vb, self.workspace_bytes = bool(fast.value), vb.value, wsb.value

    def _regrid(self, ph, grid2):
        """psfhat on the caller's grid -> psfhat on the (2 nx2, 2 ny2) grid, or None when the
        library cannot do it (prime factor > 13).  Done in fp64 (memory permitting: the detour holds
        the spectrum, the PSF and the embedded PSF of all bands at once), so that fp32 plans see no extra
        rounding from it."""
        nx2, ny2 = grid2
        cdt = ph.dtype
        # fp64 working set of pfb_psfhat_regrid: spec + psf + psf2 + psfhat2 (+ transform scratch)
        need64 = 16 * self.nband * (self.nx_psf * self.nyo2 * 2 + 2 * nx2 * (ny2 + 1) * 2) + 8 * self.nband * (
            self.nx_psf * self.lastsize + 4 * nx2 * ny2) * 2
        try:
            free = torch.cuda.mem_get_info(ph.device)[0]
        except Exception:
            free = 0
        works = ((torch.complex128, 1), (cdt, self.code)) if need64 < 0.6 * free else ((cdt, self.code),)
        for work in works:
            src = ph.to(work[0]) if ph.dtype != work[0] else ph
            dst = torch.empty((self.nband, 2 * nx2, ny2 + 1), dtype=work[0], device=ph.device)
            rc = self._lib.pfb_psfhat_regrid(work[1], _dev.ptr(src), self.nband, self.nx, self.ny,
                                             self.nx_psf, self.lastsize, 2 * nx2, 2 * ny2, _dev.ptr(dst),
                                             _dev.stream())
            if rc == _lib.PFB_OK:
                return dst.to(cdt)
            if rc != _lib.PFB_ERR_UNSUPPORTED:
                _lib.check(rc)
            if work[0] == cdt:
                break
        return None

    def _pad(self, t, nb):
        """(nb, nx, ny) -> zero-padded (nb, nx2, ny2) (embedded plans only)."""
        z = torch.zeros((nb,) + self.embed, dtype=self.rdtype, device=t.device)
        z[:, :self.nx, :self.ny] = t
        return z

    @property
    def handle(self):
        return self._h

    @classmethod
    def from_psf(cls, psf, nx, ny, want_psfhat=False):
        """Build the plan straight from the real PSF cube (nband, nx_psf, ny_psf) | (nx_psf, ny_psf):
        psfhat = r2c(ifftshift(psf)) is produced by the library's own kernels
        (pfb_psfconv_set_psf; gridder.py:712-714) and never leaves the device.  With
        want_psfhat=True also returns it in the reference's layout."""
        lib = _lib.load()
        p = _dev.to_dev(psf)
        if p.ndim == 2:
            p = p[None]
        if p.ndim != 3 or p.dtype not in (torch.float32, torch.float64):
            raise ValueError("psf must be a real (nband, nx_psf, ny_psf) or (nx_psf, ny_psf) array")
        p = p.contiguous()
        if _embed_grid(int(nx), int(ny), int(p.shape[1]), int(p.shape[2]), p.dtype) is not None:
            # not a fast-path size: transform on the PSF's own grid first, then let the constructor
            # re-grid it onto the power-of-two plan (operators/fft.py picks the native producer)
            from .fft import psfhat_from_psf
            ph = psfhat_from_psf(p)
            plan = cls(ph, nx, ny, int(p.shape[2]))
            return (plan, ph) if want_psfhat else plan
        self = cls.__new__(cls)
        self.embed = None
        self.lock = threading.RLock()
        self.nband, self.nx_psf, self.lastsize = (int(v) for v in p.shape)
        self.nyo2 = self.lastsize // 2 + 1
        self.nx, self.ny = int(nx), int(ny)
        self.rdtype = p.dtype
        self.code = _dev.code(self.rdtype)
        self.device = p.device
        h = C.c_void_p()
        _lib.check(lib.pfb_psfconv_plan_create(self.nx, self.ny, self.nx_psf, self.lastsize,
                                               self.nband, self.code, C.byref(h)))
        self._h = h
        self._lib = lib
        cdt = torch.complex64 if self.rdtype == torch.float32 else torch.complex128
        ph = torch.empty((self.nband, self.nx_psf, self.nyo2), dtype=cdt, device=p.device) if want_psfhat else None
        try:
            _lib.check(lib.pfb_psfconv_set_psf(h, _dev.ptr(p), _dev.ptr(ph), _dev.stream()))
        except Exception:
            self.close()
            raise
        torch.cuda.current_stream().synchronize()      # p may be a temporary
        fast, vb, wsb = C.c_int(), C.c_int(), C.c_size_t()
        lib.pfb_psfconv_plan_info(h, C.byref(fast), C.byref(vb), C.byref(wsb))
        self.fast_path, self.vb, self.workspace_bytes = bool(fast.value), vb.value, wsb.value
        return (self, ph) if want_psfhat else self

    def apply(self, x, out=None, beam=None, wsum=None, sigmainv=0.0, band0=0,
              dot_with=None, dot_out=None):
        """out = [beam*]conv([beam*]x)[/wsum] + sigmainv*x on bands
        [band0, band0+nb) where nb = x.shape[0] (x is (nb, nx, ny) or (nx, ny))."""
        squeeze = x.ndim == 2
        x3 = x[None] if squeeze else x
        if x3.dtype != self.rdtype or not x3.is_cuda:
            raise TypeError(f"x must be a {self.rdtype} GPU tensor, got {x3.dtype}")
        nb = x3.shape[0]
        if tuple(x3.shape[1:]) != (self.nx, self.ny):
            raise ValueError(f"x has shape {tuple(x.shape)}, plan is for ({self.nx},{self.ny})")
        x3 = x3.contiguous()
        if out is None:
            out3 = torch.empty_like(x3)
        else:
            out3 = out[None] if out.ndim == 2 else out
            if not out3.is_contiguous() or out3.dtype != self.rdtype or out3.shape != x3.shape:
                raise ValueError("out must be a contiguous tensor shaped like x")
        if beam is not None:
            beam = beam[None] if beam.ndim == 2 else beam
            if beam.shape != x3.shape:
                raise ValueError('Beam has incorrect shape')
            beam = beam.contiguous()
        if dot_with is not None:
            dot_with = dot_with.contiguous()
        if self.embed is not None:
            # padded input / output buffers are kept per (band count, host thread): the margins of the input
            # buffer are written once (zeros) and never touched again, so a call costs one copy in and one out
            cache = self.__dict__.setdefault('_pad_cache', {})
            key = (nb, x3.device, threading.get_ident())
            if key not in cache:
                cache[key] = (torch.zeros((nb,) + self.embed, dtype=self.rdtype, device=x3.device),
                              torch.empty((nb,) + self.embed, dtype=self.rdtype, device=x3.device))
            xs, os_ = cache[key]
            xs[:, :self.nx, :self.ny] = x3
            bs = None if beam is None else self._pad(beam, nb)
            ds = None if dot_with is None else self._pad(dot_with if dot_with.ndim == 3 else dot_with[None], nb)
            with self.lock:
                self._enter_stream()
                _lib.check(self._lib.pfb_psfconv_apply(
                    self._h, int(band0), int(nb), _dev.ptr(xs), _dev.ptr(bs),
                    float(wsum) if wsum is not None else 0.0, float(sigmainv), _dev.ptr(os_),
                    _dev.ptr(ds), _dev.ptr(dot_out), _dev.stream()))
                out3.copy_(os_[:, :self.nx, :self.ny])
            return out3[0] if squeeze else out3
        xs, bs, ds, os_ = x3, beam, dot_with, out3
        with self.lock:
            self._enter_stream()
            _lib.check(self._lib.pfb_psfconv_apply(
                self._h, int(band0), int(nb), _dev.ptr(xs), _dev.ptr(bs),
                float(wsum) if wsum is not None else 0.0, float(sigmainv), _dev.ptr(os_),
                _dev.ptr(ds), _dev.ptr(dot_out), _dev.stream()))
        return out3[0] if squeeze else out3

    def _enter_stream(self):
        """Call with `lock` held, BEFORE enqueueing on the current stream: the plan's workspace is about to be used
        there.  If the previous user enqueued on a DIFFERENT stream its work must finish first -- as a DEVICE-side
        dependency (an event recorded behind the previous user's enqueue, waited for by the current stream), never a
        host wait: with the reference's dask-thread pattern (several threads, one plan, a stream per thread,
        pcg.py:346-356) a host synchronize here would stall every thread queued on the lock until the previous thread's
        whole stream had drained.  Same stream: stream order already serialises the kernels."""
        cur = torch.cuda.current_stream()
        last = self.__dict__.get('_last_stream')
        if last is not None and last != cur:
            ev = torch.cuda.Event()
            ev.record(last)               # behind everything the previous user has enqueued so far (it holds no lock now)
            cur.wait_event(ev)
        self._last_stream = cur

    def set_profiling(self, on):
        """on: False/0 off, True/1 every apply, N > 1 every N-th apply (each timed apply puts
        four event records = ~20 us on the stream)."""
        _lib.check(self._lib.pfb_psfconv_set_profiling(self._h, int(on)))

    def get_profile(self):
        """(ms_row_fwd, ms_col, ms_row_inv) summed over `napply` applies, napply."""
        ms = (C.c_double * 3)()
        n = C.c_int()
        _lib.check(self._lib.pfb_psfconv_get_profile(self._h, ms, C.byref(n)))
        return tuple(ms), n.value

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            torch.cuda.synchronize()
            self._lib.pfb_psfconv_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------- plan cache
_cache = OrderedDict()
_cache_lock = threading.Lock()
_CACHE_MAX = 8


def _fingerprint(psfhat):
    if isinstance(psfhat, np.ndarray):
        flat = psfhat.reshape(-1)
        step = max(1, flat.size // 257)
        samp = flat[::step][:257]
        return ('np', psfhat.__array_interface__['data'][0], psfhat.shape, str(psfhat.dtype),
                complex(samp.sum()))
    return ('t', psfhat.data_ptr(), tuple(psfhat.shape), str(psfhat.dtype), psfhat._version,
            str(psfhat.device))


def plan_for(psfhat, nx, ny, lastsize):
    """Plan cache for the reference-shaped call sites (a functools.partial re-presents the same
    psfhat on every call).  numpy arrays are keyed on address + shape + a strided content sample;
    tensors on data_ptr + shape + version counter, and the cache entry keeps a reference to the
    tensor so that its memory cannot be freed and re-used by a DIFFERENT psfhat at the same
    address while the entry lives (clear_plan_cache() releases plans and references)."""
    key = (_fingerprint(psfhat), int(nx), int(ny), int(lastsize))
    with _cache_lock:
        hit = _cache.get(key)
        if hit is not None:
            _cache.move_to_end(key)
            return hit[0]
    plan = PsfConvPlan(psfhat, nx, ny, lastsize)
    with _cache_lock:
        _cache[key] = (plan, psfhat if isinstance(psfhat, torch.Tensor) else None)
        while len(_cache) > _CACHE_MAX:
            _cache.popitem(last=False)
    return plan


def clear_plan_cache():
    with _cache_lock:
        _cache.clear()


# ---------------------------------------------------------------- reference API
def _run(psfhat, lastsize, x, xout, beam=None, wsum=None, sigmainv=0.0):
    xd = _dev.to_dev(x)
    nx, ny = xd.shape[-2:]
    plan = plan_for(psfhat, nx, ny, lastsize)
    if xd.dtype != plan.rdtype:
        raise TypeError(f"x is {xd.dtype} but psfhat is {psfhat.dtype}")
    bd = _dev.to_dev(beam, plan.rdtype) if beam is not None else None
    direct = isinstance(xout, torch.Tensor) and xout.is_cuda and xout.is_contiguous() \
        and xout.dtype == plan.rdtype and tuple(xout.shape) == tuple(xd.shape) \
        and xout.data_ptr() != xd.data_ptr()
    res = plan.apply(xd, out=xout if direct else None, beam=bd, wsum=wsum, sigmainv=sigmainv)
    if direct:
        return xout
    if xout is None:
        return res.cpu().numpy() if _dev.is_numpy(x) else res
    if _dev.is_numpy(xout):
        xout[...] = res.cpu().numpy()
    else:
        xout.copy_(res)
    return xout


def psf_convolve_slice(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    """pfb/operators/psf.py:11-29.  x (nx, ny), psfhat (nx_psf, nyo2)."""
    if x.ndim != 2 or psfhat.ndim != 2:
        raise ValueError("psf_convolve_slice expects 2-D x and psfhat")
    return _run(psfhat, lastsize, x, xout)


def psf_convolve_cube(xpad, xhat, xout, psfhat, lastsize, x, nthreads=1):
    """pfb/operators/psf.py:32-56.  x (nband, nx, ny), psfhat (nband, nx_psf, nyo2)."""
    if x.ndim != 3 or psfhat.ndim != 3:
        raise ValueError("psf_convolve_cube expects 3-D x and psfhat")
    if x.shape[0] != psfhat.shape[0]:
        raise ValueError("x and psfhat disagree on the number of bands")
    return _run(psfhat, lastsize, x, xout)
