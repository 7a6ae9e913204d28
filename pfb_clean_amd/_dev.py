"""Device plumbing: numpy / torch interop, raw pointers, streams, scratch buffers.
PyTorch-ROCm is used for device memory, streams and torch.distributed only."""
import threading
import numpy as np
import torch

from . import _lib

_tls = threading.local()

_NP2T = {np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64,
         np.dtype('complex64'): torch.complex64, np.dtype('complex128'): torch.complex128}
REAL_OF = {torch.complex64: torch.float32, torch.complex128: torch.float64}
CPLX_OF = {torch.float32: torch.complex64, torch.float64: torch.complex128}


def require_device():
    if not torch.cuda.is_available():
        raise RuntimeError("pfb_clean_amd needs a ROCm GPU (MI355X); there is no CPU path")
    return torch.device('cuda', torch.cuda.current_device())


def is_numpy(a):
    return isinstance(a, np.ndarray)


def to_dev(a, dtype=None):
    """numpy array / CPU tensor / GPU tensor -> contiguous GPU tensor (copy only if
    needed)."""
    if a is None:
        return None
    dev = require_device()
    if isinstance(a, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=False)
    elif isinstance(a, torch.Tensor):
        t = a if a.is_cuda else a.to(dev)
    else:
        t = torch.as_tensor(a, device=dev)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def code(dtype):
    if dtype == torch.float32:
        return _lib.PFB_F32
    if dtype == torch.float64:
        return _lib.PFB_F64
    raise TypeError(f"unsupported dtype {dtype}: float32 / float64 only")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def scratch():
    """(ws, out) fp64 device scratch for the reduction kernels, one pair per host
    thread and stream (the reference may be driven from several dask threads)."""
    key = (torch.cuda.current_device(), stream())
    cache = getattr(_tls, 'scratch', None)
    if cache is None:
        cache = _tls.scratch = {}
    if key not in cache:
        dev = require_device()
        cache[key] = (torch.empty(_lib.REDUCE_WS_DOUBLES, dtype=torch.float64, device=dev),
                      torch.zeros(16, dtype=torch.float64, device=dev))
    return cache[key]


def give_back(t, like, out=None):
    """Return a result the way the caller handed data in: numpy in -> numpy out
    (written into `out` when given), tensor in -> tensor out."""
    if isinstance(like, np.ndarray):
        if out is not None and isinstance(out, np.ndarray):
            out[...] = t.cpu().numpy()
            return out
        return t.cpu().numpy()
    if out is not None and isinstance(out, torch.Tensor) and out is not t:
        out.copy_(t)
        return out
    return t
