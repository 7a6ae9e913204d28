"""
ctypes binding of libpfb_hip.so (C-ABI declared in include/pfb_hip.h).

The product path has NO CPU fallback: if the shared library is missing or does not
export a symbol this module raises ImportError / AttributeError, and every compute
call requires a ROCm device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PFB_HIP_LIB') or os.path.join(_HERE, 'libpfb_hip.so')   # override: A/B builds

PFB_F32, PFB_F64 = 0, 1
PFB_OK = 0
ERRORS = {-1: 'invalid argument', -2: 'unsupported size/dtype', -3: 'HIP runtime error',
          -4: 'non-finite value', -5: 'device allocation failed', -6: 'band-shard exchange failed'}
PCG_STATUS = {0: 'converged', 1: 'maxit', 2: 'zero-residual', 3: 'breakdown'}
REDUCE_WS_DOUBLES = 8192


PFB_ERR_INVALID = -1
PFB_ERR_UNSUPPORTED = -2
PFB_ERR_ALLOC = -5
PFB_ERR_COMM = -6


class PfbHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpfb_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


class PfbCommError(PfbHipError):
    """PFB_ERR_COMM: the band-shard exchange of a distributed solve failed, was aborted or timed out.  The exchange has
    been torn down; this rank holds no result and should exit non-zero (its peers get the same error)."""


class PcgResult(C.Structure):
    _fields_ = [('status', C.c_int), ('iters', C.c_int), ('matvecs', C.c_int),
                ('backtracks', C.c_int), ('eps', C.c_double), ('rnorm', C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

# name -> (restype, argtypes); every symbol include/pfb_hip.h declares
_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
SIGNATURES = {
    'pfb_abi_version': (_i, []),
    'pfb_last_error': (C.c_char_p, []),
    'pfb_psfconv_plan_create': (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    'pfb_psfconv_plan_destroy': (_i, [_vp]),
    'pfb_psfconv_set_psfhat': (_i, [_vp, _vp, _vp]),
    'pfb_psfconv_apply': (_i, [_vp, _i, _i, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp]),
    'pfb_psfconv_set_psf': (_i, [_vp, _vp, _vp, _vp]),
    'pfb_psfhat_regrid': (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'pfb_psfhat_from_psf': (_i, [_i, _vp, _i, _i, _i, _vp, _vp]),
    'pfb_psfconv_apply_dots': (_i, [_vp, _i, _i, _vp, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp]),
    'pfb_psfconv_plan_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz)]),
    'pfb_psfconv_set_profiling': (_i, [_vp, _i]),
    'pfb_psfconv_get_profile': (_i, [_vp, C.POINTER(_d), C.POINTER(_i)]),
    'pfb_dot': (_i, [_i, _vp, _vp, _sz, _vp, _vp, _vp]),
    'pfb_norm_diff_sums': (_i, [_i, _vp, _vp, _sz, _vp, _vp, _vp]),
    'pfb_any_nonzero': (_i, [_i, _vp, _sz, _vp, _vp, _vp]),
    'pfb_axpby': (_i, [_i, _d, _vp, _d, _vp, _sz, _vp]),
    'pfb_pcg_work_bytes': (_sz, [_vp, _i]),
    'pfb_pcg_solve': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _i, _i, _i,
                           _vp, ALLREDUCE_FN, _vp, C.POINTER(PcgResult), _vp]),
    'pfb_comm_bind': (_i, [C.c_char_p]),
    'pfb_comm_unique_id': (_i, [_vp]),
    'pfb_comm_init': (_i, [_i, _i, _vp, C.POINTER(_vp)]),
    'pfb_comm_destroy': (_i, [_vp]),
    'pfb_comm_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'pfb_comm_allreduce': (_i, [_vp, _vp, _i, _vp]),
    'pfb_comm_check': (_i, [_vp]),
    'pfb_comm_abort': (_i, [_vp]),
    'pfb_psi_plan_create': (_i, [_i, _i, _i, _i, C.POINTER(_i), C.POINTER(_d), _i, _i,
                                 C.POINTER(_vp)]),
    'pfb_psi_plan_destroy': (_i, [_vp]),
    'pfb_psi_plan_dims': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'pfb_psi_dot': (_i, [_vp, _vp, _vp, _vp]),
    'pfb_psi_hdot': (_i, [_vp, _vp, _vp, _vp]),
    'pfb_dual_update': (_i, [_i, _vp, _vp, _vp, _d, _d, _i, _sz, _vp, _vp]),
    'pfb_dual_bandsum': (_i, [_i, _vp, _vp, _d, _i, _sz, _vp, _vp]),
    'pfb_dual_apply': (_i, [_i, _vp, _vp, _vp, _vp, _d, _d, _i, _sz, _vp, _vp]),
    'pfb_prox_21': (_i, [_i, _vp, _vp, _vp, _d, _d, _i, _sz, _vp]),
    'pfb_dual_update_l2': (_i, [_i, _vp, _vp, _vp, _d, _d, _i, _sz, _vp]),
    'pfb_dual_bandsum_chunk': (_i, [_i, _vp, _vp, _d, _i, _sz, _sz, _vp, _vp]),
    'pfb_dual_apply_chunk': (_i, [_i, _vp, _vp, _vp, _vp, _d, _d, _i, _sz, _sz, _vp, _vp]),
    'pfb_clark_subminor': (_i, [_i, _vp, _sz, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _d, _d, _i, _vp, _vp]),
    'pfb_hogbom': (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _i, _vp, _sz, _vp, _vp, _vp]),
    'pfb_freqmul': (_i, [_i, _vp, _vp, _vp, _i, _sz, _vp, _vp, _vp]),
    'pfb_prox_21m': (_i, [_i, _vp, _vp, _vp, _d, _d, _i, _sz, _vp]),
    'pfb_pd_primal_update': (_i, [_i, _vp, _vp, _vp, _d, _i, _i, _sz, _vp, _vp, _vp, _vp]),
    'pfb_pd_primal_update2': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _sz, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load (once) and type the library.  Raises ImportError when it has not been built:
    run `python -c "import __graft_entry__ as g; g.build()"` or `make -C pfb_clean_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled libamdhip64 must be the HIP runtime of the process (loading
    # ours first binds /opt/rocm's copy and torch then sees "no ROCm-capable device")
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C pfb_clean_amd/csrc` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != PFB_OK:
        msg = load().pfb_last_error()
        raise (PfbCommError if code == PFB_ERR_COMM else PfbHipError)(code, msg.decode() if msg else '')
    return code
