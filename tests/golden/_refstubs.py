"""
Stub modules that let the reference's hot-path source files be imported and EXECUTED in
the build container, where numba / ducc0 / pywt / numexpr / dask / distributed /
pyscilog are not installed.  Only third-party modules are
substituted; every line of pfb/ that runs is the reference's own.

  numba            -> identity decorators, prange = range (loops run as plain Python)
  ducc0.fft        -> scipy.fft (pocketfft): r2c(inorm=0)=rfftn, c2r(inorm=2)=irfftn
  ducc0.misc       -> make_noncritical = identity
  pywt             -> Wavelet(name).filter_bank from oracle.daubechies (exact tables)
  numexpr.evaluate -> eval in the caller's frame, honouring out=
  numba.extending.overload -> calls the generator with the arguments and runs what it returns
  dask / dask.array -> zeros / stack / compute on numpy (eager), for misc.py's dds2cubes
  daskms, omegaconf, skimage, africanus, xarray, quartical, jax -> inert placeholders (misc.py
                      imports them at module level; nothing exercised here calls them)

Used by tests/golden/make_golden.py ONLY, in the build container ONLY
(/root/reference does not exist on the GPU box).
"""
import sys
import types
import numpy as np
import scipy.fft as sfft


class _Any:
    """Callable, subscriptable placeholder for numba type objects."""

    def __call__(self, *a, **k):
        return _Any()

    def __getitem__(self, item):
        return _Any()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Any()


def _identity_decorator(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]

    def wrap(f):
        return f
    return wrap


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _r2c(a, axes=None, forward=True, inorm=0, out=None, nthreads=1):
    assert forward and inorm == 0
    res = sfft.rfftn(a, axes=axes)
    if out is not None:
        out[...] = res
        return out
    return res


def _c2r(a, axes=None, lastsize=0, forward=False, inorm=0, out=None, nthreads=1,
         allow_overwriting_input=False):
    assert (not forward) and inorm == 2
    axes = tuple(axes)
    s = [a.shape[ax] for ax in axes]
    s[-1] = lastsize
    res = sfft.irfftn(a, s=s, axes=axes)
    if out is not None:
        out[...] = res
        return out
    return res


def _c2c(a, axes=None, forward=True, inorm=0, out=None, nthreads=1):
    res = sfft.fftn(a, axes=axes) if forward else sfft.ifftn(a, axes=axes, norm='forward')
    if inorm == 2:
        n = np.prod([a.shape[ax] for ax in axes])
        res = res / n
    if out is not None:
        out[...] = res
        return out
    return res


def _evaluate(expr, local_dict=None, out=None, casting=None):
    frame = sys._getframe(1)
    env = dict(frame.f_globals)
    env.update(frame.f_locals)
    if local_dict:
        env.update(local_dict)
    res = eval(expr, {'__builtins__': {}}, env)
    if out is not None:
        out[...] = res
        return out
    return res


class _Log:
    def write(self, *a, **k):
        pass

    def flush(self):
        pass


def install(repo_root):
    """Insert the stub modules, then make /root/reference importable."""
    sys.path.insert(0, repo_root)
    from oracle import daubechies as db

    nb = _mod('numba', njit=_identity_decorator, jit=_identity_decorator, prange=range,
              int64=_Any(), float64=_Any(), float32=_Any(), complex128=_Any(),
              literally=lambda x: x)
    nb.types = _mod('numba.types', unicode_type=_Any(), ListType=_Any(), DictType=_Any(),
                    UniTuple=_Any(), int64=_Any(), float64=_Any())
    nb.typed = _mod('numba.typed', List=lambda *a: list(*a), Dict=dict)
    nb.experimental = _mod('numba.experimental',
                           jitclass=lambda spec=None: (lambda cls: cls))
    nb.extending = _mod('numba.extending', overload=_identity_decorator)

    class Wavelet:
        def __init__(self, name):
            self.name = name
            self.filter_bank = tuple(np.array(f) for f in db.filter_bank(name))

    _mod('pywt', Wavelet=Wavelet, dwt_max_level=db.dwt_max_level)

    d = _mod('ducc0')
    d.fft = _mod('ducc0.fft', r2c=_r2c, c2r=_c2r, c2c=_c2c,
                 good_size=lambda n, real=False: sfft.next_fast_len(n, real=real))
    d.misc = _mod('ducc0.misc', make_noncritical=lambda a: a,
                  roll_resize_roll=lambda *a, **k: None)
    d.wgridder = _mod('ducc0.wgridder')
    d.wgridder.experimental = _mod('ducc0.wgridder.experimental',
                                   vis2dirty=None, dirty2vis=None)

    class Array:        # dask.array.Array placeholder for isinstance checks
        pass

    dk = _mod('dask', delayed=_identity_decorator)
    dk.array = _mod('dask.array', Array=Array)
    _mod('distributed', wait=None, get_client=None, as_completed=None)
    _mod('numexpr', evaluate=_evaluate)
    _mod('pyscilog', get_logger=lambda name: _Log(), init=lambda *a: None)

    sys.path.insert(0, '/root/reference')
    import pfb  # noqa: F401  (no import-time side effects, pfb/__init__.py:33-137)
    pu = _mod('pfb.utils')
    # the reference's real misc.py (norm_diff for pcg / primal_dual, give_edges for hogbom's import)
    pu.misc = load_misc()
    sys.modules['pfb.utils.misc'] = pu.misc


class _AnyModule(types.ModuleType):
    """Module whose every attribute is an inert placeholder (names imported by misc.py that the
    functions exercised here never touch: daskms readers, skimage.label, jax, ...)."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Any()


def _overload(target, **kw):
    """numba.extending.overload stand-in: numba would compile gen(*types)(*args); here the generator
    is called with the arguments themselves (numpy arrays carry .ndim like the numba types do) and the
    implementation it returns is run as plain Python -- every line executed is the reference's."""
    def deco(gen):
        def dispatch(*args):
            return gen(*args)(*args)
        target.__globals__[target.__name__] = dispatch
        return gen
    return deco


def load_misc():
    """Load the reference's REAL pfb/utils/misc.py (as module 'pfb_ref_misc'; install() registers it as
    pfb.utils.misc) for norm_diff, l1reweight_func, dds2cubes, freqmul and setup_parametrisation.  Its ~20 absent third-party imports become inert placeholders, except
    dask / dask.array, which dds2cubes really uses: zeros / stack / compute map onto numpy (eager
    evaluation of the same graph)."""
    import importlib.util
    sys.modules['numba.extending'].overload = _overload
    sys.modules['numba'].jit = _identity_decorator
    sys.modules['ducc0.fft'].good_size = lambda n, real=False: sfft.next_fast_len(n, real=real)

    dk = sys.modules['dask']
    dk.compute = lambda *a, **k: tuple(a)
    da = sys.modules['dask.array']
    da.zeros = lambda shape, chunks=None, dtype=float: np.zeros(shape, dtype=dtype)
    da.stack = lambda seq, axis=0: np.stack(list(seq), axis=axis)
    for name in ('dask.distributed', 'dask.diagnostics', 'daskms', 'daskms.experimental',
                 'daskms.experimental.zarr', 'omegaconf', 'skimage', 'skimage.morphology',
                 'africanus', 'africanus.coordinates', 'africanus.coordinates.coordinates',
                 'xarray', 'quartical', 'quartical.utils', 'quartical.utils.dask',
                 'jax', 'jax.numpy'):
        if name not in sys.modules:
            sys.modules[name] = _AnyModule(name)
    spec = importlib.util.spec_from_file_location('pfb_ref_misc',
                                                  '/root/reference/pfb/utils/misc.py')
    mod = importlib.util.module_from_spec(spec)
    sys.modules['pfb_ref_misc'] = mod
    spec.loader.exec_module(mod)
    return mod
