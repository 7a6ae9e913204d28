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
  pfb.utils.misc   -> norm_diff restated (the real module needs ~20 absent packages)

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


def _norm_diff(x, xp):
    # pfb/utils/misc.py:1326-1351
    if x.ndim not in (2, 3):
        raise ValueError("norm_diff is only implemented for 2D or 3D arrays")
    num = float(np.sum((x.astype(np.float64) - xp.astype(np.float64)) ** 2))
    den = 1e-12 + float(np.sum(x.astype(np.float64) ** 2))
    return np.sqrt(num / den)


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
    # give_edges: imported by deconv/hogbom.py:3 but never called there
    pu.misc = _mod('pfb.utils.misc', norm_diff=_norm_diff, give_edges=None)
