"""
CPU restatement of the l21 proximal operators and the fused dual update.

Test infrastructure (see oracle/__init__.py).  Follows
  pfb/prox/prox_21m.py:5-27    prox_21m       (numpy form, "m" = band-SUM variant)
  pfb/prox/prox_21m.py:31-61   prox_21m_numba (writes `result`)
  pfb/prox/prox_21m.py:64-71   dual_update    (numpy form)
  pfb/prox/prox_21m.py:76-103  dual_update_numba (in place on v)
  pfb/prox/prox_21.py:5-20     prox_21        (band-l2-NORM variant)
  pfb/prox/prox_21.py:23-48    prox_21_numba  (band-l2-NORM variant, writes `result`)
  pfb/prox/prox_21.py:62-88    dual_update_numba of that file (here: dual_update_numba_l2)

The numba loops are vectorised over (basis, y, x); the band axis is axis 0.
"""
import numpy as np


def prox_21m(v, sigma, weight=1.0, axis=0):
    """prox_21m.py:5-27."""
    l2_norm = np.sum(v, axis=axis)
    l2_soft = np.maximum(np.abs(l2_norm) - sigma * weight, 0.0) * np.sign(l2_norm)
    mask = l2_norm != 0
    ratio = np.zeros(mask.shape, dtype=v.dtype)
    ratio[mask] = l2_soft[mask] / l2_norm[mask]
    return v * np.expand_dims(ratio, axis=axis)


def prox_21(v, sigma, weight=None, axis=0):
    """prox_21.py:5-20."""
    l2_norm = np.linalg.norm(v, axis=axis)
    l2_soft = np.maximum(l2_norm - sigma * weight, 0.0)
    mask = l2_norm != 0
    ratio = np.zeros(mask.shape, dtype=v.dtype)
    ratio[mask] = l2_soft[mask] / l2_norm[mask]
    return v * np.expand_dims(ratio, axis=axis)


def prox_21m_numba(v, result, lam, sigma=1.0, weight=None):
    """prox_21m.py:31-61: t = sum_band(v)/sigma; result = 0 where t == 0 else
    v * max(|t| - lam*w/sigma, 0) / |t| / sigma."""
    t = np.sum(v, axis=0) / sigma                     # (nbasis, nymax, nxmax)
    a = np.abs(t)
    soft = np.maximum(a - lam * weight / sigma, 0.0)
    nz = t != 0
    fac = np.zeros_like(a)
    fac[nz] = soft[nz] / a[nz] / sigma
    result[...] = v * fac[None]
    result[:, ~nz] = 0.0


def dual_update(v, x, psiH, lam, sigma=1.0, weight=1.0):
    """prox_21m.py:64-71 (numpy reference form used by test_dual_update)."""
    vp = v.copy()
    vout = np.zeros_like(v)
    psiH(x, vout)
    vtilde = vp + sigma * vout
    return vtilde - sigma * prox_21m(vtilde / sigma, lam / sigma, weight=weight)


def dual_update_numba(vp, v, lam, sigma=1.0, weight=None):
    """prox_21m.py:76-103, in place on v:
       vt = vp + sigma*v ; a = |sum_band vt / sigma| ; v = vt ;
       where a != 0: v *= 1 - max(a - lam*w/sigma, 0)/a."""
    vt = vp + sigma * v
    a = np.abs(np.sum(vt, axis=0) / sigma)
    soft = np.maximum(a - lam * weight / sigma, 0.0)
    nz = a != 0
    fac = np.ones_like(a)
    fac[nz] = 1.0 - soft[nz] / a[nz]
    v[...] = vt * fac[None]


def prox_21_numba(v, result, lam, sigma=1.0, weight=None):
    """prox_21.py:23-48: a = ||v[:, b, i]||_2 / sigma; result = 0 where a == 0 else
    v * max(a - lam*w/sigma, 0) / a / sigma."""
    a = np.linalg.norm(v, axis=0) / sigma
    soft = np.maximum(a - lam * weight / sigma, 0.0)
    nz = a != 0
    fac = np.zeros_like(a)
    fac[nz] = soft[nz] / a[nz] / sigma
    result[...] = v * fac[None]
    result[:, ~nz] = 0.0


def dual_update_numba_l2(vp, v, lam, sigma=1.0, weight=None):
    """prox_21.py:62-88, in place on v: vt = vp + sigma*v ; a = ||vt||_2 / sigma ; v = vt ;
       where a != 0: v *= 1 - max(a - lam*w/sigma, 0)/a."""
    vt = vp + sigma * v
    a = np.linalg.norm(vt, axis=0) / sigma
    soft = np.maximum(a - lam * weight / sigma, 0.0)
    nz = a != 0
    fac = np.ones_like(a)
    fac[nz] = 1.0 - soft[nz] / a[nz]
    v[...] = vt * fac[None]
