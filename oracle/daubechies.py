"""
Daubechies orthonormal filter banks dbK, K = 1..9, built from first principles.

Test infrastructure (see oracle/__init__.py).  The reference takes its filters from
PyWavelets (pfb/operators/psi.py:37-43: pywt.Wavelet(name).filter_bank =
(dec_lo, dec_hi, rec_lo, rec_hi)); PyWavelets is NOT installed here and is un-pinned
in the reference's setup.py, so the filters are rebuilt by Daubechies' spectral
factorisation in extended precision (mpmath):

  P(y) = sum_{k<K} C(K-1+k, k) y^k,   y = (2 - z - 1/z)/4
  keep the K-1 roots z with |z| < 1 (minimum phase = PyWavelets' "db" convention up
  to time reversal), multiply by (1+z)^K, normalise sum(h) = sqrt(2).

The result satisfies sum_n h[n] h[n+2m] = delta_m to < 1e-15 (checked in
tests/test_oracle_golden.py) which is what the reference's own
tests/test_psi_operator.py:48 (perfect reconstruction to 1e-12) requires; published
17-digit tables only reach ~1e-12.  Orientation is chosen to
match PyWavelets: rec_lo is the sequence whose largest taps come first
(db2: rec_lo = [1+s3, 3+s3, 3-s3, 1-s3]/(4 sqrt 2)).

  dec_lo = rec_lo[::-1]
  dec_hi[k] = (-1)^(k+1) dec_lo[F-1-k]
  rec_hi = dec_hi[::-1]
"""
import functools
import numpy as np


@functools.lru_cache(maxsize=None)
def rec_lo(K, dps=60):
    import mpmath as mp
    mp.mp.dps = dps
    if K == 1:
        h = [mp.mpf(1), mp.mpf(1)]
    else:
        # polynomial in y, ascending coefficients
        coeffs = [mp.binomial(K - 1 + k, k) for k in range(K)]
        yroots = mp.polyroots(list(reversed(coeffs)), maxsteps=500, extraprec=200)
        zroots = []
        for y in yroots:
            # z^2 - (2 - 4y) z + 1 = 0
            bq = 2 - 4 * y
            disc = mp.sqrt(bq * bq - 4)
            z1 = (bq + disc) / 2
            z2 = (bq - disc) / 2
            zroots.append(z1 if abs(z1) < 1 else z2)
        # h(z) = (1+z)^K * prod (z - z_r)
        poly = [mp.mpc(1)]
        for _ in range(K):
            poly = _polymul(poly, [mp.mpc(1), mp.mpc(1)])
        for zr in zroots:
            poly = _polymul(poly, [-zr, mp.mpc(1)])
        h = [mp.re(c) for c in poly]
    s = sum(h)
    h = [c * mp.sqrt(2) / s for c in h]
    out = np.array([float(c) for c in h], dtype=np.float64)
    # PyWavelets orientation for rec_lo: energy concentrated at the start
    half = len(out) // 2
    if np.sum(out[:half] ** 2) < np.sum(out[half:] ** 2):
        out = out[::-1].copy()
    out.setflags(write=False)
    return out


def _polymul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            out[i + j] = out[i + j] + ai * bj
    return out


def filter_bank(name):
    """(dec_lo, dec_hi, rec_lo, rec_hi) for 'dbK' -- the tuple pywt exposes as
    Wavelet(name).filter_bank (psi.py:38-41)."""
    if not name.startswith('db'):
        raise ValueError(f"unsupported wavelet {name!r}")
    K = int(name[2:])
    rl = np.array(rec_lo(K))
    F = rl.size
    dl = rl[::-1].copy()
    dh = np.array([(-1) ** (k + 1) * dl[F - 1 - k] for k in range(F)])
    rh = dh[::-1].copy()
    return dl, dh, rl, rh


def dwt_max_level(data_len, name):
    """pywt.dwt_max_level: floor(log2(data_len / (filter_len - 1))), >= 0."""
    F = 2 * int(name[2:])
    if F < 2 or data_len < F - 1:
        return 0
    if F == 2:
        return int(np.floor(np.log2(data_len)))
    return max(int(np.floor(np.log2(data_len / (F - 1)))), 0)
