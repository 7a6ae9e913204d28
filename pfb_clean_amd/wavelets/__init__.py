"""Filter tables + size helpers of pfb/wavelets (the transforms themselves run inside
operators.psi.Psi; pfb.wavelets.{dwt2d,idwt2d} are numba-internal entry points that no
worker imports)."""
from .filters import filter_bank, dwt_max_level  # noqa: F401


def coeff_size(nsignal, nfilter):
    """pfb/wavelets/wavelets.py:21-22"""
    return (nsignal + nfilter - 1) // 2


def signal_size(ncoeff, nfilter):
    """pfb/wavelets/wavelets.py:26-27"""
    return 2 * ncoeff - nfilter + 2
