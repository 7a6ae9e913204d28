"""Randomised psi.dot / psi.hdot sweep against the CPU oracle: random even image sizes (tiny to a few tiles,
partial edge tiles), random basis sets out of self + db1..db9, 1-3 levels, both precisions.
Development aid: python tools/stress_psi.py [ncases] [seed]"""
import sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wavelets as owv
from oracle import daubechies as db
from pfb_clean_amd.operators.psi import Psi

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
names = ['self'] + [f'db{k}' for k in range(1, 10)]
worst = {np.float64: 0.0, np.float32: 0.0}
done = 0
while done < ncases:
    nx, ny = 2 * int(rng.integers(8, 200)), 2 * int(rng.integers(8, 200))
    nb = int(rng.integers(1, 4))
    bases = list(rng.choice(names, size=int(rng.integers(1, 5)), replace=False))
    nl = int(rng.integers(1, 4))
    wav = [b for b in bases if b != 'self']
    if not wav:               # 'self' alone: the reference sizes the coefficient planes from the wavelet bases (psi.py:75-78)
        continue
    if wav and nl > min(db.dwt_max_level(min(nx, ny), w) for w in wav):
        continue
    dt = np.float64 if rng.random() < 0.5 else np.float32
    x = rng.standard_normal((nb, nx, ny)).astype(dt)
    po = owv.Psi(nb, nx, ny, bases, nl)
    a_ref = np.zeros((nb, po.nbasis, po.Nymax, po.Nxmax))
    po.dot(x.astype(np.float64), a_ref)
    psi = Psi(nb, nx, ny, bases, nl, 1)
    a = np.zeros(a_ref.shape, dtype=dt)
    psi.dot(x, a)
    c = rng.standard_normal(a_ref.shape).astype(dt)
    xo_ref = np.zeros((nb, nx, ny))
    po.hdot(c.astype(np.float64), xo_ref)
    xo = np.full((nb, nx, ny), np.nan, dtype=dt)
    psi.hdot(c, xo)
    e1 = np.abs(a - a_ref).max() / max(np.abs(a_ref).max(), 1e-300)
    e2 = np.abs(xo - xo_ref).max() / max(np.abs(xo_ref).max(), 1e-300)
    tol = 1e-12 if dt == np.float64 else 3e-5
    flag = '' if (e1 < tol and e2 < tol) else '   <-- FAIL'
    print(f"{done:3d} nb={nb} ({nx},{ny}) {bases} nl={nl} {dt.__name__}: dot {e1:.2e} hdot {e2:.2e}{flag}")
    assert not flag
    worst[dt] = max(worst[dt], e1, e2)
    done += 1
print('worst', {k.__name__: v for k, v in worst.items()})
