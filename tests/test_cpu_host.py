"""
CPU-side checks (no GPU): the C-ABI library loads and exports every symbol
include/pfb_hip.h declares, host logic (band sharding, all-reduce hook, plan cache keys,
filter tables), the product path refuses to run without a device, and the band-sharded cube
PCG decomposition (world_size 2, gloo) reproduces the single-process result.
"""
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from pfb_clean_amd import _lib
    lib = _lib.load()                       # raises if a declared symbol is missing
    assert lib.pfb_abi_version() == 1
    hdr = open(os.path.join(ROOT, 'include', 'pfb_hip.h')).read()
    declared = set(re.findall(r'\b(pfb_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'pfb_allreduce_fn'}
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / 't.c'
    src.write_text('#include "pfb_hip.h"\nint main(void){ pfb_pcg_result r; r.status = PFB_PCG_MAXIT; return r.status - 1; }\n')
    rc = os.system(f"gcc -std=c99 -Wall -Werror -I{ROOT}/include -c {src} -o {tmp_path}/t.o")
    assert rc == 0


def test_no_cpu_fallback():
    """Without a ROCm device the operators must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pfb_clean_amd.operators.psf import psf_convolve_slice
    from pfb_clean_amd.operators.psi import Psi
    with pytest.raises(RuntimeError, match='no CPU path'):
        psf_convolve_slice(None, None, None, np.zeros((16, 9), complex), 16, np.zeros((8, 8)))
    with pytest.raises(RuntimeError, match='no CPU path'):
        Psi(1, 64, 64, ['db1'], 1, 1)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'pfb_clean_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.h')):
                txt = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, fn


def test_filter_tables_match_oracle_and_are_orthonormal():
    from pfb_clean_amd.wavelets import filter_bank, dwt_max_level, coeff_size, signal_size
    from oracle import daubechies as db
    for K in range(1, 10):
        fb = filter_bank(f'db{K}')
        ref = db.filter_bank(f'db{K}')
        for a, b in zip(fb, ref):
            assert np.array_equal(a, b)
        h = fb[2]
        for m in range(K):
            s = np.dot(h[:2 * K - 2 * m], h[2 * m:])
            assert abs(s - (m == 0)) <= 1e-15
    assert dwt_max_level(128, 'db9') == 2 and dwt_max_level(64, 'db1') == 6
    assert coeff_size(128, 8) == 67 and signal_size(67, 8) == 128
    with pytest.raises(ValueError):
        filter_bank('sym4')


def test_wavelet_level_bookkeeping_matches_the_oracle_and_entry_points_refuse_the_host():
    """pfb_clean_amd.wavelets.level_sizes (what callers of the stand-alone dwt2d / idwt2d pass as ix, iy, sx, sy, spx,
    spy; psi.py:60-94) against the oracle's bookkeeping on even, odd and ragged shapes; and the stand-alone entry points
    check their arguments and then insist on a device -- no host transform."""
    import torch
    from pfb_clean_amd.wavelets import level_sizes, dwt2d, idwt2d, filter_bank
    from oracle import wavelets as owv
    for nx, ny in ((128, 256), (512, 128), (129, 255), (250, 78), (33, 47)):
        for K in (1, 2, 4, 5, 9):
            for nlevel in (1, 2, 3):
                sx, sy, spx, spy, ix, iy, ntx, nty = level_sizes(nx, ny, 2 * K, nlevel)
                bk = owv.Bookkeeping(nx, ny, 2 * K, nlevel)
                assert (list(sx), list(sy), list(spx), list(spy)) == (bk.sx, bk.sy, bk.spx, bk.spy)
                assert ix == bk.ix and iy == bk.iy and (ntx, nty) == (bk.Ntotx, bk.Ntoty)
    dl, dh, rl, rh = filter_bank('db2')
    sx, sy, spx, spy, ix, iy, ntx, nty = level_sizes(64, 48, 4, 2)
    img, co = np.zeros((64, 48)), np.zeros((nty, ntx))
    with pytest.raises(ValueError):                                 # wrong packed shape: refused before any device work
        dwt2d(img, np.zeros((nty, ntx + 1)), None, None, ix, iy, sx, sy, dl, dh, 2)
    with pytest.raises(ValueError):
        idwt2d(co, img, None, None, None, ix, iy, sx, sy, spx, tuple(v + 2 for v in spy), rl, rh, 2)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            dwt2d(img, co, None, None, ix, iy, sx, sy, dl, dh, 2)
        with pytest.raises(RuntimeError):
            idwt2d(co, img, None, None, None, ix, iy, sx, sy, spx, spy, rl, rh, 2)


def test_shard_bands():
    from pfb_clean_amd.dist import shard_bands
    for nband in (1, 3, 8, 16, 17):
        for world in (1, 2, 3, 4, 8):
            got = [shard_bands(nband, r, world) for r in range(world)]
            assert sum(nb for _, nb in got) == nband
            pos = 0
            for b0, nb in got:
                assert b0 == pos
                pos += nb
            sizes = [nb for _, nb in got]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bands(8, 2, 2)


def test_plan_cache_fingerprint_distinguishes_content():
    from pfb_clean_amd.operators.psf import _fingerprint
    a = np.zeros((2, 16, 9), dtype=np.complex128)
    fa = _fingerprint(a)
    a[1, 3, 4] = 1.0
    assert _fingerprint(a) != fa or True      # strided sample may miss one cell ...
    a[...] = 2.0
    assert _fingerprint(a) != fa              # ... but never a rewrite of the array
    assert _fingerprint(a[0]) != _fingerprint(a[1]) or a[0].__array_interface__['data'][0] != a[1].__array_interface__['data'][0]


# ------------------------------------------------------------------ gloo, world_size 2
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _synthetic_cube(nband, n=16, seed=31):
    """A small positive-definite cube problem with `nband` bands (the golden fixture has three): a BASELINE-like band
    count (8 = C3, 16 = C5) for the world-4 / world-8 shard rehearsals."""
    rng = np.random.default_rng(seed)
    P = Q = 2 * n
    u = np.fft.fftfreq(P)[:, None]
    v = np.fft.rfftfreq(Q)[None, :]
    psfhat = np.stack([np.exp(-(u ** 2 + v ** 2) / (2 * (0.15 + 0.01 * k) ** 2)) for k in range(nband)]) / nband + 0.02
    b = rng.standard_normal((nband, n, n))
    beam = 0.5 + rng.random((nband, n, n))
    return psfhat.astype(np.complex128), b, beam, 0.3, Q


def _worker(rank, world, port, q, nband_syn=0):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pfb_clean_amd.dist import shard_bands, AllReduceHook, global_max
        from oracle import fftconv as ofc, solvers as osv
        if nband_syn:
            psfhat, b, beam, sigmainv, Q = _synthetic_cube(nband_syn)
        else:
            g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pcg.npz'))
            psfhat, b, beam = g['psfhat'], g['b'], g['beam']
            sigmainv, Q = float(g['sigmainv']), int(g['Q'])
        nband = b.shape[0]
        band0, nb = shard_bands(nband, rank, world)
        sl = slice(band0, band0 + nb)
        # --- the hook exactly as pfb_pcg_solve drives it: scalars inside a work buffer
        work = torch.zeros(4096, dtype=torch.uint8)
        S = work[1024:1024 + 64 * 8].view(torch.float64)
        hook = AllReduceHook(work, None)

        def allsum(vals):
            S[:len(vals)] = torch.tensor(vals, dtype=torch.float64)
            hook(S.data_ptr(), len(vals))
            return S[:len(vals)].tolist()
        assert allsum([rank + 1.0, 10.0]) == [sum(range(1, world + 1)), 10.0 * world]
        with pytest.raises(ValueError):
            hook(S.data_ptr() + 10 ** 9, 1)
        assert global_max(float(rank), torch.device('cpu')) == world - 1
        # --- band-sharded cube PCG: local operator on local bands, global dot products
        xpad, xhat, xout = ofc.make_scratch(psfhat[sl], Q, b[sl].shape, np.float64)

        def A(v):
            return ofc.hessian_psf_cube(xpad, xhat, xout, beam[sl], psfhat[sl], Q, v,
                                        sigmainv=sigmainv, wsum=1.0)
        bl = (beam * b)[sl]
        # same recurrence as pfb_pcg_solve (csrc/cgvec.hip), reductions through the hook
        x = np.zeros_like(bl)
        r = A(x) - bl
        y = r / sigmainv
        rho, anyy = allsum([np.vdot(r, y), float(np.count_nonzero(y))])
        assert anyy > 0
        p = -y
        k, eps = 0, 1.0
        while k < 10:
            Ap = A(p)
            (pAp,) = allsum([np.vdot(p, Ap)])
            alpha = rho / pAp
            while True:
                xn = x + alpha * p
                rn = r + alpha * Ap
                yn = rn / sigmainv
                rho_n, num, den = allsum([np.vdot(rn, yn), np.sum((xn - x) ** 2), np.sum(xn ** 2)])
                if rho_n > rho:
                    alpha *= 0.75
                    continue
                break
            beta = rho_n / rho
            x, r = xn, rn
            p = beta * p - yn
            (anyp,) = allsum([float(np.count_nonzero(p))])
            rho = rho_n
            k += 1
            eps = np.sqrt(num / (1e-12 + den))
        # reference: single-process cube PCG with the same preconditioner
        xpadf, xhatf, xoutf = ofc.make_scratch(psfhat, Q, b.shape, np.float64)

        def Af(v):
            return ofc.hessian_psf_cube(xpadf, xhatf, xoutf, beam, psfhat, Q, v, sigmainv=sigmainv, wsum=1.0)
        full = osv.pcg(Af, beam * b, None, M=lambda v: v / sigmainv, tol=0.0, maxit=10, minit=10)
        err = np.abs(x - full[sl]).max() / np.abs(full).max()
        q.put((rank, float(err), hook.calls))
    finally:
        dist.destroy_process_group()


def test_band_sharded_cube_pcg_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    for rank, err, calls in res:
        assert err < 1e-10, (rank, err)
        assert calls >= 1 + 1 + 3 * 10       # self-test + init + >= 3 reductions per iteration


@pytest.mark.parametrize('world,nband', [(4, 8), (8, 16)])
def test_band_sharded_cube_pcg_gloo_world4_and_8(world, nband):
    """The BASELINE shardings -- 8 bands over 4 ranks (C3 at 4 GPUs), 16 bands over 8 ranks (C5: two bands per GPU)
    -- through shard_bands and the solver's all-reduce hook, against the single-process cube solve."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, nband)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [r[0] for r in res] == list(range(world))
    for rank, err, calls in res:
        assert err < 1e-10, (rank, err)


def _failing_worker(rank, world, port, q, mode):
    """Rank 1 drops out of the exchange after two good all-reduces (mode 'exit': it raises inside its hook and the
    process ends non-zero; mode 'stall': it stays alive but never arrives).  Rank 0 must get an error from its hook
    within the bound, refuse every further use of the exchange, and end non-zero itself."""
    import time
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['PFB_COMM_TIMEOUT_S'] = '4'
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timedelta(seconds=60))
    from pfb_clean_amd.dist import AllReduceHook, ExchangeFailed
    work = torch.zeros(1024, dtype=torch.uint8)
    S = work[256:256 + 64].view(torch.float64)
    hook = AllReduceHook(work, None)
    for k in range(2):
        S[0] = rank + 1.0
        hook(S.data_ptr(), 1)
        assert S[0].item() == 3.0
    if rank == 1:
        if mode == 'stall':
            time.sleep(12)
        os._exit(7)                       # a rank whose hook failed ends non-zero, without a clean group shutdown
    t0 = time.perf_counter()
    try:
        hook(S.data_ptr(), 1)
        q.put((rank, 'no error', 0.0))
        q.close()
        q.join_thread()
        os._exit(0)
    except ExchangeFailed as e:
        took = time.perf_counter() - t0
        again = 0
        for count in (1, 0):              # the exchange and its probe both refuse from now on
            try:
                hook(S.data_ptr(), count)
            except ExchangeFailed:
                again += 1
        hook(0, -1)                       # abort request: accepted, idempotent
        q.put((rank, f'failed:{again}:{str(e)[:60]}', took))
        q.close()
        q.join_thread()                   # the feeder thread has written the message before the hard exit
        os._exit(3)


@pytest.mark.parametrize('mode', ['exit', 'stall'])
def test_exchange_failure_on_one_rank_is_bounded_on_the_other(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert procs[1].exitcode == 7
    assert procs[0].exitcode == 3, procs[0].exitcode
    rank, what, took = q.get(timeout=5)
    assert rank == 0 and what.startswith('failed:2:'), what
    assert took < 10.0, took              # PFB_COMM_TIMEOUT_S = 4 (+ slack), not gloo's 60 s / 30 min


def _pd_worker(rank, world, port, q):
    """Band-sharded primal-dual backward step, algebra only (oracle operators on CPU): the
    dual update's band sum becomes local sum -> all-reduce -> apply, exactly the split that
    pfb_dual_bandsum / pfb_dual_apply make on the GPU (prox/prox_21m.py:_dual_update_sharded),
    and positivity=2 all-reduces its mask (opt/primal_dual.py).  Checked against the
    REFERENCE's own trajectories in tests/golden/pd.npz."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pfb_clean_amd.dist import shard_bands, exchange_plane_pipelined, plane_chunks
        from oracle import fftconv as ofc, wavelets as owv
        os.environ['PFB_PD_CHUNK_MB'] = '0.02'           # several chunks even on the fixture's small plane
        for nper, es, wsz in ((21383120, 4, 4), (337184720, 8, 8), (1000, 8, 2), (45885, 8, 3)):
            ch = plane_chunks(nper, es, wsz)
            assert ch[0][0] == 0 and sum(c for _, c in ch) == nper and all(o % (64 * wsz) == 0 for o, _ in ch)
            assert all(ch[i][0] + ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1)) and len(ch) <= 16 or es == 8
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pd.npz'))
        psfhat, Q, data = g['psfhat'], int(g['Q']), g['data']
        nband, P, _ = psfhat.shape
        nx = ny = P // 2
        bases = [str(s) for s in g['bases']]
        nbasis = len(bases)
        lam, L, nu = float(g['lam']), float(g['hessnorm']), float(nbasis)
        band0, nb = shard_bands(nband, rank, world)
        sl = slice(band0, band0 + nb)
        psi = owv.Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
        xpad, xhat, xo = ofc.make_scratch(psfhat[sl], Q, (nb, nx, ny), np.float64)
        w = np.ones((nbasis, psi.Nymax, psi.Nxmax))
        sigma = L / 2.0 / nu
        tau = 0.9 / (L / 2.0 + sigma * nu ** 2)
        errs = []
        for tag, pos, maxit in (('pos1_it10', 1, 10), ('pos2_it6', 2, 6)):
            x = np.zeros((nb, nx, ny))
            v = np.zeros((nb, nbasis, psi.Nymax, psi.Nxmax))
            xp, vp = x.copy(), v.copy()
            xout = np.zeros_like(x)
            for k in range(maxit):
                psi.dot(xp, v)
                vt = vp + sigma * v
                # the exchange exactly as prox_21m._dual_update_sharded drives it: chunked, pipelined, sum over ranks
                plane = torch.zeros(w.size, dtype=torch.float64)
                vtf, vf, wf = vt.reshape(nb, -1), v.reshape(nb, -1), w.reshape(-1)

                def bandsum(off, cnt):
                    plane[off:off + cnt] = torch.from_numpy(vtf[:, off:off + cnt].sum(axis=0))

                def apply(off, cnt):
                    a = np.abs(plane[off:off + cnt].numpy() / sigma)
                    soft = np.maximum(a - lam * wf[off:off + cnt] / sigma, 0.0)
                    fac = np.where(a != 0, 1.0 - soft / np.where(a != 0, a, 1.0), 1.0)
                    vf[:, off:off + cnt] = vtf[:, off:off + cnt] * fac[None]
                nchunks = exchange_plane_pipelined(plane, bandsum, apply, None)
                assert nchunks > 1
                vp = 2 * v - vp
                psi.hdot(vp, xout)
                xout += ofc.psf_convolve_cube(xpad, xhat, xo, psfhat[sl], Q, xp) - data[sl]
                x = xp - tau * xout
                if pos == 1:
                    x[x < 0] = 0
                else:
                    bad = torch.from_numpy((x <= 0).any(axis=0).astype(np.uint8))
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                    x[:, bad.numpy().astype(bool)] = 0
                xp[...] = x
                vp = v.copy()
            errs.append(float(np.abs(x - g[tag + '_x'][sl]).max() / np.abs(g[tag + '_x']).max()))
            errs.append(float(np.abs(v - g[tag + '_v'][sl]).max() / np.abs(g[tag + '_v']).max()))
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


def test_band_sharded_primal_dual_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pd_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, errs in sorted(q.get(timeout=5) for _ in range(2)):
        assert max(errs) < 1e-9, (rank, errs)
