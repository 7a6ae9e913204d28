"""
GPU tests for (1) the RCCL all-reduce hook of the fused PCG, exercised with a real `nccl`
process group of world size 1 (the only multi-process GPU configuration a 1-GPU box
allows; world size 2 runs on CPU/gloo in tests/test_cpu_host.py), and (2) the BASELINE
config #4 composition -- power_method -> pcg -> primal_dual_optimised with psi/psi^H --
at reduced size against the CPU oracle.
"""
import os
import socket
from functools import partial

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import fftconv as ofc, solvers as osv, wavelets as owv   # noqa: E402 (checker only)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _psd_psfhat(rng, nb, P, Q):
    u = np.fft.fftfreq(P)[:, None]
    v = np.fft.rfftfreq(Q)[None, :]
    W = rng.poisson(4 * np.exp(-(u ** 2 + v ** 2) / (2 * 0.12 ** 2)), size=(nb, P, Q // 2 + 1)).astype(np.float64)
    W /= nb * np.fft.irfft2(W, s=(P, Q)).max(axis=(1, 2))[:, None, None]
    return W.astype(np.complex128)


def test_pcg_allreduce_hook_with_rccl_world1():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    from pfb_clean_amd.operators.hessian import HessianPsf
    from pfb_clean_amd.opt.pcg import pcg_fused
    rng = np.random.default_rng(3)
    nb, nx, ny = 2, 128, 128
    psfhat = _psd_psfhat(rng, nb, 2 * nx, 2 * ny)
    b = rng.standard_normal((nb, nx, ny))
    A = HessianPsf(torch.from_numpy(psfhat).cuda(), nx, ny, 2 * ny, sigmainv=1e-2)
    bt = torch.from_numpy(b).cuda()
    x_ref, _, res_ref = pcg_fused(A, bt, None, mdiv=1e-2, tol=0.0, maxit=8, minit=8)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1)
    from pfb_clean_amd import dist as pdist
    try:
        # (1) the library's own RCCL communicator: all-reduce enqueued from C on the solver's stream
        comm = pdist.native_comm()
        assert comm is not None and comm.world == 1 and comm.rank == 0 and comm.rccl_version > 20000
        t = torch.arange(7, dtype=torch.float64, device='cuda') + 0.5
        assert torch.equal(comm.all_reduce_(t.clone()), t)
        x, _, res = pcg_fused(A, bt, None, mdiv=1e-2, tol=0.0, maxit=8, minit=8, distributed=True)
        assert res.exchange == 'rccl-native'
        x1, _, res1 = pcg_fused(A, bt, None, mdiv=1e-2, tol=0.0, maxit=8, minit=8, distributed=True,
                                backtrack='exact')
        # (1b) the band-sharded dual update over the same group: chunked, pipelined reduce-scatter + all-gather on RCCL
        # (dist.exchange_plane_pipelined) -- one rank: bitwise the unsharded kernel's result, odd tail chunk included
        from pfb_clean_amd.prox.prox_21m import dual_update_numba
        os.environ['PFB_PD_CHUNK_MB'] = '0.05'
        try:
            gq = torch.Generator(device='cuda').manual_seed(5)
            shp = (3, 4, 67, 133)
            vp_ = torch.randn(shp, generator=gq, device='cuda', dtype=torch.float64)
            v_ = torch.randn(shp, generator=gq, device='cuda', dtype=torch.float64)
            w_ = torch.rand(shp[1:], generator=gq, device='cuda', dtype=torch.float64)
            assert len(pdist.plane_chunks(v_[0].numel(), 8, 1)) > 2
            va, vb = v_.clone(), v_.clone()
            oa, ob = torch.empty_like(v_), torch.empty_like(v_)
            dual_update_numba(vp_, va, 0.3, sigma=1.7, weight=w_, vp_out=oa)
            dual_update_numba(vp_, vb, 0.3, sigma=1.7, weight=w_, vp_out=ob, group=True)
            assert torch.equal(va, vb) and torch.equal(oa, ob)
        finally:
            del os.environ['PFB_PD_CHUNK_MB']
        # (1c) failure protocol of the native exchange (include/pfb_hip.h): healthy -> probe OK; aborted -> the probe,
        # the exchange and a distributed solve all answer PFB_ERR_COMM instead of enqueueing on a dead communicator
        from pfb_clean_amd import _lib
        lib = _lib.load()
        comm.check()
        assert lib.pfb_comm_allreduce(comm.handle, None, 0, None) == 0
        comm.abort()
        assert lib.pfb_comm_check(comm.handle) == _lib.PFB_ERR_COMM
        assert lib.pfb_comm_allreduce(comm.handle, None, 0, None) == _lib.PFB_ERR_COMM
        assert lib.pfb_comm_allreduce(comm.handle, t.data_ptr(), 7, None) == _lib.PFB_ERR_COMM
        with pytest.raises(_lib.PfbCommError):
            pcg_fused(A, bt, None, mdiv=1e-2, tol=0.0, maxit=8, minit=8, distributed=True)
        # (2) the fallback: ctypes callback -> torch.distributed.all_reduce
        pdist.close_native_comms()
        os.environ['PFB_NATIVE_COMM'] = '0'
        try:
            assert pdist.native_comm() is None
            xh, _, resh = pcg_fused(A, bt, None, mdiv=1e-2, tol=0.0, maxit=8, minit=8, distributed=True)
            assert resh.exchange == 'torch-hook' and resh.hook_calls > 0
        finally:
            del os.environ['PFB_NATIVE_COMM']
            pdist.close_native_comms()
    finally:
        pdist.close_native_comms()
        dist.destroy_process_group()
    assert res.iters == res_ref.iters == resh.iters == 8
    assert torch.equal(x, x_ref)                   # sum over one rank: bitwise identical
    assert torch.equal(xh, x_ref)
    assert (x1 - x_ref).abs().max().item() < 1e-12 * x_ref.abs().max().item()


def test_comm_entry_points_reject_bad_arguments():
    from pfb_clean_amd import _lib
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    ids = bytes(128)
    assert lib.pfb_comm_init(0, 0, ids, C.byref(h)) == _lib.PFB_ERR_INVALID
    assert lib.pfb_comm_init(2, 2, ids, C.byref(h)) == _lib.PFB_ERR_INVALID
    assert lib.pfb_comm_init(0, 1, None, C.byref(h)) == _lib.PFB_ERR_INVALID
    assert lib.pfb_comm_unique_id(None) == _lib.PFB_ERR_INVALID
    assert lib.pfb_comm_allreduce(None, None, 4, None) == _lib.PFB_ERR_INVALID
    assert lib.pfb_comm_info(None, None, None, None, None) == _lib.PFB_ERR_INVALID
    assert lib.pfb_comm_destroy(None) == 0


def test_fwdbwd_composition_config4_reduced():
    """power_method (hessnorm) -> pcg (forward step) -> primal_dual_optimised (backward
    step) as in workers/fwdbwd.py:310-453 with the live operators, 2 bands x 128^2,
    bases self+db1..db3, 2 levels, everything GPU resident; oracle runs the same recipe."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.operators.psf import psf_convolve_cube
    from pfb_clean_amd.operators.hessian import HessianPsf
    from pfb_clean_amd.operators.psi import Psi
    from pfb_clean_amd.opt.pcg import pcg, DivPrecond
    from pfb_clean_amd.opt.power_method import power_method
    from pfb_clean_amd.opt.primal_dual import primal_dual_optimised
    rng = np.random.default_rng(12)
    nb, nx, ny = 2, 128, 128
    P, Q = 2 * nx, 2 * ny
    bases = ['self', 'db1', 'db2', 'db3']
    nbasis, nlevel = len(bases), 2
    psfhat = _psd_psfhat(rng, nb, P, Q)
    truth = np.zeros((nb, nx, ny))
    truth[:, 40, 50] = 1.0
    truth[:, 80:84, 30:34] = 0.3
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, truth.shape, np.float64)
    oconv = partial(ofc.psf_convolve_cube, xpad, xhat, xout, psfhat, Q)
    dirty = oconv(truth).copy() + 1e-4 * rng.standard_normal(truth.shape)
    sigmainv = 1e-3 * np.abs(dirty).max()
    b0 = rng.standard_normal(truth.shape)
    lam = 5e-4

    # ---------------- oracle
    L_ref, _ = osv.power_method(oconv, truth.shape, b0=b0.copy(), tol=1e-6, maxit=60)

    def oA(v):
        return ofc.hessian_psf_cube(xpad, xhat, xout, None, psfhat, Q, v, sigmainv=sigmainv)
    xf_ref = osv.pcg(oA, dirty, None, M=lambda v: v / sigmainv, tol=0.0, maxit=15, minit=15)
    opsi = owv.Psi(nb, nx, ny, bases, nlevel, 1)
    data = oconv(xf_ref).copy()
    ov = np.zeros((nb, nbasis, opsi.Nymax, opsi.Nxmax))
    xb_ref, vb_ref = osv.primal_dual_optimised(xf_ref.copy(), ov, lam, opsi.hdot, opsi.dot, 1.05 * L_ref,
                                               None, np.ones(ov.shape[1:]), None,
                                               lambda v: oconv(v) - data, nu=nbasis, tol=0.0, maxit=12,
                                               positivity=1)
    # ---------------- device
    dev = torch.device('cuda')
    ph = torch.from_numpy(psfhat).to(dev)
    conv = partial(psf_convolve_cube, None, None, None, ph, Q)
    L, _ = power_method(conv, truth.shape, b0=torch.from_numpy(b0).to(dev), tol=1e-6, maxit=60, verbosity=0)
    assert abs(L - L_ref) < 1e-10 * L_ref
    A = HessianPsf(ph, nx, ny, Q, sigmainv=sigmainv)
    xf = pcg(A, torch.from_numpy(dirty).to(dev), None, M=DivPrecond(sigmainv), tol=0.0, maxit=15, minit=15,
             verbosity=0)
    assert (xf.cpu().numpy() - xf_ref).__abs__().max() < 1e-9 * np.abs(xf_ref).max()
    psi = Psi(nb, nx, ny, bases, nlevel, 1)
    datad = conv(xf).clone()
    v = torch.zeros((nb, nbasis, psi.Nymax, psi.Nxmax), dtype=torch.float64, device=dev)
    xb, vb = primal_dual_optimised(xf.clone(), v, lam, psi.hdot, psi.dot, 1.05 * L, None,
                                   torch.ones_like(v[0]), None, lambda t: conv(t) - datad, nu=nbasis,
                                   tol=0.0, maxit=12, positivity=1, verbosity=0)
    assert np.abs(xb.cpu().numpy() - xb_ref).max() < 1e-8 * np.abs(xb_ref).max()
    assert np.abs(vb.cpu().numpy() - vb_ref).max() < 1e-8 * np.abs(vb_ref).max()


def test_plain_cg_and_psfhat_producer():
    """pcg.py:12-50 (cg) against the oracle; gridder.py:712-714 (psfhat producer) against the
    oracle's scipy version and through a convolution with a centred delta PSF (= identity)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.opt.pcg import cg
    from pfb_clean_amd.operators.hessian import HessianPsf
    from pfb_clean_amd.operators.fft import psfhat_from_psf
    from pfb_clean_amd.operators.psf import psf_convolve_slice
    rng = np.random.default_rng(21)
    nb, nx, ny = 1, 64, 48
    psfhat = _psd_psfhat(rng, nb, 2 * nx, 2 * ny)
    b = rng.standard_normal((nx, ny))
    xpad, xhat, xout = ofc.make_scratch(psfhat[0], 2 * ny, b.shape, np.float64)

    def oA(v):
        return ofc._hessian_psf_slice(xpad, xhat, xout, psfhat[0], None, 2 * ny, v, sigmainv=0.05)
    ref = osv.cg(oA, b, None, tol=1e-12, maxit=25)
    A = HessianPsf(psfhat[0], nx, ny, 2 * ny, sigmainv=0.05)
    x = cg(A, b, None, tol=1e-12, maxit=25, verbosity=0)
    assert np.abs(x - ref).max() < 1e-9 * np.abs(ref).max()
    psf = rng.standard_normal((2, 32, 40))
    assert np.abs(psfhat_from_psf(psf) - ofc.psfhat_from_psf(psf)).max() < 1e-12
    delta = np.zeros((2 * nx, 2 * ny))
    delta[nx, ny] = 1.0                                   # centred unit PSF -> identity
    y = psf_convolve_slice(None, None, None, psfhat_from_psf(delta), 2 * ny, b)
    assert np.abs(y - b).max() < 1e-13


def _pd_rank(rank, world, port, q):
    """One rank of the band-sharded backward step through the real kernels: gloo process
    group (two processes share the single GPU of the test box; RCCL needs one GPU per
    rank), CUDA tensors, pfb_dual_bandsum -> all_reduce -> pfb_dual_apply."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pfb_clean_amd.dist import shard_bands
        from pfb_clean_amd.operators.psf import psf_convolve_cube
        from pfb_clean_amd.operators.psi import Psi
        from pfb_clean_amd.opt.primal_dual import primal_dual_optimised
        here = os.path.dirname(os.path.abspath(__file__))
        g = np.load(os.path.join(here, 'golden', 'pd.npz'))
        psfhat, Q, data = g['psfhat'], int(g['Q']), g['data']
        nband, P, _ = psfhat.shape
        nx = ny = P // 2
        bases = [str(s) for s in g['bases']]
        nbasis = len(bases)
        band0, nb = shard_bands(nband, rank, world)
        sl = slice(band0, band0 + nb)
        dev = torch.device('cuda')
        ph = torch.from_numpy(psfhat[sl]).to(dev)
        dd = torch.from_numpy(data[sl]).to(dev)
        psi = Psi(nb, nx, ny, bases, int(g['nlevel']), 1)
        errs = []
        for tag, pos, maxit in (('pos1_it10', 1, 10), ('pos0_it4', 0, 4), ('pos2_it6', 2, 6)):
            x = torch.zeros((nb, nx, ny), dtype=torch.float64, device=dev)
            v = torch.zeros((nb, nbasis, psi.Nymax, psi.Nxmax), dtype=torch.float64, device=dev)
            x, v = primal_dual_optimised(x, v, float(g['lam']), psi.hdot, psi.dot, float(g['hessnorm']),
                                         None, torch.ones_like(v[0]), None,
                                         lambda t: psf_convolve_cube(None, None, None, ph, Q, t) - dd,
                                         nu=nbasis, tol=0.0, maxit=maxit, positivity=pos, verbosity=0,
                                         group=True)
            errs.append(float(np.abs(x.cpu().numpy() - g[tag + '_x'][sl]).max() / np.abs(g[tag + '_x']).max()))
            errs.append(float(np.abs(v.cpu().numpy() - g[tag + '_v'][sl]).max() / np.abs(g[tag + '_v']).max()))
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


def test_band_sharded_primal_dual_two_ranks_one_gpu():
    """backward step with the bands split over ranks reproduces the REFERENCE's
    golden trajectories (tests/golden/pd.npz, all three positivity modes)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pd_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for rank, errs in sorted(q.get(timeout=5) for _ in range(2)):
        assert max(errs) < 1e-9, (rank, errs)


@pytest.mark.parametrize('tag', ['dirty', 'resid'])
def test_hessian_psf_slice_class_and_pcg_dist(tag):
    """SURVEY 8a rows a5 / a8 (hessian.py:161-251, pcg.py:363-420) against the REFERENCE's
    outputs in tests/golden/dist.npz, fp64, GPU resident."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.operators.hessian import hessian_psf_slice
    from pfb_clean_amd.opt.pcg import pcg_dist

    class _Var:
        def __init__(self, a):
            self.values, self.dtype, self.shape = a, a.dtype, a.shape

    class _DS(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    def dist_dataset(g, tag, wrap):
        extra = {}
        if tag == 'resid':
            extra = {'RESIDUAL': _Var(wrap(g['resid_in'])), 'MODEL': _Var(wrap(g['model_in']))}
        return _DS(DIRTY=_Var(wrap(g['dirty'])), PSFHAT=_Var(wrap(g['psfhat'])), PSF=_Var(wrap(g['psf'])),
                   BEAM=_Var(wrap(g['beam'])), WSUM=_Var(np.array([float(g['wsumb'])])), bandid=3, **extra)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dist.npz'))
    dev = torch.device('cuda')
    A = hessian_psf_slice(dist_dataset(g, tag, wrap=lambda a: torch.from_numpy(np.array(a)).to(dev)),
                          2, 10, 1, float(g['sigmainv']), 1.0, False, 1e-7, True)
    A.set_wsum(float(g['wsum']))
    probe = torch.from_numpy(g[f'{tag}_probe']).to(dev)
    assert np.abs(A(probe).cpu().numpy() - g[f'{tag}_Ax']).max() < 1e-11 * np.abs(g[f'{tag}_Ax']).max()
    assert torch.equal(A.model.cpu(), torch.from_numpy(g[f'{tag}_model']))
    assert torch.equal(A.residual.cpu(), torch.from_numpy(g[f'{tag}_residual']))
    assert tuple(A.dual.shape) == tuple(g[f'{tag}_dual_shape'])
    for name, (maxit, minit, tol) in (('a', (30, 5, 1e-6)), ('b', (8, 8, 0.0))):
        x = pcg_dist(A, maxit, minit, tol, float(g['sigmainv']))
        ref = g[f'{tag}_x_{name}']
        assert np.abs(x.cpu().numpy() - ref).max() < 1e-8 * np.abs(ref).max()
    with pytest.raises(NotImplementedError):
        A.compute_residual(probe)


@pytest.mark.parametrize('apparent', [False, True])
def test_dds2cubes_device_loader(apparent):
    """device-resident cube assembly (misc.py:664-739) against the oracle restatement
    (the reference's own outputs: test_dds2cubes_against_reference_vectors)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.utils.misc import dds2cubes
    from test_oracle_golden import make_dds
    ref = osv.dds2cubes(make_dds(np.random.default_rng(5)), 3, apparent=apparent)
    dev = torch.device('cuda')
    got = dds2cubes(make_dds(np.random.default_rng(5), wrap=lambda a: torch.from_numpy(a).to(dev)), 3,
                    apparent=apparent)
    for r, g_ in zip(ref, got):
        assert g_.is_cuda
        assert np.abs(g_.cpu().numpy() - r).max() <= 1e-13 * max(1.0, np.abs(r).max())
    got = dds2cubes(make_dds(np.random.default_rng(5), with_resid=False, with_dual=False), 3)
    assert got[2] is None and got[7] is None and got[0].is_cuda       # numpy datasets in, tensors out


@pytest.mark.parametrize('kind', ['tensor', 'numpy'])
def test_cg_dct_nested_dict(kind):
    """cg_dct on the GPU against the reference's iterates (tests/golden/dct.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.opt.pcg import cg_dct
    from pfb_clean_amd.operators.hessian import HessianPsf
    from test_oracle_golden import dct_problem
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dct.npz'))
    dev = torch.device('cuda')
    wrap = (lambda a: torch.from_numpy(np.array(a)).to(dev)) if kind == 'tensor' else (lambda a: np.array(a))
    shapes, keys, b, _ = dct_problem(g, wrap)
    ops = {f: {i: HessianPsf(torch.from_numpy(g[f'{f}_{i}_psfhat'][None]).to(dev), shapes[f][0], shapes[f][1],
                             2 * shapes[f][1], sigmainv=float(g['sigmainv'])) for i in keys} for f in shapes}

    def A(v):
        return {f: {i: ops[f][i](v[f][i]) for i in v[f]} for f in v}
    for tag, (tol, maxit) in (('it6', (0.0, 6)), ('tol', (1e-3, 200))):
        x0 = dct_problem(g, wrap)[3]
        xs, rs = cg_dct(A, b, x0, tol=tol, maxit=maxit, verbosity=0)
        assert xs is x0
        for f in shapes:
            for i in keys:
                xv = xs[f][i].cpu().numpy() if kind == 'tensor' else xs[f][i]
                rv = rs[f][i].cpu().numpy() if kind == 'tensor' else rs[f][i]
                # 'tol' runs ~100 unpreconditioned CG steps on an ill-conditioned system: rounding
                # differences between the two FFTs are amplified to ~1e-8 relative
                rtol = 1e-10 if tag == 'it6' else 1e-6
                assert np.abs(xv - g[f'{tag}_{f}_{i}_x']).max() < rtol * np.abs(g[f'{tag}_{f}_{i}_x']).max()
                bmax = np.abs(g[f'{f}_{i}_b']).max()          # the recursive residual has shrunk 1000-fold
                assert np.abs(rv - g[f'{tag}_{f}_{i}_r']).max() < (1e-10 if tag == 'it6' else 1e-5) * bmax


@pytest.mark.parametrize('rdt', [np.float64, np.float32])
@pytest.mark.parametrize('shape', [(2, 128, 128), (1, 96, 80), (2, 60, 100), (1, 45, 66), (1, 512, 2048),
                                   # power-of-two: the fast path's row / column kernels (largest column / row lines)
                                   (2, 2048, 1024), (1, 16384, 256), (1, 128, 16384),
                                   # lines beyond the LDS, not powers of two: global-memory Stockham passes
                                   (1, 12000, 96), (1, 96, 12000), (1, 14400, 128)])
def test_native_psfhat_producer(shape, rdt):
    """pfb_psfconv_set_psf (gridder.py:712-714: r2c(ifftshift(psf))) against the oracle, pow2 and
    mixed-radix grids, odd nx_psf; and a plan built straight from the PSF convolves like one built
    from the oracle's psfhat."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.operators.fft import psfhat_from_psf
    from pfb_clean_amd.operators.psf import PsfConvPlan
    rng = np.random.default_rng(shape[1] + shape[2])
    psf = rng.standard_normal(shape).astype(rdt)
    ref = ofc.psfhat_from_psf(psf.astype(np.float64))
    got = psfhat_from_psf(psf)
    tol = 1e-12 if rdt == np.float64 else 2e-6
    assert got.shape == ref.shape and got.dtype == (np.complex128 if rdt == np.float64 else np.complex64)
    assert np.abs(got - ref).max() < tol * np.abs(ref).max()
    assert np.abs(psfhat_from_psf(psf[0]) - ref[0]).max() < tol * np.abs(ref).max()       # 2-D input
    nb, P, Q = shape
    nx, ny = P // 2, Q // 2
    dev = torch.device('cuda')
    plan, ph = PsfConvPlan.from_psf(torch.from_numpy(psf).to(dev), nx, ny, want_psfhat=True)
    assert np.abs(ph.cpu().numpy() - ref).max() < tol * np.abs(ref).max()
    x = rng.standard_normal((nb, nx, ny)).astype(rdt)
    xpad, xhat, xout = ofc.make_scratch(ref, Q, x.shape, np.float64)
    want = ofc.psf_convolve_cube(xpad, xhat, xout, ref, Q, x.astype(np.float64))
    y = plan.apply(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.abs(y - want).max() < (1e-12 if rdt == np.float64 else 1e-5) * np.abs(want).max()
    plan.close()


def test_psfhat_producer_falls_back_for_odd_last_axis():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.operators.fft import psfhat_from_psf
    psf = np.random.default_rng(2).standard_normal((1, 30, 33))
    assert np.abs(psfhat_from_psf(psf) - ofc.psfhat_from_psf(psf)).max() < 1e-11
    # an even grid with a prime factor above 13 (34 = 2 x 17): the kernels answer PFB_ERR_UNSUPPORTED and the producer
    # degrades to torch.fft like the reference's r2c, which takes any size
    psf = np.random.default_rng(3).standard_normal((1, 34, 34))
    assert np.abs(psfhat_from_psf(psf) - ofc.psfhat_from_psf(psf)).max() < 1e-11


def _pcg_rank(rank, world, port, q):
    """One rank of the band-sharded cube PCG through pfb_pcg_solve + AllReduceHook (gloo group, two
    processes on the single test GPU; uneven shard 2 + 1 bands of the golden cube)."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pfb_clean_amd.dist import shard_bands
        from pfb_clean_amd.operators.hessian import HessianPsf
        from pfb_clean_amd.opt.pcg import pcg_fused
        here = os.path.dirname(os.path.abspath(__file__))
        g = np.load(os.path.join(here, 'golden', 'pcg.npz'))
        psfhat, b, beam = g['psfhat'], g['b'], g['beam']
        sigmainv, Q = float(g['sigmainv']), int(g['Q'])
        nband, nx, ny = b.shape
        band0, nb = shard_bands(nband, rank, world)
        sl = slice(band0, band0 + nb)
        dev = torch.device('cuda')
        A = HessianPsf(torch.from_numpy(psfhat[sl]).to(dev), nx, ny, Q, beam=torch.from_numpy(beam[sl]).to(dev),
                       sigmainv=sigmainv, wsum=1.0)
        bt = torch.from_numpy((beam * b)[sl]).to(dev)
        res = {}
        for tag, kw in (('fixed', dict(tol=0.0, maxit=10, minit=10)), ('stop', dict(tol=2e-2, maxit=60, minit=3))):
            x, _, r = pcg_fused(A, bt, None, mdiv=sigmainv, distributed=True, **kw)
            res[tag] = (x.cpu().numpy(), r.iters)
        # band-sharded power method (reference power_method_dist, power_method.py:70-116): the three inner
        # products summed over the ranks, every rank the same beta
        from pfb_clean_amd.opt.power_method import power_method
        b0 = np.random.default_rng(77).standard_normal((nband, nx, ny))
        beta, vec = power_method(A, (nb, nx, ny), b0=torch.from_numpy(b0[sl]).to(dev), tol=1e-9, maxit=40, verbosity=0,
                                 group=True)
        res['pm'] = (beta, vec.cpu().numpy())
        q.put((rank, band0, nb, res))
    finally:
        dist.destroy_process_group()


def test_band_sharded_cube_pcg_two_ranks_one_gpu():
    """cube PCG with the bands split over two ranks equals the oracle's single-process
    solve -- once with k < minit throughout (ONE merged all-reduce per iteration) and once with the
    stopping rule live after minit (two reduction points, host looks every iteration)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pcg.npz'))
    psfhat, b, beam = g['psfhat'], g['b'], g['beam']
    sigmainv, Q = float(g['sigmainv']), int(g['Q'])
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, b.shape, np.float64)

    def oA(v):
        return ofc.hessian_psf_cube(xpad, xhat, xout, beam, psfhat, Q, v, sigmainv=sigmainv, wsum=1.0)
    ref = {}
    for tag, kw in (('fixed', dict(tol=0.0, maxit=10, minit=10)), ('stop', dict(tol=2e-2, maxit=60, minit=3))):
        tr = osv.PCGTrace()
        ref[tag] = (osv.pcg(oA, beam * b, None, M=lambda v: v / sigmainv, trace=tr, **kw), len(tr.eps))
    b0 = np.random.default_rng(77).standard_normal(b.shape)
    beta_ref, vec_ref = osv.power_method(oA, b.shape, b0=b0, tol=1e-9, maxit=40, verbosity=0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pcg_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, band0, nb, res in out:
        for tag in ('fixed', 'stop'):
            x, iters = res[tag]
            xr, kref = ref[tag]
            assert iters == kref, (tag, iters, kref)
            assert np.abs(x - xr[band0:band0 + nb]).max() < 1e-9 * np.abs(xr).max(), (rank, tag)
        beta, vec = res['pm']
        assert abs(beta - beta_ref) < 1e-10 * abs(beta_ref), (rank, beta, beta_ref)
        assert np.abs(vec - vec_ref[band0:band0 + nb]).max() < 1e-9 * np.abs(vec_ref).max(), rank


@pytest.mark.parametrize('rdt', [np.float64, np.float32])
def test_clark_minor_cycle(rdt):
    """Clark CLEAN (deconv/clark.py) with the sub-minor loop in one resident workgroup and the
    fused convolution, against the REFERENCE's outputs (tests/golden/clark.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.deconv.clark import clark
    from pfb_clean_amd import _lib, _dev
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'clark.npz'))
    cdt = np.complex128 if rdt == np.float64 else np.complex64
    ID, PSF, PSFHAT, wsums = g['ID'].astype(rdt), g['PSF'].astype(rdt), g['PSFHAT'].astype(cdt), g['wsums'].astype(rdt)
    dev = torch.device('cuda')
    # the sub-minor loop alone through the C-ABI
    lib = _lib.load()
    Ip, Iq = g['sub_Ip'], g['sub_Iq']
    A = torch.from_numpy(np.ascontiguousarray(ID[:, Ip, Iq])).to(dev)
    model = torch.zeros(ID.shape, dtype=A.dtype, device=dev)
    it = torch.zeros(1, dtype=torch.int32, device=dev)
    psf_d, w_d = torch.from_numpy(PSF).to(dev), torch.from_numpy(wsums).to(dev)     # keep alive across the call
    ip_d, iq_d = torch.from_numpy(Ip.astype(np.int32)).to(dev), torch.from_numpy(Iq.astype(np.int32)).to(dev)
    _lib.check(lib.pfb_clark_subminor(_dev.code(A.dtype), _dev.ptr(A), A.shape[1], ID.shape[0],
                                      _dev.ptr(psf_d), PSF.shape[1], PSF.shape[2], _dev.ptr(ip_d), _dev.ptr(iq_d),
                                      _dev.ptr(model), ID.shape[1], ID.shape[2], _dev.ptr(w_d),
                                      0.1, float(g['sub_th']), 25, _dev.ptr(it), _dev.stream()))
    torch.cuda.synchronize()
    tol = 1e-11 if rdt == np.float64 else 2e-4
    assert it.item() > 3
    assert np.abs(model.cpu().numpy() - g['sub_model']).max() < tol * np.abs(g['sub_model']).max()
    # full minor cycle, numpy in -> numpy out and tensors in -> tensors out
    for tag, kw in (('a', dict(gamma=0.1, pf=0.05, maxit=6, subpf=0.5, submaxit=40)),
                    ('b', dict(gamma=0.05, pf=0.3, maxit=50, subpf=0.7, submaxit=1000, threshold=0.0))):
        m, status = clark(ID.copy(), PSF, PSFHAT, wsums, verbosity=0, **kw)
        ref = g[f'clark_{tag}_model']
        assert status == int(g[f'clark_{tag}_status']) and isinstance(m, np.ndarray)
        assert np.abs(m - ref).max() < tol * np.abs(ref).max(), tag
    mt, _ = clark(torch.from_numpy(ID).to(dev), torch.from_numpy(PSF).to(dev), torch.from_numpy(PSFHAT).to(dev),
                  torch.from_numpy(wsums).to(dev), verbosity=0, gamma=0.1, pf=0.05, maxit=6, subpf=0.5, submaxit=40)
    assert mt.is_cuda and np.abs(mt.cpu().numpy() - g['clark_a_model']).max() < tol * np.abs(g['clark_a_model']).max()


@pytest.mark.parametrize('rdt', [np.float64, np.float32])
def test_freqmul_and_parametrisations(rdt):
    """misc.py:1366-1423 (freqmul, setup_parametrisation) against numpy in both precisions (the
    reference's own outputs: test_freqmul_parametrisation_against_reference_vectors)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.utils.misc import freqmul, setup_parametrisation
    rng = np.random.default_rng(8)
    nband, nx, ny = 5, 37, 41
    A = rng.standard_normal((nband, nband)).astype(rdt)
    x = rng.standard_normal((nband, nx, ny)).astype(rdt)
    tol = 1e-13 if rdt == np.float64 else 2e-6
    ref = np.einsum('kl,lij->kij', A.astype(np.float64), x.astype(np.float64))
    assert np.abs(freqmul(A, x) - ref).max() < tol * np.abs(ref).max()
    xt = torch.from_numpy(x).cuda()
    assert np.abs(freqmul(torch.from_numpy(A).cuda(), xt).cpu().numpy() - ref).max() < tol * np.abs(ref).max()
    freq = np.linspace(1.0e9, 1.7e9, nband)
    nu = freq / freq.mean()
    K = 0.8 ** 2 * np.exp(-(nu[:, None] - nu[None, :]) ** 2 / (2 * 0.5 ** 2))
    L = np.linalg.cholesky(K + 1e-10 * np.eye(nband))
    mul = lambda M, v: np.einsum('kl,lij->kij', M, v.astype(np.float64))
    for mode in ('id', 'exp'):
        func, finv, dfunc, dhfunc = setup_parametrisation(mode=mode, sigma=0.8, freq=freq, lscale=0.5)
        s0 = 0.3 * x
        v = rng.standard_normal(x.shape).astype(rdt)
        if mode == 'id':
            want = (mul(L, s0), mul(L, v), mul(L.T, v))
        else:
            e = np.exp(mul(L, s0))
            want = (e, e * mul(L, v), mul(L.T, v * e))
        got = (func(s0), dfunc(s0, v), dhfunc(s0, v))
        for g_, w_ in zip(got, want):
            assert np.abs(g_ - w_).max() < 50 * tol * np.abs(w_).max()
        y = want[0]                                 # finv as the reference writes it (for 'exp' it is NOT
        Li = np.linalg.solve(L, np.eye(nband))      # the inverse of func: log(max(|L^-1 y|, minval)))
        wf = mul(Li, y) if mode == 'id' else np.log(np.maximum(np.abs(mul(Li, y)), 1e-5))
        gf = finv(y.astype(rdt))
        assert np.abs(gf - wf).max() < (1e-7 if rdt == np.float64 else 5e-2) * max(1.0, np.abs(wf).max())
        # adjointness <dfunc v, u> = <v, dhfunc u>
        u = rng.standard_normal(x.shape).astype(rdt)
        lhs = np.vdot(dfunc(s0, v).astype(np.float64), u.astype(np.float64))
        rhs = np.vdot(v.astype(np.float64), dhfunc(s0, u).astype(np.float64))
        assert abs(lhs - rhs) < 200 * tol * (np.linalg.norm(dfunc(s0, v)) * np.linalg.norm(u))


@pytest.mark.parametrize('rdt', [np.float64, np.float32])
def test_hogbom(rdt):
    """deconv/hogbom.py on the GPU (pfb_hogbom: loop state on the device) against the REFERENCE's outputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.deconv.hogbom import hogbom
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'clark.npz'))
    ID, PSF = g['ID'].astype(rdt), g['PSF'].astype(rdt)
    tol = 1e-11 if rdt == np.float64 else 5e-4
    for tag, kw in (('a', dict(gamma=0.1, pf=0.1, maxit=10000)), ('b', dict(gamma=0.2, pf=0.01, maxit=37))):
        m, status = hogbom(ID.copy(), PSF, verbosity=0, **kw)
        ref = g[f'hogbom_{tag}_model']
        assert status == int(g[f'hogbom_{tag}_status'])
        assert np.abs(m - ref).max() < tol * np.abs(ref).max(), tag
    mt, _ = hogbom(torch.from_numpy(ID).cuda(), torch.from_numpy(PSF).cuda(), verbosity=0, gamma=0.2, pf=0.01, maxit=37)
    assert mt.is_cuda and np.abs(mt.cpu().numpy() - g['hogbom_b_model']).max() < tol * np.abs(g['hogbom_b_model']).max()


def test_plan_from_psf_for_embedded_sizes():
    """PsfConvPlan.from_psf on a size that is not a fast-path size: native PSFHAT, then the re-gridded plan."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.operators.psf import PsfConvPlan
    rng = np.random.default_rng(4)
    nb, nx, ny = 2, 100, 120
    P, Q = 2 * nx, 2 * ny
    psf = rng.standard_normal((nb, P, Q))
    ref_hat = ofc.psfhat_from_psf(psf)
    plan, ph = PsfConvPlan.from_psf(torch.from_numpy(psf).cuda(), nx, ny, want_psfhat=True)
    assert plan.embed == (128, 128) and plan.fast_path
    assert np.abs(ph.cpu().numpy() - ref_hat).max() < 1e-12 * np.abs(ref_hat).max()
    x = rng.standard_normal((nb, nx, ny))
    xpad, xhat, xout = ofc.make_scratch(ref_hat, Q, x.shape, np.float64)
    want = ofc.psf_convolve_cube(xpad, xhat, xout, ref_hat, Q, x)
    got = plan.apply(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()


def test_new_entry_points_reject_bad_arguments():
    """Error behaviour of the round's later C-ABI entry points: status codes + message, nothing launched."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd import _lib, _dev
    from pfb_clean_amd._lib import PfbHipError
    from pfb_clean_amd.deconv.hogbom import hogbom
    from pfb_clean_amd.deconv.clark import clark
    lib = _lib.load()
    dev = torch.device('cuda')
    a = torch.zeros((2, 16), dtype=torch.float64, device=dev)
    psf_small = torch.ones((2, 8, 8), dtype=torch.float64, device=dev)       # cannot cover a 16 x 16 image
    idx = torch.zeros(16, dtype=torch.int32, device=dev)
    model = torch.zeros((2, 16, 16), dtype=torch.float64, device=dev)
    w = torch.full((2,), 0.5, dtype=torch.float64, device=dev)
    rc = lib.pfb_clark_subminor(1, _dev.ptr(a), 16, 2, _dev.ptr(psf_small), 8, 8, _dev.ptr(idx), _dev.ptr(idx),
                                _dev.ptr(model), 16, 16, _dev.ptr(w), 0.1, 0.0, 10, None, _dev.stream())
    assert rc == _lib.PFB_ERR_UNSUPPORTED and b'must cover' in lib.pfb_last_error()
    rc = lib.pfb_clark_subminor(1, None, 16, 2, _dev.ptr(psf_small), 8, 8, _dev.ptr(idx), _dev.ptr(idx),
                                _dev.ptr(model), 16, 16, _dev.ptr(w), 0.1, 0.0, 10, None, _dev.stream())
    assert rc == -1
    with pytest.raises(PfbHipError):
        hogbom(np.ones((2, 16, 16)), np.ones((2, 8, 8)), verbosity=0)
    with pytest.raises(ValueError):
        hogbom(np.ones((2, 16, 16)), np.zeros((2, 32, 32)), verbosity=0)     # PSF peaks must be positive
    with pytest.raises(AssertionError):
        clark(np.ones((2, 16, 16)), np.ones((2, 32, 32)), np.ones((2, 32, 17), complex), np.array([0.7, 0.7]), verbosity=0)
    # regrid: target grid too small for the image, and a grid length with a prime factor > 13 (every buffer is
    # sized for the call, so that nothing is out of bounds should a check ever go missing)
    ph = torch.zeros((1, 32, 17), dtype=torch.complex128, device=dev)
    out = torch.zeros((1, 34, 18), dtype=torch.complex128, device=dev)
    assert lib.pfb_psfhat_regrid(1, _dev.ptr(ph), 1, 20, 20, 32, 32, 32, 32, _dev.ptr(out), _dev.stream()) == -1
    assert lib.pfb_psfhat_regrid(1, _dev.ptr(ph), 1, 8, 8, 32, 32, 34, 34, _dev.ptr(out), _dev.stream()) \
        == _lib.PFB_ERR_UNSUPPORTED
    psf17 = torch.zeros((1, 34, 34), dtype=torch.float64, device=dev)
    assert lib.pfb_psfhat_from_psf(1, _dev.ptr(psf17), 1, 34, 34, _dev.ptr(out), _dev.stream()) == _lib.PFB_ERR_UNSUPPORTED
    assert lib.pfb_psfhat_from_psf(1, _dev.ptr(psf17), 1, 34, 33, _dev.ptr(out), _dev.stream()) == _lib.PFB_ERR_UNSUPPORTED
    x = torch.zeros((2, 4, 4), dtype=torch.float64, device=dev)
    A = torch.eye(2, dtype=torch.float64, device=dev)
    assert lib.pfb_freqmul(1, _dev.ptr(A), _dev.ptr(x), _dev.ptr(x), 2, 16, None, None, _dev.stream()) == -1   # aliasing


# ------------------------------------------------ pfb/utils/misc.py against tests/golden/misc.npz
def _misc():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'misc.npz'))


def test_norm_diff_against_reference_vectors():
    """misc.py:1316-1351: the device reduction against the values the reference's loop gives."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.utils.misc import norm_diff
    g = _misc()
    for tag, rtol in (('f64', 1e-13), ('f32', 1e-6)):
        x, xp = g[f'nd_{tag}_x'], g[f'nd_{tag}_xp']
        for a, b, key in ((x, xp, '3d'), (x[1], xp[1], '2d')):
            want = float(g[f'nd_{tag}_{key}'])
            assert abs(norm_diff(a, b) - want) <= rtol * want
            assert abs(norm_diff(torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b.copy()).cuda()) - want) <= rtol * want
    assert norm_diff(np.zeros((4, 4)), np.zeros((4, 4))) == 0.0


@pytest.mark.parametrize('kind', ['numpy', 'tensor'])
def test_l1reweight_against_reference_vectors(kind):
    """misc.py:1070-1080 with the device Psi.dot as psiH."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.utils.misc import l1reweight_func
    from pfb_clean_amd.operators.psi import Psi
    g = _misc()
    nband, nx, ny, nlevel = (int(v) for v in g['rw_meta'])
    psi = Psi(nband, nx, ny, [str(b) for b in g['rw_bases']], nlevel)
    wrap = (lambda a: torch.from_numpy(np.array(a)).cuda()) if kind == 'tensor' else (lambda a: np.array(a))
    outvar = wrap(np.zeros((nband, psi.nbasis, psi.Nymax, psi.Nxmax)))
    for alpha in (4, 2):
        got = l1reweight_func(psi.dot, outvar, 1.5, wrap(g['rw_rms']), wrap(g['rw_model']), alpha=alpha)
        got = got.cpu().numpy() if kind == 'tensor' else got
        assert np.abs(got - g[f'rw_a{alpha}']).max() < 1e-10


@pytest.mark.parametrize('kind', ['numpy', 'tensor'])
def test_dds2cubes_against_reference_vectors(kind):
    """misc.py:664-739: all four call shapes of the fixture (beam-weighted, apparent, no
    residual/psf/dual variables, dual=False)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.utils.misc import dds2cubes
    from test_oracle_golden import misc_dds, CUBE_NAMES, CUBE_CASES
    g = _misc()
    wrap = (lambda a: torch.from_numpy(a).cuda()) if kind == 'tensor' else (lambda a: a)
    for tag, kw, skip in CUBE_CASES:
        got = dds2cubes(misc_dds(g, wrap=wrap, skip=skip), 3, **kw)
        for n, r in zip(CUBE_NAMES, got):
            key = f'cubes_{tag}_{n}'
            if key in g.files:
                assert r.is_cuda and tuple(r.shape) == g[key].shape
                assert np.abs(r.cpu().numpy() - g[key]).max() <= 1e-14 * max(1.0, np.abs(g[key]).max())
            else:
                assert r is None


def test_freqmul_parametrisation_against_reference_vectors():
    """misc.py:1366-1423: pfb_freqmul and the four closures of both parametrisations."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.utils.misc import freqmul, setup_parametrisation
    g = _misc()
    assert np.abs(freqmul(g['fm_A'], g['fm_x']) - g['fm_out']).max() < 1e-13
    for mode in ('id', 'exp'):
        func, finv, dfunc, dhfunc = setup_parametrisation(mode=mode, minval=1e-5, sigma=0.8,
                                                          freq=g['par_freq'], lscale=0.5)
        x0, v = g['par_x0'], g['par_v']
        for got, key, tol in ((func(x0), 'func', 1e-12), (dfunc(x0, v), 'dfunc', 1e-12),
                              (dhfunc(x0, v), 'dhfunc', 1e-12), (finv(func(x0)), 'finv', 1e-7)):
            want = g[f'par_{mode}_{key}']
            assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), (mode, key)
        xt, vt = torch.from_numpy(x0).cuda(), torch.from_numpy(v).cuda()
        got = dfunc(xt, vt)
        assert got.is_cuda and np.abs(got.cpu().numpy() - g[f'par_{mode}_dfunc']).max() < 1e-12


@pytest.mark.parametrize('kind', ['numpy', 'tensor'])
def test_plain_cg_against_reference_vectors(kind):
    """pcg.py:12-50 (cg) on the GPU against the reference's outputs (tests/golden/cg.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pfb_clean_amd.opt.pcg import cg
    from pfb_clean_amd.operators.hessian import HessianPsf
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cg.npz'))
    b, Q, sigmainv = g['b'], int(g['Q']), float(g['sigmainv'])
    nb, nx, ny = b.shape
    A = HessianPsf(g['psfhat'], nx, ny, Q, sigmainv=sigmainv)
    wrap = (lambda a: torch.from_numpy(np.array(a)).cuda()) if kind == 'tensor' else (lambda a: np.array(a))
    back = (lambda t: t.cpu().numpy()) if kind == 'tensor' else (lambda a: a)
    for k in (1, 4, 12):
        x = back(cg(A, wrap(b), None, tol=0.0, maxit=k, verbosity=0))
        assert np.abs(x - g[f'k{k}']).max() < 1e-9 * np.abs(g[f'k{k}']).max(), k
    x = back(cg(A, wrap(b), None, tol=1e-6, maxit=500, verbosity=0))
    assert np.abs(x - g['tol']).max() < 1e-6 * np.abs(g['tol']).max()
    x0 = wrap(g['x0'])
    x = back(cg(A, wrap(b), x0, tol=0.0, maxit=5, verbosity=0))
    assert np.abs(x - g['warm_k5']).max() < 1e-9 * np.abs(g['warm_k5']).max()
    assert np.array_equal(back(x0), g['x0'])                        # x0 is copied, not updated (pcg.py:23)


def test_bench_two_ranks_rehearsal():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU, band shard,
    all-reduce hook, max-over-ranks timing, one JSON line from rank 0) -- rehearsed with two ranks sharing
    the test GPU over gloo (RCCL refuses two ranks on one device; PFB_DIST_BACKEND is a bench-only switch)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PFB_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(root, 'bench.py'), '--gpus', '2', '--size', '1024', '--bands', '4', '--steps', '6',
           '--warmup', '2', '--no-cpu']
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['matvecs'] == 7
    assert d['config']['bands_per_gpu'] == 2 and d['value'] > 0 and d['scaling'] == 'strong'
    assert d['roofline'] is not None and d['cpu_baseline'] is None
