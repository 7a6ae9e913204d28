#!/usr/bin/env python
"""
bench.py -- PCG matvecs/sec on the BASELINE workload: 4096 x 4096 x 8-band cube PCG
(fluxmop semantics: ONE system over all bands, global dot products), fp32, synthetic
dirty/PSF cubes, bands sharded over N GPUs with one small RCCL all-reduce per
reduction point.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one PCG iteration (one A^H A apply on the whole cube + the fused vector
updates and reductions).  The timed region is ONE pfb_pcg_solve of exactly K iterations
(tol = 0, minit = maxit = K => K + 1 matvecs), all inputs resident in HBM, bracketed by
barrier + synchronize; time = max over ranks; value = (K + 1) cube-matvecs / time.
Total work is fixed as N grows => "scaling": "strong".

Extra objects on the JSON line:
  roofline     the FFT-convolution kernel group (row-fwd, column, row-inv = one band
               matvec), algorithmic bytes B_alg = s (2 nx ny + 2 nx_psf (ny_psf/2+1)) per
               band-matvec (SURVEY 8d / BASELINE.md 4) over its HIP-event-measured
               duration inside the timed solve; per-stage times alongside.
  cpu_baseline the numpy/scipy.fft oracle ("port" of the reference's CPU path; ducc0 is
               not installable) timed on this host on ONE band of the same cube for a
               bounded number of iterations, converted to cube-matvecs/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md; 6.29 TB/s measured copy)


def synth_band(band, nband, nx, ny, dtype, device):
    """psfhat of one band: non-negative Poisson uv weights under a Gaussian
    taper, normalised so that sum_band psf peaks at 1 (spotless.py:148-149).  Generated
    on the device (setup, not timed) with a per-band seed 420 + band."""
    P, Q = 2 * nx, 2 * ny
    g = torch.Generator(device=device)
    g.manual_seed(420 + band)
    u = torch.fft.fftfreq(P, device=device, dtype=torch.float32)[:, None]
    v = torch.fft.rfftfreq(Q, device=device, dtype=torch.float32)[None, :]
    lam = 4.0 * torch.exp(-(u * u + v * v) / (2 * 0.12 ** 2))
    W = torch.poisson(lam, generator=g).to(torch.float64)
    # psf peak = psf[0,0] = sum over the FULL plane / (P Q); Hermitian completion doubles
    # the interior columns of the half plane
    full = W.sum() * 2 - W[:, 0].sum() - W[:, -1].sum()
    peak = full / (P * Q)
    cdt = torch.complex64 if dtype == torch.float32 else torch.complex128
    return (W / (nband * peak)).to(cdt)


def synth_model(band, nx, ny, dtype, device):
    """25 elliptical Gaussians + 10 unit point sources, spectral index -0.7 +- 0.1
    (mirrors tests/test_spotless.py:88-107, tests/test_klean.py:71-78)."""
    rng = np.random.default_rng(420)
    yy = torch.arange(ny, device=device, dtype=torch.float32)[None, :]
    xx = torch.arange(nx, device=device, dtype=torch.float32)[:, None]
    nu = 1.0 + 0.1 * band
    model = torch.zeros((nx, ny), dtype=torch.float32, device=device)
    for _ in range(25):
        cx = rng.uniform(0.15 * nx, 0.85 * nx)
        cy = rng.uniform(0.15 * ny, 0.85 * ny)
        ex = rng.integers(3, max(4, int(0.02 * nx)))
        ey = rng.integers(3, max(4, int(0.02 * ny)))
        peak = 1 + np.exp(rng.standard_normal())
        alpha = -0.7 + 0.1 * rng.standard_normal()
        model += float(peak * nu ** alpha) * torch.exp(-((xx - cx) ** 2 / (2 * ex ** 2) + (yy - cy) ** 2 / (2 * ey ** 2)))
    for _ in range(10):
        i, j = rng.integers(0.15 * nx, 0.85 * nx), rng.integers(0.15 * ny, 0.85 * ny)
        model[i, j] += 1.0
    return model.to(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--bands', type=int, default=8)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL in production; PFB_DIST_BACKEND=gloo lets a 1-GPU box rehearse the N > 1 path
        dist.init_process_group(os.environ.get('PFB_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(ndev, 1))
    device = torch.device('cuda', local_rank % max(ndev, 1))

    from pfb_clean_amd.operators.psf import PsfConvPlan
    from pfb_clean_amd.operators.hessian import HessianPsf
    from pfb_clean_amd.opt.pcg import pcg_fused
    from pfb_clean_amd.dist import shard_bands, global_max

    n = args.size
    nband = args.bands
    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    band0, nb = shard_bands(nband, rank, world)
    if nb == 0:
        raise SystemExit("more ranks than bands")
    Q = 2 * n

    # ---------------------------------------------------------------- synthetic inputs
    psfhat = torch.stack([synth_band(band0 + k, nband, n, n, dtype, device) for k in range(nb)])
    plan = PsfConvPlan(psfhat, n, n, Q)
    model = torch.stack([synth_model(band0 + k, n, n, dtype, device) for k in range(nb)])
    b = plan.apply(model).clone()
    gen = torch.Generator(device=device)
    gen.manual_seed(1420 + rank)
    b += 1e-3 * torch.randn(b.shape, dtype=dtype, device=device, generator=gen)
    sigmainv = 1e-3 * global_max(b.abs().max().item(), device)
    A = HessianPsf(plan, n, n, Q, sigmainv=sigmainv)
    del model
    torch.cuda.synchronize()

    def solve(iters, **kw):
        return pcg_fused(A, b, None, mdiv=sigmainv, tol=0.0, maxit=iters, minit=iters,
                         backtrack=True, distributed=world > 1, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ warmup + timing
    if args.warmup > 0:
        solve(args.warmup)
    barrier()
    plan.set_profiling(4)        # every 4th launch group carries the stage events (~6 us each on the stream)
    t0 = time.perf_counter()
    x, _, res = solve(args.steps)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    stage_ms, napply = plan.get_profile()
    plan.set_profiling(False)
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert res.iters == args.steps and res.matvecs == args.steps + 1, (res.iters, res.matvecs)
    matvecs = res.matvecs
    value = matvecs / elapsed

    # ---------------------------------------------------------------------- roofline
    s = 4 if dtype == torch.float32 else 8
    balg_band = s * (2 * n * n + 2 * (2 * n) * (Q // 2 + 1))          # bytes per band-matvec
    roofline = None
    if napply > 0:
        conv_ms = sum(stage_ms) / napply                              # per launch group (nb bands)
        achieved = balg_band * nb / (conv_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(n, nb, args.dtype)
        roofline = {
            "bound": "hbm", "kernel": "fft-convolution (row_fwd + col + row_inv)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc)",
            "traffic_source": traffic_src,
            "hbm_rate_from_traffic_GBs": None if traffic is None else round(traffic / (conv_ms * 1e-3) / 1e9, 1),
            "alg_bytes_per_launch": balg_band * nb, "launches_timed": napply,
            "ms_per_launch": round(conv_ms, 4),
            "stage_ms": {"row_fwd": round(stage_ms[0] / napply, 4),
                         "col": round(stage_ms[1] / napply, 4),
                         "row_inv": round(stage_ms[2] / napply, 4)},
            "conv_share_of_step": round(conv_ms * 1e-3 * matvecs / elapsed, 3),
        }

    # ------------------------------------------------------------------ cpu baseline
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(psfhat[0], b[0], sigmainv, n, nband, args.cpu_seconds, x[0])

    if rank == 0:
        out = {
            "metric": "PCG matvecs/sec (FFT-conv A^H A apply), 4k x 4k x 8-band cube",
            "value": round(value, 3), "unit": "cube-matvecs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{n}x{n}x{nband}-band cube PCG (hessian_psf + Tikhonov, "
                                   f"M = x/sigmainv, tol=0, minit=maxit={args.steps})",
                       "nx_psf": 2 * n, "bands_per_gpu": nb, "parallelism": f"band-shard x{world}",
                       "fast_path": plan.fast_path},
            "band_matvecs_per_s": round(value * nband, 2),
            "matvecs": matvecs, "backtracks": res.backtracks,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(n, nb, dtype):
    """HBM bytes per launch of the convolution kernel group, from the committed rocprofv3
    PMC summary of THIS workload (counters cannot be read from inside the process; they are
    collected in separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py and
    corrected as MI355X_MICROARCH.md prescribes).  None when no summary matches."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*hbm_traffic.json')), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        c = d.get('config', {})
        if c.get('size') == n and c.get('bands') == nb and c.get('dtype') == dtype:
            return d.get('conv_group_hbm_bytes_per_launch'), os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(psfhat_dev, b_dev, sigmainv, n, nband, seconds, x_gpu):
    """Oracle (numpy + scipy.fft, all host cores) on ONE band of the cube: per-band PCG is
    the same arithmetic per band as the cube PCG; cube-matvecs/s = band rate / nband."""
    from oracle import fftconv as ofc, solvers as osv
    # scipy.fft gets SLOWER beyond ~16 workers at this size (tools/cpu_workers_sweep.py on the GPU box:
    # 209 ms per 4096^2 band-matvec with 16 workers, 300 ms with 64, 640 ms with 256), and 16 is
    # also the CPU share of a one-GPU box: use 16 threads and report them
    cores = min(16, os.cpu_count() or 1)
    psfhat = psfhat_dev.cpu().numpy()
    b = b_dev.cpu().numpy()
    Q = 2 * n
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, b.shape, b.dtype)

    def A(v):
        return ofc._hessian_psf_slice(xpad, xhat, xout, psfhat, None, Q, v, nthreads=cores,
                                      sigmainv=b.dtype.type(sigmainv))
    t0 = time.perf_counter()
    A(b)
    t_one = time.perf_counter() - t0
    iters = int(max(2, min(50, seconds / max(t_one * 1.3, 1e-6))))
    t0 = time.perf_counter()
    osv.pcg(A, b, None, M=lambda v: v / b.dtype.type(sigmainv), tol=0.0, maxit=iters, minit=iters,
            verbosity=0)
    dt = time.perf_counter() - t0
    band_rate = (iters + 1) / dt
    import scipy
    return {"value": round(band_rate / nband, 4), "unit": "cube-matvecs/s", "cores": cores,
            "kind": "port",
            "sample": f"oracle pcg (numpy + scipy.fft {scipy.__version__} workers={cores}) on 1 of "
                      f"{nband} bands, {iters} iterations = {iters + 1} matvecs in {dt:.1f} s; "
                      f"cube rate = band rate / {nband}; host has {os.cpu_count()} cores, "
                      f"scipy.fft is fastest at ~16 workers for this size",
            "band_matvecs_per_s": round(band_rate, 3)}


if __name__ == '__main__':
    main()
