#!/usr/bin/env python
"""
bench.py -- the BASELINE workloads on MI355X, one JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--workload pcg|pd]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

--workload pcg (default, the headline metric): PCG matvecs/sec on the 4096 x 4096 x 8-band cube
(fluxmop semantics: ONE system over all bands, global dot products), fp32, synthetic dirty/PSF
cubes, bands sharded over N GPUs with one small RCCL all-reduce per reduction point.
A "step" is one PCG iteration (one A^H A apply on the whole cube + the fused vector updates and
reductions).  A timed region is ONE pfb_pcg_solve of exactly K iterations (tol = 0, minit =
maxit = K => K + 1 matvecs), everything resident in HBM, bracketed by barrier + synchronize,
time = max over ranks.  The region is repeated R times; `value` = (K + 1) cube-matvecs / MEDIAN
time, min / max alongside.  Total work is fixed as N grows => "scaling": "strong".

--workload pd: BASELINE config #4, the backward step of the forward-backward loop: K iterations
of primal_dual_optimised on a 2048 x 2048 x 4-band cube (psi / psi^H with self + db1..db4, 3
levels, dual update, one PSF convolution for the gradient), value = iterations / s.

`python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset) starts the N ranks itself as a
child `torch.distributed.run` before anything touches the GPU and relays rank 0's line;
--gpus that disagrees with WORLD_SIZE is an error.

Extra objects on the JSON line:
  roofline     pcg: the FFT-convolution kernel group (row-fwd, column, row-inv = one band matvec),
               algorithmic bytes B_alg = s (2 nx ny + 2 nx_psf (ny_psf/2+1)) per band-matvec
                over its HIP-event-measured duration inside the timed solves (events
               recorded by the library on the solver's stream); per-stage times alongside.
               pd: the whole iteration's algorithmic bytes over its duration.
  cpu_baseline the numpy/scipy.fft oracle ("port" of the reference's CPU path; ducc0 is not
               installable) on this host: ALL bands of the same cube solved concurrently, one
               process per band (the reference runs bands concurrently too, pcg.py:320-356), a
               bounded number of iterations; the cube rate is measured, not extrapolated.
  parity       max relative difference GPU vs CPU oracle of one band-matvec on the same input.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md; 6.29 TB/s measured copy)

# the sources that define the kernels a PMC traffic summary was taken on: profiles/*hbm_traffic.json carry the hash of
# this set (tools/make_profile_summary.py), and a rate derived from a summary is only reported while it still matches
KERNEL_SRC = {'pcg': ['fftconv_pow2.hip', 'fft_pow2.hpp', 'common.hpp', 'conv_plan.hpp', 'cgvec.hip', 'Makefile'],
              'pd': ['fftconv_pow2.hip', 'fft_pow2.hpp', 'common.hpp', 'conv_plan.hpp', 'wavelet.hip', 'Makefile']}


def kernel_src_hash(workload='pcg'):
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SRC[workload]:
        with open(os.path.join(ROOT, 'pfb_clean_amd', 'csrc', name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each (median reported)')
    ap.add_argument('--workload', default='pcg', choices=['pcg', 'pd'])
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--bands', type=int, default=None)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline / parity leg')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--cpu-cores', type=int, default=0,
                    help='host cores for the cpu_baseline leg (0 = min(usable cores, 64): the fastest setting measured)')
    ap.add_argument('--configs', default='auto', choices=['auto', 'all', 'none'],
                    help="the other BASELINE configs (C1, C2, C4, C5 shard) timed in the same run and reported under "
                         "'configs' on the JSON line; auto = all for the default one-GPU headline run")
    ap.add_argument('--force-dist', action='store_true',
                    help='run the band-sharded code path (all-reduce hook) even at world size 1')
    ap.add_argument('--cpu-worker', nargs=5, metavar=('DIR', 'BAND', 'THREADS', 'ITERS', 'SIGMAINV'),
                    help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def maybe_launch(args):
    """--gpus N > 1 without a launcher: start N ranks as a CHILD process (nothing in this process has
    touched the GPU yet -- never exec / re-launch after that) and exit with its code."""
    if 'WORLD_SIZE' in os.environ:
        world = int(os.environ['WORLD_SIZE'])
        if args.gpus != world:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.exit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------- synthetic inputs
def synth_band(band, nband, nx, ny, dtype, device):
    """psfhat of one band: non-negative Poisson uv weights under a Gaussian
    taper, normalised so that sum_band psf peaks at 1 (spotless.py:148-149).  Generated
    on the device (setup, not timed) with a per-band seed 420 + band."""
    import torch
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
    import numpy as np
    import torch
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


class _stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; rank 0's stdout must carry
    exactly one JSON line, so file descriptor 1 points at stderr while the process group warms up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _median(v):
    s = sorted(v)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


# ---------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(*args.cpu_worker)
    maybe_launch(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_pg = world > 1 or args.force_dist
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(ndev, 1))
    device = torch.device('cuda', local_rank % max(ndev, 1))
    if use_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        with _stdout_to_stderr():
            # RCCL in production; PFB_DIST_BACKEND=gloo lets a 1-GPU box rehearse the N > 1 path
            dist.init_process_group(os.environ.get('PFB_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
            t = torch.zeros(1, device=device)
            dist.all_reduce(t)                     # brings the communicator up (and its banner out) here
            torch.cuda.synchronize()
            # the library's own RCCL communicator (pfb_comm_*; collective; None -> the torch hook is used)
            from pfb_clean_amd.dist import native_comm
            native_comm(None, device)
            torch.cuda.synchronize()
    ctx = dict(args=args, world=world, rank=rank, device=device, use_pg=use_pg)
    out = bench_pd(ctx) if args.workload == 'pd' else bench_pcg(ctx)
    headline = args.workload == 'pcg' and args.size is None and args.bands is None and args.dtype == 'f32'
    if args.configs == 'all' or (args.configs == 'auto' and headline and world == 1 and not args.force_dist):
        out["configs"] = extra_configs(ctx)
    if rank == 0:
        print(json.dumps(out))
    if use_pg:
        from pfb_clean_amd.dist import close_native_comms
        close_native_comms()
        dist.destroy_process_group()


def extra_configs(ctx):
    """The other BASELINE configs on this GPU, in the same run as the headline (so that the driver's record carries
    them): C1 (1024^2 x 1), C2 (4096^2 x 1 = the per-GPU work of the 8-GPU run; also once with the band-shard
    exchange live at world size 1, in a child process), C4 (the primal-dual iteration) and the C5 per-GPU shard
    (8192^2 x 2 bands, fp64).  Same code paths and JSON shape as `--size/--bands/--dtype/--workload`; no CPU leg.
    A config that fails reports its error instead of taking the headline down with it."""
    import copy
    import torch
    args = ctx['args']

    def brief(o):
        r = o.get("roofline") or {}
        keep = ("achieved", "frac", "unit", "ms_per_launch", "ms_per_iteration", "stage_ms", "alg_bytes_per_launch",
                "alg_bytes_per_iteration", "traffic", "traffic_source", "traffic_commit", "traffic_stale",
                "hbm_rate_from_traffic_GBs")
        b = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
             "repeats": o["repeats"], "value_min": o["value_min"], "value_max": o["value_max"], "dtype": o["dtype"],
             "workload": o["config"]["workload"], "roofline": {k: r[k] for k in keep if k in r}}
        if "allreduce_hook" in o:
            b["allreduce_hook"] = o["allreduce_hook"]
        return b

    specs = [
        ("C1_1024x1_f32", "pcg", dict(size=1024, bands=1, dtype='f32', steps=50, warmup=5, repeats=3)),
        ("C2_4096x1_f32", "pcg", dict(size=4096, bands=1, dtype='f32', steps=50, warmup=5, repeats=3)),
        ("C4_pd_2048x4_f32", "pd", dict(size=2048, bands=4, dtype='f32', steps=20, warmup=3, repeats=3)),
        ("C5_shard_8192x2_f64", "pcg", dict(size=8192, bands=2, dtype='f64', steps=50, warmup=2, repeats=3)),
    ]
    res = {}
    for name, workload, kw in specs:
        a = copy.copy(args)
        a.no_cpu, a.configs, a.workload = True, 'none', workload
        for k, v in kw.items():
            setattr(a, k, v)
        try:
            o = (bench_pd if workload == 'pd' else bench_pcg)(dict(ctx, args=a))
            res[name] = brief(o)
        except Exception as e:          # noqa: BLE001 -- reported, never fatal for the headline
            res[name] = {"error": repr(e)[:300]}
        from pfb_clean_amd.operators.psf import clear_plan_cache
        clear_plan_cache()
        torch.cuda.empty_cache()
    # C2 with the exchange of the sharded solve in the loop (RCCL communicator of ONE rank: what a GPU of the 8-GPU run
    # executes per iteration, minus the wire): its own process, bounded -- a process group is never brought up in the
    # process that owns the headline measurement
    cmd = [sys.executable, os.path.abspath(__file__), '--bands', '1', '--steps', '50', '--warmup', '5', '--repeats', '3',
           '--no-cpu', '--force-dist', '--configs', 'none']
    try:
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), PFB_COMM_TIMEOUT_S='60')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        if p.returncode != 0 or not line:
            raise RuntimeError(f"rc={p.returncode}: {p.stderr[-200:]}")
        res["C2_4096x1_f32_exchange_world1"] = brief(json.loads(line[-1]))
    except Exception as e:              # noqa: BLE001
        res["C2_4096x1_f32_exchange_world1"] = {"error": repr(e)[:300]}
    return res


def _timed_regions(run, barrier, repeats, world, device):
    """`repeats` timed regions of run() between barrier + synchronize; per-region time = max over ranks."""
    import torch
    import torch.distributed as dist
    times, last = [], None
    for _ in range(max(1, repeats)):
        barrier()
        t0 = time.perf_counter()
        last = run()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        times.append(dt)
    return times, last


def bench_pcg(ctx):
    import torch
    import torch.distributed as dist
    args, world, rank, device = ctx['args'], ctx['world'], ctx['rank'], ctx['device']
    from pfb_clean_amd.operators.psf import PsfConvPlan
    from pfb_clean_amd.operators.hessian import HessianPsf
    from pfb_clean_amd.opt.pcg import pcg_fused
    from pfb_clean_amd.dist import shard_bands, global_max

    n = args.size or 4096
    nband = args.bands or 8
    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    band0, nb = shard_bands(nband, rank, world)
    if nb == 0:
        raise SystemExit("more ranks than bands")
    Q = 2 * n

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

    def solve(iters):
        return pcg_fused(A, b, None, mdiv=sigmainv, tol=0.0, maxit=iters, minit=iters,
                         backtrack=True, distributed=ctx['use_pg'])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        solve(args.warmup)
    barrier()
    plan.set_profiling(4)        # every 4th launch group carries the stage events (~6 us each on the stream)
    times, (x, _, res) = _timed_regions(lambda: solve(args.steps), barrier, args.repeats, world, device)
    stage_ms, napply = plan.get_profile()
    plan.set_profiling(False)
    assert res.iters == args.steps and res.matvecs == args.steps + 1, (res.iters, res.matvecs)
    matvecs = res.matvecs
    elapsed = _median(times)
    value = matvecs / elapsed

    s = 4 if dtype == torch.float32 else 8
    balg_band = s * (2 * n * n + 2 * (2 * n) * (Q // 2 + 1))          # bytes per band-matvec
    roofline = None
    if napply > 0:
        conv_ms = sum(stage_ms) / napply                              # per launch group (nb bands)
        achieved = balg_band * nb / (conv_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_commit, stale = pmc_traffic(n, nb, args.dtype)
        roofline = {
            "bound": "hbm", "kernel": "fft-convolution (row_fwd + col + row_inv)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc)",
            "traffic_source": traffic_src, "traffic_commit": traffic_commit, "traffic_stale": stale,
            # bytes of the committed PMC pass over THIS run's stage times: only while the kernels are the ones it saw
            "hbm_rate_from_traffic_GBs": None if (traffic is None or stale) else round(traffic / (conv_ms * 1e-3) / 1e9, 1),
            "alg_bytes_per_launch": balg_band * nb, "launches_timed": napply,
            "ms_per_launch": round(conv_ms, 4),
            "stage_ms": {"row_fwd": round(stage_ms[0] / napply, 4),
                         "col": round(stage_ms[1] / napply, 4),
                         "row_inv": round(stage_ms[2] / napply, 4)},
            "conv_share_of_step": round(conv_ms * 1e-3 * matvecs / elapsed, 3),
        }

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu, parity = cpu_baseline(plan, psfhat, b, sigmainv, n, nband, args)

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
                   "fast_path": plan.fast_path, "allreduce_hook": bool(ctx['use_pg'])},
        "repeats": len(times), "value_min": round(matvecs / max(times), 3),
        "value_max": round(matvecs / min(times), 3),
        "band_matvecs_per_s": round(value * nband, 2),
        "matvecs": matvecs, "backtracks": res.backtracks,
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
    }
    if cpu:
        out["gpu_over_cpu"] = round(value / cpu["value"], 1)
    if ctx['use_pg'] and getattr(res, 'exchange', '') == 'rccl-native':
        out["allreduce_hook"] = {"backend": "rccl-native",
                                 "what": "pfb_comm_allreduce: RCCL all-reduce of 4-7 fp64 scalars enqueued from C on the "
                                         "solver's stream (no host callback)"}
    elif ctx['use_pg'] and getattr(res, 'hook_calls', 0):
        out["allreduce_hook"] = {"backend": os.environ.get('PFB_DIST_BACKEND', 'nccl'), "calls_per_solve": res.hook_calls,
                                 "host_us_per_call": round(1e6 * res.hook_host_s / res.hook_calls, 2),
                                 "what": "ctypes callback -> torch.distributed.all_reduce of 4-7 fp64 scalars on the solver's stream"}
    return out


def _pmc_lookup(pattern, workload, n, nb, dtype, key):
    """Newest committed PMC summary of this workload: (bytes, file, commit it was taken at, True when the kernel
    sources have changed since).  Counters cannot be read from inside the process: they are collected in separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py (tools/collect_profile.sh) and corrected as
    MI355X_MICROARCH.md prescribes; the summary records the commit and the hash of the kernel sources it saw."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        c = d.get('config', {})
        if c.get('workload', 'pcg') == workload and c.get('size') == n and c.get('bands') == nb and c.get('dtype') == dtype:
            stale = d.get('kernel_src_sha16') != kernel_src_hash(workload)
            return d.get(key), os.path.relpath(path, ROOT), d.get('commit'), stale
    return None, None, None, None


def pmc_traffic(n, nb, dtype):
    """HBM bytes per launch of the convolution kernel group (None when no summary matches)."""
    return _pmc_lookup('*hbm_traffic.json', 'pcg', n, nb, dtype, 'conv_group_hbm_bytes_per_launch')


def pmc_traffic_pd(n, nb, dtype):
    """The same for one primal-dual iteration (`--workload pd`): profiles/*pd_hbm_traffic.json."""
    return _pmc_lookup('*pd_hbm_traffic.json', 'pd', n, nb, dtype, 'hbm_bytes_per_iteration')


# ------------------------------------------------------------------------------ CPU leg
def _cpu_share(args):
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    if args.cpu_cores > 0:
        return min(args.cpu_cores, avail), avail
    # 8 bands x 8 FFT workers: the fastest setting of a sweep on the 256-core MI355X host (16 cores 1.90, 64 cores
    # 2.28, 256 cores 0.90 cube-matvecs/s: scipy.fft stops scaling and the bands start to fight over memory)
    return min(64, avail), avail


def cpu_worker(d, band, threads, iters, sigmainv):
    """Child process of cpu_baseline: the oracle's pcg on ONE band with `threads` FFT workers."""
    import numpy as np
    band, threads, iters, sigmainv = int(band), int(threads), int(iters), float(sigmainv)
    from oracle import fftconv as ofc, solvers as osv
    psfhat = np.load(os.path.join(d, f'psfhat{band}.npy'), mmap_mode='r')
    psfhat = np.ascontiguousarray(psfhat)
    b = np.load(os.path.join(d, f'b{band}.npy'))
    Q = 2 * (psfhat.shape[1] - 1)
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, b.shape, b.dtype)

    def A(v):
        return ofc._hessian_psf_slice(xpad, xhat, xout, psfhat, None, Q, v, nthreads=threads,
                                      sigmainv=b.dtype.type(sigmainv))
    y = A(b)                                            # warm-up matvec (plans, page faults) + parity vector
    if band == 0:
        np.save(os.path.join(d, 'Ab0.npy'), y)
    open(os.path.join(d, f'ready{band}'), 'w').close()
    go = os.path.join(d, 'go')
    while not os.path.exists(go):
        time.sleep(0.005)
    t0 = time.time()
    osv.pcg(A, b, None, M=lambda v: v / b.dtype.type(sigmainv), tol=0.0, maxit=iters, minit=iters, verbosity=0)
    t1 = time.time()
    with open(os.path.join(d, f'done{band}.json.tmp'), 'w') as f:
        json.dump({"t0": t0, "t1": t1}, f)
    os.replace(os.path.join(d, f'done{band}.json.tmp'), os.path.join(d, f'done{band}.json'))
    return 0


def cpu_baseline(plan, psfhat_dev, b_dev, sigmainv, n, nband, args):
    """All `nband` bands of the cube solved CONCURRENTLY by the oracle (numpy + scipy.fft), one child
    process per band with cores/nband FFT workers each -- the cube PCG's arithmetic per band is the
    per-band PCG's, and the reference runs bands side by side (pcg.py:320-356) -- for a bounded number of
    iterations.  Cube rate = (iterations + 1) / wall time of the slowest band.  Also returns the parity
    object: one band-matvec of the oracle against the GPU plan on the same vector."""
    import numpy as np
    import torch
    import scipy
    cores, avail = _cpu_share(args)
    threads = max(1, cores // nband)
    base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
    d = tempfile.mkdtemp(prefix='pfb_bench_', dir=base)
    procs = []
    try:
        for k in range(nband):
            np.save(os.path.join(d, f'psfhat{k}.npy'), psfhat_dev[k].cpu().numpy())
            np.save(os.path.join(d, f'b{k}.npy'), b_dev[k].cpu().numpy())
        # one matvec costs ~ t_one with `threads` workers: calibrate on the host before choosing the count
        t_one = _calibrate_matvec(d, threads, sigmainv)
        iters = int(max(2, min(50, args.cpu_seconds / max(t_one * 1.5, 1e-6))))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        for k in range(nband):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', d, str(k),
                                           str(threads), str(iters), repr(float(sigmainv))], env=env))
        deadline = time.time() + 600
        while not all(os.path.exists(os.path.join(d, f'ready{k}')) for k in range(nband)):
            if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
                raise RuntimeError("cpu_baseline worker failed during setup")
            time.sleep(0.02)
        open(os.path.join(d, 'go'), 'w').close()
        for p in procs:
            if p.wait(timeout=1200) != 0:
                raise RuntimeError("cpu_baseline worker failed")
        marks = [json.load(open(os.path.join(d, f'done{k}.json'))) for k in range(nband)]
        wall = max(m["t1"] for m in marks) - min(m["t0"] for m in marks)
        rate = (iters + 1) / wall
        # parity: the same band-0 matvec on the GPU
        ref = np.load(os.path.join(d, 'Ab0.npy'))
        got = plan.apply(b_dev[0:1].contiguous(), sigmainv=sigmainv, band0=0)[0].cpu().numpy()
        relerr = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / np.abs(ref).max())
        tol = 1e-5 if ref.dtype == np.float32 else 1e-12
        parity = {"conv_relerr": relerr, "tol": tol, "ok": bool(relerr <= tol),
                  "what": f"max|gpu - cpu| / max|cpu| of hessian_psf on band 0 of the dirty cube ({n}x{n}, "
                          f"{'fp32' if ref.dtype == np.float32 else 'fp64'}), oracle/fftconv._hessian_psf_slice vs pfb_psfconv_apply"}
        cpu = {"value": round(rate, 4), "unit": "cube-matvecs/s", "cores": threads * nband, "kind": "port",
               "sample": f"oracle pcg (numpy + scipy.fft {scipy.__version__}) on all {nband} bands concurrently, "
                         f"{nband} processes x {threads} FFT workers, {iters} iterations = {iters + 1} cube-matvecs "
                         f"in {wall:.1f} s; host has {os.cpu_count()} logical cores, {avail} usable by this process; "
                         "core-count sweep on the 256-core MI355X host (tools/cpu_workers_sweep.py, round 2): 16 cores 1.90, "
                         "64 cores 2.28, 256 cores 0.90 cube-matvecs/s -- 64 is the fastest setting and the default",
               "band_matvecs_per_s": round(rate * nband, 3)}
        return cpu, parity
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)


def _calibrate_matvec(d, threads, sigmainv):
    import numpy as np
    from oracle import fftconv as ofc
    psfhat = np.load(os.path.join(d, 'psfhat0.npy'))
    b = np.load(os.path.join(d, 'b0.npy'))
    Q = 2 * (psfhat.shape[1] - 1)
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, b.shape, b.dtype)
    ofc._hessian_psf_slice(xpad, xhat, xout, psfhat, None, Q, b, nthreads=threads, sigmainv=b.dtype.type(sigmainv))
    t0 = time.perf_counter()
    ofc._hessian_psf_slice(xpad, xhat, xout, psfhat, None, Q, b, nthreads=threads, sigmainv=b.dtype.type(sigmainv))
    # with all bands running side by side the memory system is shared: expect ~1.5x this
    return time.perf_counter() - t0


# ------------------------------------------------------------------------ primal-dual (C4)
def bench_pd(ctx):
    """BASELINE config #4: primal_dual_optimised (primal_dual.py:91-180) on 2048^2 x 4 bands with the
    dictionary self + db1..db4, 3 levels; grad(x) = psf_convolve_cube(x) - dirty as in
    workers/spotless.py:259-260.  One step = one primal-dual iteration."""
    import torch
    import torch.distributed as dist
    args, world, rank, device = ctx['args'], ctx['world'], ctx['rank'], ctx['device']
    if world > 1:
        raise SystemExit("--workload pd is a one-GPU workload (config #4)")
    from pfb_clean_amd.operators.psf import PsfConvPlan
    from pfb_clean_amd.operators.psi import Psi
    from pfb_clean_amd.opt.primal_dual import primal_dual_optimised, PsfGradient

    n = args.size or 2048
    nband = args.bands or 4
    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    bases = ['self', 'db1', 'db2', 'db3', 'db4']
    nlevel = 3
    Q = 2 * n
    psfhat = torch.stack([synth_band(k, nband, n, n, dtype, device) for k in range(nband)])
    plan = PsfConvPlan(psfhat, n, n, Q)
    model = torch.stack([synth_model(k, n, n, dtype, device) for k in range(nband)])
    dirty = plan.apply(model).clone()
    gen = torch.Generator(device=device)
    gen.manual_seed(1420)
    dirty += 1e-3 * torch.randn(dirty.shape, dtype=dtype, device=device, generator=gen)
    del model
    psi = Psi(nband, n, n, bases, nlevel, 1, dtype=dtype)
    nbasis = len(bases)
    grad = PsfGradient(plan, dirty)                          # spotless.py:259-260: conv(x) - dirty
    w = torch.ones((nbasis, psi.Nymax, psi.Nxmax), dtype=dtype, device=device)
    lam = 1e-3 * float(dirty.abs().max().item())
    L = 1.0                                                   # sum_band psf peaks at 1 => ||A^H A|| ~ 1

    def run(iters):
        x = torch.zeros_like(dirty)
        v = torch.zeros((nband, nbasis, psi.Nymax, psi.Nxmax), dtype=dtype, device=device)
        return primal_dual_optimised(x, v, lam, psi.hdot, psi.dot, L, None, w, None, grad, nu=nbasis,
                                     tol=0.0, maxit=iters, positivity=1, verbosity=0)

    def barrier():
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    times, _ = _timed_regions(lambda: run(args.steps), barrier, args.repeats, world, device)
    elapsed = _median(times)
    value = args.steps / elapsed
    s = 4 if dtype == torch.float32 else 8
    N = n * n
    coef = nbasis * psi.Nymax * psi.Nxmax
    # algorithmic bytes of one iteration: analysis reads the image once per basis and writes the
    # coefficients, synthesis the reverse with ONE image write, the dual update reads v, vp, w and writes
    # v, vp, the convolution B_alg per band, the gradient subtraction 3 N, the primal update reads xp,
    # xout, g and writes x
    balg = s * nband * ((N + coef) + (coef + N) + 4 * coef + (2 * N + 2 * (2 * n) * (Q // 2 + 1)) + 3 * N + 4 * N) \
        + s * coef
    achieved = balg / elapsed * args.steps / 1e9
    traffic, traffic_src, traffic_commit, stale = pmc_traffic_pd(n, nband, args.dtype)
    roofline = {"bound": "hbm", "kernel": "primal-dual iteration (psi^H, dual update, psi, PSF conv, primal update)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per iteration (FETCH_SIZE x2 + WRITE_SIZE over the iteration's kernels, rocprofv3 --pmc)",
                "traffic_source": traffic_src, "traffic_commit": traffic_commit, "traffic_stale": stale,
                "hbm_rate_from_traffic_GBs": round(traffic / elapsed * args.steps / 1e9, 1) if (traffic and not stale) else None,
                "alg_bytes_per_iteration": balg, "ms_per_iteration": round(1e3 * elapsed / args.steps, 4),
                "timing": "wall clock of the timed region / steps (the iteration is several launches; the host "
                          "reads three scalars per iteration)"}
    out = {
        "metric": "primal-dual iterations/sec (psi/psi^H + L1 prox + PSF-conv gradient), 2k x 2k x 4-band (BASELINE config 4)",
        "value": round(value, 3), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{n}x{n}x{nband}-band primal_dual_optimised, bases {'+'.join(bases)}, {nlevel} levels, "
                               f"positivity=1, tol=0, maxit={args.steps}",
                   "coeff_shape": [nband, nbasis, psi.Nymax, psi.Nxmax], "fast_path": plan.fast_path},
        "repeats": len(times), "value_min": round(args.steps / max(times), 3), "value_max": round(args.steps / min(times), 3),
        "roofline": roofline, "cpu_baseline": None,
    }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_pd(psfhat, dirty, lam, L, n, nband, bases, nlevel, args)
        if out["cpu_baseline"]:
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    return out


def cpu_baseline_pd(psfhat_dev, dirty_dev, lam, L, n, nband, bases, nlevel, args):
    """The oracle's primal_dual_optimised (numpy wavelets + scipy.fft convolution) on the same cube for a
    bounded number of iterations, scipy.fft with the process's CPU share."""
    import numpy as np
    import scipy
    from oracle import fftconv as ofc, solvers as osv, wavelets as owv
    cores, avail = _cpu_share(args)
    psfhat = psfhat_dev.cpu().numpy()
    dirty = dirty_dev.cpu().numpy()
    Q = 2 * n
    xpad, xhat, xout = ofc.make_scratch(psfhat, Q, dirty.shape, dirty.dtype)
    psi = owv.Psi(nband, n, n, bases, nlevel)
    nbasis = len(bases)

    def grad(v):
        return ofc.psf_convolve_cube(xpad, xhat, xout, psfhat, Q, v, nthreads=cores) - dirty
    w = np.ones((nbasis, psi.Nymax, psi.Nxmax), dtype=dirty.dtype)

    def run(iters):
        x = np.zeros_like(dirty)
        v = np.zeros((nband, nbasis, psi.Nymax, psi.Nxmax), dtype=dirty.dtype)
        t0 = time.perf_counter()
        osv.primal_dual_optimised(x, v, lam, psi.hdot, psi.dot, L, None, w, None, grad, nu=nbasis, tol=0.0,
                                  maxit=iters, positivity=1, verbosity=0)
        return time.perf_counter() - t0
    t1 = run(1)
    iters = int(max(1, min(20, args.cpu_seconds / max(t1, 1e-6))))
    dt = run(iters)
    return {"value": round(iters / dt, 4), "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"oracle primal_dual_optimised (numpy wavelets, scipy.fft {scipy.__version__} workers={cores}) on the "
                      f"same {n}x{n}x{nband} cube, {iters} iterations in {dt:.1f} s; the numpy wavelet passes are "
                      f"single-threaded; host has {os.cpu_count()} logical cores, {avail} usable by this process"}


if __name__ == '__main__':
    sys.exit(main())
