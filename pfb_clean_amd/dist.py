"""
Band sharding across the GPUs of one node (one process per GPU, torch.distributed with
the `nccl` backend = RCCL over xGMI).

The reference parallelises over imaging bands with dask (pcg.py:320-356 blockwise per
band; psi.py:284-310 thread pool over bands).  Here each rank owns a contiguous slice of
bands -- its psfhat, beam and every CG vector for those bands -- and no image data ever
crosses GPUs:

  * per-band PCG (pcg_psf semantics, klean.py:310-317): bands are independent ->
    replicas, no collective at all;
  * cube PCG (one system over all bands, fluxmop.py:193-199): the only exchange is the
    sum of the CG inner products -- per iteration [p.Ap], [r.y, |x-xp|^2, |x|^2] and
    [any(p)], 8..24 bytes each -- done by ONE small all-reduce per reduction point on
    the solver's stream (latency-bound).  pfb_pcg_solve calls back into
    `AllReduceHook` with the device address of the scalars; the hook wraps that memory
    as a tensor view and issues torch.distributed.all_reduce (stream-ordered, no host
    synchronisation).  On GPUs with the `nccl` backend the exchange does not go through
    Python at all: `native_comm()` builds an RCCL communicator inside libpfb_hip
    (include/pfb_hip.h: pfb_comm_*) and pfb_pcg_solve is handed the C function
    pfb_comm_allreduce, which enqueues the all-reduce on the solver's own stream.  The
    torch hook is the DEFAULT beyond world size 1 (the native exchange has not run on more than one GPU
    yet: opt in with PFB_NATIVE_COMM=1) and the fallback (gloo, CPU rehearsals, any rank failing to
    build the communicator -- the ranks agree on the choice collectively).

Failure handling (include/pfb_hip.h, "Failure protocol"): neither exchange waits unboundedly.  A rank whose hook
fails, whose communicator reports an asynchronous error, or whose stream makes no progress for
PFB_COMM_TIMEOUT_S seconds aborts the exchange and raises _lib.PfbCommError; it must exit non-zero.
"""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist


def shard_bands(nband, rank, world):
    """Contiguous, balanced partition: returns (band0, nb) for `rank`.  The first
    nband % world ranks get one extra band; ranks beyond nband get nb = 0."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(nband, world)
    nb = base + (1 if rank < extra else 0)
    band0 = rank * base + min(rank, extra)
    return band0, nb


class ExchangeFailed(RuntimeError):
    """Raised inside AllReduceHook when the collective could not complete (peer gone, timeout)."""


def comm_timeout_s():
    """Seconds a rank waits for the exchange before it declares it failed (PFB_COMM_TIMEOUT_S, default 600): the
    same knob bounds the C solver's waits for the device (include/pfb_hip.h, "Failure protocol")."""
    try:
        v = float(os.environ.get('PFB_COMM_TIMEOUT_S', '600'))
    except ValueError:
        v = 600.0
    return v if v > 0 else 600.0


class AllReduceHook:
    """Callable(buf_address, count) used by opt.pcg.pcg_fused: sums `count` fp64 scalars
    living inside `work` (the solver's device scratch) over the process group, in place.
    Device agnostic (CPU tensors + gloo in the tests, GPU tensors + RCCL in production).

    It also speaks the failure protocol of pfb_allreduce_fn (include/pfb_hip.h): count == 0 is a PROBE (raises once
    the exchange has failed), count < 0 an ABORT request.  A collective on host tensors (gloo) is waited for with a
    timeout -- a peer that died or never arrives turns into ExchangeFailed within PFB_COMM_TIMEOUT_S instead of a
    hang; on GPU tensors the collective is stream-ordered and the C solver's bounded wait does the timing.  After a
    failure the hook refuses every further call: the reference's sums run over all bands in one process
    (pfb/opt/pcg.py:90-107) and a partial sum must never be mistaken for one."""

    def __init__(self, work, group=None):
        self.work = work
        self.group = group
        self.calls = 0
        self.host_s = 0.0            # host time spent inside the hook (enqueue of the collective), for bench.py
        self.failed = None           # text of the first failure

    def __call__(self, addr, count):
        import time
        from datetime import timedelta
        if count < 0:                # abort: nothing may use this exchange any more
            self.failed = self.failed or 'aborted by the solver'
            abort_exchange(self.group)
            return
        if self.failed:
            raise ExchangeFailed(self.failed)
        if count == 0:               # probe: a torch process group has no portable asynchronous error query
            return
        t0 = time.perf_counter()
        off = addr - self.work.data_ptr()
        if off < 0 or off + 8 * count > self.work.numel() * self.work.element_size():
            raise ValueError("allreduce buffer outside the solver workspace")
        view = self.work.view(torch.uint8)[off:off + 8 * count].view(torch.float64)
        try:
            if view.is_cuda:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            else:
                w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if w.wait(timeout=timedelta(seconds=comm_timeout_s())) is False:
                    raise ExchangeFailed("all-reduce timed out")
        except Exception as e:       # a closed connection, a timeout, an aborted group
            self.failed = f"all-reduce over the band shards failed: {e!r}"
            raise ExchangeFailed(self.failed) from e
        self.calls += 1
        self.host_s += time.perf_counter() - t0


def abort_exchange(group=None):
    """Tear down what this process holds of the exchange over `group` so that no peer stays blocked on it: the
    library-owned RCCL communicator is aborted (ncclCommAbort makes the peers' queued collectives fail); torch's own
    process group is left to its watchdog / to process exit -- a rank that gets here is expected to exit non-zero."""
    for key, c in list(_native.items()):
        if c is not None and key[0] is group:
            try:
                c.abort()
            except Exception:
                pass


class NativeComm:
    """An RCCL communicator owned by libpfb_hip (pfb_comm_*), one per process group.  `fn` / `ctx` are what
    pfb_pcg_solve takes as (allreduce, allreduce_ctx)."""

    def __init__(self, handle, lib):
        from . import _lib
        self.handle = handle
        self._lib = lib
        self.fn = C.cast(lib.pfb_comm_allreduce, _lib.ALLREDUCE_FN)
        self.ctx = handle
        r, n, d, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.pfb_comm_info(handle, C.byref(r), C.byref(n), C.byref(d), C.byref(v)))
        self.rank, self.world, self.device, self.rccl_version = r.value, n.value, d.value, v.value

    def all_reduce_(self, t):
        """In-place sum of a contiguous fp64 device tensor over the ranks, on torch's current stream."""
        from . import _lib, _dev
        if t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("NativeComm.all_reduce_: contiguous fp64 GPU tensor expected")
        _lib.check(self._lib.pfb_comm_allreduce(self.handle, _dev.ptr(t), t.numel(), _dev.stream()))
        return t

    def check(self):
        """Raise PfbCommError once a collective on this communicator has failed or it was aborted."""
        from . import _lib
        _lib.check(self._lib.pfb_comm_check(self.handle))

    def abort(self):
        if self.handle:
            self._lib.pfb_comm_abort(self.handle)

    def close(self):
        if self.handle:
            self._lib.pfb_comm_destroy(self.handle)
            self.handle = self.ctx = None


_native = {}          # process group -> NativeComm | None (None: tried, not available -> torch hook)


def _loaded_rccl_path():
    """The librccl this process already has mapped (torch's own copy), so that libpfb_hip binds the same one."""
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                if 'librccl' in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


def native_comm(group=None, device=None):
    """The libpfb_hip RCCL communicator for `group` (built on first use: COLLECTIVE over the group), or None
    when the exchange has to go through torch.distributed.  Every rank gets the same answer."""
    live = dist.is_available() and dist.is_initialized()
    # keyed on the shape of the process group as well: a communicator outlives the torch group it was built over, and
    # after destroy / re-init with another backend or size it must not be reused (every rank computes the same key, so
    # building a new one below stays collective)
    key = (group, dist.get_backend(group), dist.get_world_size(group), dist.get_rank(group)) if live else (group,)
    if key in _native:
        return _native[key]
    comm = None
    # The library-owned communicator has only ever run at world size 1 (no multi-GPU node was available to this build):
    # beyond that it is OPT-IN (PFB_NATIVE_COMM=1) and the torch.distributed hook -- the path the world-2 / 4 / 8 gloo
    # tests cover -- is the default.  PFB_NATIVE_COMM=0 switches it off at every size.
    want = os.environ.get('PFB_NATIVE_COMM', '')
    usable = (want != '0' and live and dist.get_backend(group) == 'nccl' and torch.cuda.is_available()
              and (want == '1' or dist.get_world_size(group) == 1))
    if usable:
        from . import _lib
        lib = _lib.load()
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ok, err = 1, ''
        idbuf = (C.c_ubyte * 128)()
        # everything that can fail LOCALLY happens before the first agreement, so that the one collective step left
        # (ncclCommInitRank inside pfb_comm_init) is only entered when every rank is known to arrive
        try:
            path = _loaded_rccl_path()
            _lib.check(lib.pfb_comm_bind(path.encode() if path else None))
            if dev.type != 'cuda' or not (0 <= (dev.index or 0) < torch.cuda.device_count()):
                raise RuntimeError(f"no such device {dev}")
            with torch.cuda.device(dev):
                torch.cuda.current_stream().synchronize()       # the device answers
            if rank == 0:
                _lib.check(lib.pfb_comm_unique_id(idbuf))
        except Exception as e:      # this rank cannot: tell the others below
            ok, err = 0, repr(e)
        # the 128-byte id travels over the existing process group; a failed rank 0 sends zeros and flags it
        t = torch.tensor(list(bytes(idbuf)) + [ok], dtype=torch.uint8, device=dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        flag = torch.tensor([float(ok)], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if flag.item() >= 1.0 and int(t[-1].item()) == 1:
            ids = bytes(t[:128].cpu().tolist())
            handle = C.c_void_p()
            with torch.cuda.device(dev):
                code = lib.pfb_comm_init(rank, world, ids, C.byref(handle))
            ok = 1 if code == 0 else 0
            if not ok:
                err = (lib.pfb_last_error() or b'').decode()
            flag = torch.tensor([float(ok)], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if flag.item() >= 1.0:
                comm = NativeComm(handle, lib)
                _register_atexit()
            elif ok:
                lib.pfb_comm_destroy(handle)
        if comm is None and err:
            print(f"pfb_clean_amd: native RCCL communicator not available on rank {rank} ({err}); "
                  "using the torch.distributed hook", file=sys.stderr)
    _native[key] = comm
    return comm


def close_native_comms():
    for c in _native.values():
        if c is not None:
            c.close()
    _native.clear()


_atexit_done = False


def _register_atexit():
    """The communicators are destroyed when the interpreter exits, BEFORE torch tears its process group down (atexit
    runs in reverse registration order and this is registered after torch.distributed was initialised)."""
    global _atexit_done
    if not _atexit_done:
        import atexit
        atexit.register(close_native_comms)
        _atexit_done = True


def global_max(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.item()


# ------------------------------------------------------------------ coefficient-plane exchange (primal-dual, band-sharded)
def plane_chunks(nper, itemsize, world):
    """Chunk boundaries [(off, count), ...] of an nper-element plane for the pipelined exchange: about 32 MiB per chunk
    (PFB_PD_CHUNK_MB), at most 16 chunks, every boundary a multiple of world x 64 elements so that a chunk splits evenly
    into the reduce-scatter shards and each shard stays 256-byte aligned; the last chunk takes the remainder."""
    try:
        mb = float(os.environ.get('PFB_PD_CHUNK_MB', '32'))
    except ValueError:
        mb = 32.0
    quantum = 64 * max(1, int(world))
    target = max(quantum, int(mb * (1 << 20) / itemsize))
    nchunk = max(1, min(16, -(-nper // target)))
    per = -(-nper // nchunk)
    per = -(-per // quantum) * quantum
    out, off = [], 0
    while off < nper:
        cnt = min(per, nper - off)
        out.append((off, cnt))
        off += cnt
    return out


def exchange_plane_pipelined(plane, bandsum, apply, group=None):
    """The band sum of the l21 dual update (prox_21m.py:89-103 sums over ALL bands) with the bands sharded over ranks:
    `plane` (1-D view of the (nbasis, nymax, nxmax) coefficient plane) is filled chunk by chunk with the LOCAL sums by
    bandsum(off, count), summed over the ranks, and handed to apply(off, count) -- software-pipelined so that the
    exchange of chunk c overlaps the band sums of chunk c + 1 and the threshold of chunk c - 1:

        bandsum(0) | exchange(0)   bandsum(1) | exchange(1)   apply(0)  bandsum(2) | exchange(2)  apply(1) ...

    With the `nccl` backend (RCCL) the exchange of a chunk is an explicit REDUCE-SCATTER followed by an ALL-GATHER,
    both asynchronous on RCCL's stream (the plane is bandwidth bound -- 86 MB at config #4, 2.7 GB at
    config #5 -- and a ring over all xGMI links moves 2 (W-1)/W of it per rank either way; the explicit pair keeps
    each collective a single-algorithm, full-ring transfer and lets the first all-gathers start while later chunks
    are still being reduced).  Other backends (gloo: CPU rehearsals, two ranks on one test GPU) use all_reduce per
    chunk.  PFB_PD_RSAG=0 forces all_reduce.  Returns the number of chunks."""
    pg = None if group is True else group
    world = dist.get_world_size(pg)
    n = plane.numel()
    chunks = plane_chunks(n, plane.element_size(), world)
    rsag = (dist.get_backend(pg) == 'nccl' and plane.is_cuda and os.environ.get('PFB_PD_RSAG', '1') != '0')
    pending = []                      # (off, count, work handle)

    def finish(item):
        off, cnt, work = item
        for w in work:
            w.wait()                  # device-side for nccl (the current stream waits for RCCL's), host-side for gloo
        apply(off, cnt)

    for off, cnt in chunks:
        bandsum(off, cnt)
        view = plane[off:off + cnt]
        if rsag and cnt % world == 0:
            shard = torch.empty(cnt // world, dtype=plane.dtype, device=plane.device)
            w1 = dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.SUM, group=pg, async_op=True)
            w2 = dist.all_gather_into_tensor(view, shard, group=pg, async_op=True)   # same RCCL stream: ordered behind w1
            work = (w1, w2)
        else:
            work = (dist.all_reduce(view, op=dist.ReduceOp.SUM, group=pg, async_op=True),)
        pending.append((off, cnt, work))
        if len(pending) > 1:
            finish(pending.pop(0))
    while pending:
        finish(pending.pop(0))
    return len(chunks)
