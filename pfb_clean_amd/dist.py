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
    synchronisation).
"""
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


class AllReduceHook:
    """Callable(buf_address, count) used by opt.pcg.pcg_fused: sums `count` fp64 scalars
    living inside `work` (the solver's device scratch) over the process group, in place.
    Device agnostic (CPU tensors + gloo in the tests, GPU tensors + RCCL in production)."""

    def __init__(self, work, group=None):
        self.work = work
        self.group = group
        self.calls = 0
        self.host_s = 0.0            # host time spent inside the hook (enqueue of the collective), for bench.py

    def __call__(self, addr, count):
        import time
        t0 = time.perf_counter()
        off = addr - self.work.data_ptr()
        if off < 0 or off + 8 * count > self.work.numel() * self.work.element_size():
            raise ValueError("allreduce buffer outside the solver workspace")
        view = self.work.view(torch.uint8)[off:off + 8 * count].view(torch.float64)
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        self.host_s += time.perf_counter() - t0


def global_max(value, device, group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.item()
