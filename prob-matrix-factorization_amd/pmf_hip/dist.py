"""Multi-GPU orchestration of one CAVI iteration.

Ratings are sharded by USER RANGE: rank g owns a contiguous block of users,
all of their ratings and their theta-side state; the item-side state is
replicated.  The user half-sweeps are purely local.  An item half-sweep is

    local raw sums over this rank's ratings  ->  all-reduce (RCCL)  ->  finalise

so every rank ends each iteration with identical item factors.  The exchanged
quantity is the per-item sufficient statistic, not a gradient: [I x 2Kpad] for
Poisson/HPF, [I x (Kp + Kpad)] (packed normal matrix + right-hand side) and
[I x 2] (bias) for the Gaussian model.

The functions here only sequence engine calls and collectives.  `engine` is any
object with the `Context` half-sweep methods (pmf_hip.Context on a GPU; the CPU
tests drive the same code with an oracle-backed stand-in over gloo), `comm` is
a `Comm` (or None for a single process).
"""
from __future__ import annotations

import numpy as np

from . import ITEM, USER


class Comm:
    """torch.distributed wrapper: backend 'nccl' (= RCCL on ROCm) for device
    tensors, 'gloo' for the CPU tests."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce(self, tensor):
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_async(self, tensor):
        """Returns the work handle; `.wait()` orders the current stream after the collective."""
        return self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group, async_op=True)

    def barrier(self):
        self._dist.barrier(group=self.group)


def shard_bounds(user_ids, n_users, world):
    """User-range boundaries [b_0 = 0, ..., b_world = n_users] balanced by the
    number of ratings per shard, not by the number of users.  Every range holds at
    least one user (a function of the inputs only, so all ranks agree -- and all
    ranks raise together when there are fewer users than ranks)."""
    if n_users < world:
        raise ValueError(f"cannot shard {n_users} users over {world} ranks")
    counts = np.bincount(np.asarray(user_ids, dtype=np.int64), minlength=n_users)
    cum = np.cumsum(counts)
    total = int(cum[-1]) if len(cum) else 0
    bounds = [0]
    for g in range(1, world):
        target = total * g / world
        b = int(np.searchsorted(cum, target, side="left")) + 1
        bounds.append(min(max(b, bounds[-1] + 1), n_users - (world - g)))
    bounds.append(n_users)
    return np.asarray(bounds, dtype=np.int64)


def take_shard(user_ids, item_ids, ratings, bounds, rank):
    """This rank's ratings with user ids made local to its range (original
    order kept, so per-row summation order is unchanged)."""
    u = np.asarray(user_ids)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sel = (u >= lo) & (u < hi)
    return u[sel] - lo, np.asarray(item_ids)[sel], np.asarray(ratings)[sel]


# ---- one iteration per model ------------------------------------------------
def _item_half_sweep(engine, comm, stats, width, accumulate, finalize):
    """accumulate -> all-reduce -> finalize over the item rows.  When the engine has item row
    chunks (`set_row_chunks(ITEM, n)`, the same n on every rank) the three stages are pipelined:
    the statistics of chunk c travel (asynchronous all-reduce on the collective's own stream)
    while chunk c+1 is accumulated, and chunk c is finalised while later chunks still travel.
    The arithmetic per row is that of the unchunked call, so results do not depend on n."""
    n = getattr(engine, "n_chunks", {}).get(ITEM, 1)
    if n <= 1:
        accumulate()
        comm.all_reduce(stats.tensor)
        finalize()
        return
    works = []
    try:
        for c in range(n):
            engine.select_chunk(ITEM, c)
            accumulate()
            lo, hi = engine.chunk_rows(ITEM, c)
            works.append(comm.all_reduce_async(stats.tensor[lo * width:hi * width]))
        for c in range(n):
            works[c].wait()
            engine.select_chunk(ITEM, c)
            finalize()
    finally:
        engine.select_chunk(ITEM, -1)


def gamma_iteration(engine, comm, stats_item, user_prior, item_prior):
    """Poisson MF / HPF.  `*_prior` = (shape_prior, rate_prior, hierarchical,
    hyper_shape, hyper_rate_prior) as taken by `gamma_sweep`.
    `stats_item` is the [I x 2 x Kpad] buffer object with `.ptr` and `.tensor`."""
    engine.gamma_sweep(USER, *user_prior)
    if comm is None or comm.world == 1:
        engine.gamma_sweep(ITEM, *item_prior)
        return
    _item_half_sweep(engine, comm, stats_item, 2 * engine.kpad,
                     lambda: engine.gamma_accumulate(ITEM, stats_item.ptr),
                     lambda: engine.gamma_finalize(ITEM, stats_item.ptr, *item_prior))


def gaussian_iteration(engine, comm, stats_item, stats_bias, sigma2, eta_theta2, eta_beta2,
                       eta_bias2=None):
    """Gaussian MF (+ biases when eta_bias2 is given), order as
    gaussian_mf_cavi_bias.py:129-263."""
    single = comm is None or comm.world == 1
    engine.gauss_factor_sweep(USER, sigma2, eta_theta2)
    if single:
        engine.gauss_factor_sweep(ITEM, sigma2, eta_beta2)
    else:
        _item_half_sweep(engine, comm, stats_item, engine.cov_stride + engine.kpad,
                         lambda: engine.gauss_factor_accumulate(ITEM, stats_item.ptr),
                         lambda: engine.gauss_factor_finalize(ITEM, stats_item.ptr, sigma2, eta_beta2))
    if eta_bias2 is None:
        return
    engine.gauss_bias_sweep(USER, sigma2, eta_bias2)
    if single:
        engine.gauss_bias_sweep(ITEM, sigma2, eta_bias2)
    else:
        # [I x 2]: latency-bound, one message (chunk selection is -1 = all rows here)
        engine.gauss_bias_accumulate(ITEM, stats_bias.ptr)
        comm.all_reduce(stats_bias.tensor)
        engine.gauss_bias_finalize(ITEM, stats_bias.ptr, sigma2, eta_bias2)


def gaussian_sgd_iteration(engine, comm, stats_item, lr, sigma2, eta_theta2, eta_beta2, eta_bias2):
    """One epoch of the MAP / gradient mode (no reference counterpart): users, then items; on
    several ranks the items' rating-count-weighted displacement sums are all-reduced."""
    engine.gauss_sgd_sweep(USER, lr, sigma2, eta_theta2, eta_bias2)
    if comm is None or comm.world == 1:
        engine.gauss_sgd_sweep(ITEM, lr, sigma2, eta_beta2, eta_bias2)
        return
    _item_half_sweep(engine, comm, stats_item, engine.sgd_stats_width,
                     lambda: engine.gauss_sgd_accumulate(ITEM, stats_item.ptr, lr, sigma2, eta_beta2, eta_bias2),
                     lambda: engine.gauss_sgd_finalize(ITEM, stats_item.ptr))


class StreamScope:
    """Puts one engine context and torch (its allocator, its collectives) on the SAME
    non-default HIP stream, so that kernels, all-reduces and copies are ordered without
    host synchronisation.  (torch's default stream has the null handle, which
    `pmf_ctx_set_stream` would read as "use the context's own stream".)

        scope = StreamScope(ctx, device); scope.enter()   ...   scope.exit()
    """

    def __init__(self, ctx, device):
        import torch
        self._torch = torch
        self.stream = torch.cuda.Stream(device=device)
        self._ctx = ctx
        self._cm = None

    def enter(self):
        assert self.stream.cuda_stream != 0
        self._ctx.set_stream(self.stream.cuda_stream)
        self._cm = self._torch.cuda.stream(self.stream)
        self._cm.__enter__()
        return self

    def exit(self):
        if self._cm is not None:
            self.stream.synchronize()
            self._cm.__exit__(None, None, None)
            self._cm = None
            self._ctx.set_stream(None)

    __enter__ = enter

    def __exit__(self, *exc):
        self.exit()


class DeviceStats:
    """A device buffer owned by torch (so RCCL can all-reduce it) whose raw
    pointer is handed to the C-ABI accumulate / finalize calls."""

    def __init__(self, n_elems, np_dtype, device):
        import torch
        tdt = torch.float64 if np_dtype == np.float64 else torch.float32
        self.tensor = torch.zeros(int(n_elems), dtype=tdt, device=device)
        self.ptr = self.tensor.data_ptr()


def default_item_chunks(world, message_bytes=0):
    """Item row chunks of a sharded run.  PMF_DIST_CHUNKS if set; else slices of about 256 MB,
    at least 4 (so that at most a quarter of the all-reduce is exposed) and at most 32, but never
    below 4 MB per slice (latency-bound collectives).  1 = no pipelining."""
    import os
    if world <= 1:
        return 1
    if "PMF_DIST_CHUNKS" in os.environ:
        return max(1, int(os.environ["PMF_DIST_CHUNKS"]))
    n = min(max(int(message_bytes) // (256 << 20), 4), 32)
    return max(1, min(n, max(1, int(message_bytes) // (4 << 20)))) if message_bytes else 4


def item_message_bytes(ctx, gaussian):
    """Bytes of the item statistics one iteration all-reduces (the factor half-sweep's message)."""
    width = (ctx.cov_stride + ctx.kpad) if gaussian else 2 * ctx.kpad
    return ctx.n_items * width * np.dtype(ctx.np_dtype).itemsize


def gamma_stats(ctx, device):
    return DeviceStats(ctx.n_items * 2 * ctx.kpad, ctx.np_dtype, device)


def sgd_stats(ctx, device):
    return DeviceStats(ctx.n_items * ctx.sgd_stats_width, ctx.np_dtype, device)


def gauss_stats(ctx, device):
    return (DeviceStats(ctx.n_items * (ctx.cov_stride + ctx.kpad), ctx.np_dtype, device),
            DeviceStats(ctx.n_items * 2, ctx.np_dtype, device))
