"""Multi-GPU runs of the CAVI engine -- no PyTorch anywhere on this path.

Ratings are sharded by USER RANGE: rank g (one process per GPU) owns a contiguous block of users,
all of their ratings and their theta-side state; the item-side state is replicated.  The user
half-sweeps are purely local.  An item half-sweep is

    local raw sums over this rank's ratings  ->  all-reduce  ->  finalise

so every rank ends each iteration with identical item factors.  The exchanged quantity is the
per-item sufficient statistic, not a gradient: [I x 2Kpad] for Poisson/HPF, [I x (Kp + Kpad)]
(packed normal matrix + right-hand side) and [I x 2] (bias) for the Gaussian model.

Where the collective runs:

* `Comm` (this module): the communicator lives INSIDE libpmf_hip.so (`pmf_comm_init`: RCCL over xGMI,
  second HIP stream, event ordering, library-owned statistics buffers, chunk pipelining).  A context
  that `Comm.attach`ed simply calls `gamma_sweep(ITEM, ...)` etc.; the library does the three stages.
  `init_from_env()` builds one from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK /
  MASTER_PORT as set by `python -m torch.distributed.run` or any other launcher) and exchanges the
  RCCL unique id through a file.
* an external collective (anything with `all_reduce(stats)` / `all_reduce_async(stats)` and
  `in_library = False`): the host sequences `*_accumulate` -> collective -> `*_finalize` on a
  caller-owned statistics buffer (`_item_half_sweep`).  That is the bring-your-own-communicator form
  of the C-ABI; the CPU tests drive it over gloo with an oracle-backed engine.
"""
from __future__ import annotations

import os
import tempfile
import time

import numpy as np

from . import ITEM, UNIQUE_ID_BYTES, USER


# ---- sharding -----------------------------------------------------------------
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


# ---- the in-library communicator -------------------------------------------------
class Comm:
    """This process's communicator inside libpmf_hip.so (one rank per GPU).

    A tiny anchor context owns it; model / bench contexts share it with `attach(ctx)`, after
    which their ITEM half-sweeps are distributed by the library itself."""

    in_library = True

    def __init__(self, rank, world, device, unique_id, transport="rccl", exchange=None):
        from .engine import Context
        self.rank, self.world, self.device, self.transport = int(rank), int(world), int(device), transport
        self.exchange = exchange      # None = the library's default (PMF_COMM_EXCHANGE, else auto)
        self._anchor = Context(1, 1, 1, dtype="f32", device=self.device)
        try:
            self._anchor.comm_init(self.world, self.rank, unique_id, transport)
        except Exception:
            self._anchor.close()
            raise

    def attach(self, ctx):
        ctx.comm_attach(self._anchor)
        if self.exchange is not None:
            ctx.comm_set_exchange(self.exchange)
        return ctx

    def all_reduce_host(self, values, op="sum"):
        """Element-wise sum / max of a small host array over the ranks (validation sums, timings)."""
        return self._anchor.comm_allreduce_host(values, op)

    def barrier(self):
        self._anchor.comm_barrier()

    def close(self):
        if self._anchor is not None:
            self._anchor.close()
            self._anchor = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_exchange_seq = 0
_PROCESS_START = None


def _process_start():
    """Wall-clock time this process was started (the launcher spawns all ranks of a launch together)."""
    global _PROCESS_START
    if _PROCESS_START is None:
        try:
            with open("/proc/self/stat") as f:
                ticks = float(f.read().rsplit(")", 1)[1].split()[19])      # starttime, clock ticks since boot
            with open("/proc/uptime") as f:
                up = float(f.read().split()[0])
            _PROCESS_START = time.time() - up + ticks / os.sysconf("SC_CLK_TCK")
        except Exception:
            _PROCESS_START = time.time()
    return _PROCESS_START


def _id_file(tag):
    explicit = os.environ.get("PMF_COMM_ID_FILE")
    if explicit:
        return f"{explicit}.{tag}"
    base = os.environ.get("PMF_COMM_DIR", tempfile.gettempdir())
    # all ranks of one launch are children of one launcher process (torch.distributed.run's agent, mp.spawn, a
    # shell): its pid, the rendezvous address / port and the launcher's run id name the launch.  That is NOT
    # unique over time (a shell reuses its pid for the next launch, the run id defaults to "none"), so a file
    # is only believed when it is younger than the reading process (see exchange_unique_id).
    key = "_".join(str(os.environ.get(k, "0")) for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                                                         "TORCHELASTIC_RESTART_COUNT")) + f"_{os.getppid()}"
    key = "".join(c if c.isalnum() or c in "._-" else "-" for c in key)
    return os.path.join(base, f"pmf_hip_uid_{key}.{tag}")


def exchange_unique_id(rank, make_id, timeout=None, world=None):
    """Rank 0 creates the 128-byte id and publishes it in a file (any file left at that path by an earlier,
    crashed launch is removed first; the new one is written whole, then renamed); the other ranks wait for a
    file that is YOUNGER THAN THEIR OWN PROCESS -- a stale id from a previous launch of the same shell / port
    is never accepted.  With `world`, every rank then acknowledges the id (`<path>.ack<rank>`) and waits until
    all `world` acknowledgements exist, so a rank that never started ends in a TimeoutError here -- before
    anybody sits in ncclCommInitRank, which has no deadline of its own.  Called by every rank, the same
    number of times.  `timeout`: seconds (default PMF_COMM_INIT_TIMEOUT_S or 180)."""
    global _exchange_seq
    if timeout is None:
        timeout = float(os.environ.get("PMF_COMM_INIT_TIMEOUT_S", "180"))
    path = _id_file(_exchange_seq)
    _exchange_seq += 1
    t0 = time.time()
    if rank == 0:
        for stale in [path] + ([f"{path}.ack{r}" for r in range(world)] if world else []):
            try:
                os.remove(stale)
            except OSError:
                pass
        uid = make_id()
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
    else:
        born = _process_start() - 2.0        # (clock granularity; the launcher starts the ranks within milliseconds)
        while True:
            try:
                if os.stat(path).st_mtime >= born:
                    with open(path, "rb") as f:
                        uid = f.read()
                    if len(uid) == UNIQUE_ID_BYTES:
                        break
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rank {rank}: no fresh communicator id at {path} after {timeout:.0f} s "
                                   "(rank 0 never published one)")
            time.sleep(0.01)
    if world:
        ack = f"{path}.ack{rank}"
        with open(ack, "wb") as f:
            f.write(uid)                      # the id this rank is about to use
        missing = list(range(world))
        while missing:
            still = []
            for r in missing:
                try:
                    with open(f"{path}.ack{r}", "rb") as f:
                        seen = f.read()
                    if len(seen) < UNIQUE_ID_BYTES:
                        still.append(r)       # (being written)
                    elif seen != uid:
                        if os.stat(f"{path}.ack{r}").st_mtime >= _process_start() - 2.0:
                            raise RuntimeError(f"rank {rank}: rank {r} acknowledged a different communicator id at {path}")
                        still.append(r)       # a leftover of an earlier launch: rank r has not written its own yet
                except FileNotFoundError:
                    still.append(r)
            missing = still
            if missing and time.time() - t0 > timeout:
                raise TimeoutError(f"rank {rank}: ranks {missing} of {world} never acknowledged the communicator id at "
                                   f"{path} within {timeout:.0f} s -- not starting the collective initialisation")
            if missing:
                time.sleep(0.01)
    return uid, path


def init_from_env(device=None, transport=None, exchange=None):
    """The communicator of a launcher-started process, or None for a single process.

    Reads RANK, WORLD_SIZE, LOCAL_RANK (the GPU of this rank unless `device` is given).  `transport`:
    'rccl' (default; PMF_COMM_TRANSPORT overrides) or 'hostshm' for ranks that share one GPU (test build of
    the library only).  `exchange`: 'auto' | 'allreduce' | 'scatter_gather' for the contexts attached later
    (default: the library's, i.e. PMF_COMM_EXCHANGE or auto)."""
    from .engine import Context
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    rank = int(os.environ.get("RANK", "0"))
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    transport = transport or os.environ.get("PMF_COMM_TRANSPORT", "rccl")
    if transport == "hostshm":
        os.environ["PMF_COMM_TRANSPORT"] = "hostshm"      # selects the test build of the library (pmf_hip.load)
    uid, path = exchange_unique_id(rank, Context.comm_unique_id, world=world)
    comm = Comm(rank, world, device, uid, transport, exchange=exchange)
    comm.barrier()            # every rank is inside the communicator
    for leftover in ([path] if rank == 0 else []) + [f"{path}.ack{rank}"]:
        try:
            os.remove(leftover)
        except OSError:
            pass
    return comm


def distributed(comm):
    return comm is not None and comm.world > 1


# ---- one iteration per model -------------------------------------------------------
def _item_half_sweep(engine, comm, stats, width, accumulate, finalize):
    """External-collective form: accumulate -> all-reduce -> finalize over the item rows, sequenced
    by the host on a caller-owned statistics buffer.  With item row chunks (`set_row_chunks(ITEM, n)`,
    the same n on every rank) the three stages are pipelined: the statistics of chunk c travel
    (asynchronous all-reduce) while chunk c+1 is accumulated, and chunk c is finalised while later
    chunks still travel.  The arithmetic per row is that of the unchunked call."""
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


def _fused_item(comm):
    """The engine's one-call ITEM sweep is the whole distributed half-sweep: single process, or the
    communicator lives in the library."""
    return (not distributed(comm)) or getattr(comm, "in_library", False)


def gamma_iteration(engine, comm, stats_item, user_prior, item_prior):
    """Poisson MF / HPF.  `*_prior` = (shape_prior, rate_prior, hierarchical,
    hyper_shape, hyper_rate_prior) as taken by `gamma_sweep`.  `stats_item` is only used with an
    external collective: the [I x 2 x Kpad] buffer object with `.ptr` and `.tensor`."""
    engine.gamma_sweep(USER, *user_prior)
    if _fused_item(comm):
        engine.gamma_sweep(ITEM, *item_prior)
        return
    _item_half_sweep(engine, comm, stats_item, 2 * engine.kpad,
                     lambda: engine.gamma_accumulate(ITEM, stats_item.ptr),
                     lambda: engine.gamma_finalize(ITEM, stats_item.ptr, *item_prior))


def gaussian_iteration(engine, comm, stats_item, stats_bias, sigma2, eta_theta2, eta_beta2,
                       eta_bias2=None):
    """Gaussian MF (+ biases when eta_bias2 is given), order as
    gaussian_mf_cavi_bias.py:129-263."""
    fused = _fused_item(comm)
    engine.gauss_factor_sweep(USER, sigma2, eta_theta2)
    if fused:
        engine.gauss_factor_sweep(ITEM, sigma2, eta_beta2)
    else:
        _item_half_sweep(engine, comm, stats_item, engine.cov_stride + engine.kpad,
                         lambda: engine.gauss_factor_accumulate(ITEM, stats_item.ptr),
                         lambda: engine.gauss_factor_finalize(ITEM, stats_item.ptr, sigma2, eta_beta2))
    if eta_bias2 is None:
        return
    engine.gauss_bias_sweep(USER, sigma2, eta_bias2)
    if fused:
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
    if _fused_item(comm):
        engine.gauss_sgd_sweep(ITEM, lr, sigma2, eta_beta2, eta_bias2)
        return
    _item_half_sweep(engine, comm, stats_item, engine.sgd_stats_width,
                     lambda: engine.gauss_sgd_accumulate(ITEM, stats_item.ptr, lr, sigma2, eta_beta2, eta_bias2),
                     lambda: engine.gauss_sgd_finalize(ITEM, stats_item.ptr))


def default_item_chunks(world, message_bytes=0):
    """Item row chunks of a sharded run.  PMF_DIST_CHUNKS if set; else slices of about 256 MB, at most 32; at least 4
    for messages of 64 MB and more (at most a quarter of the all-reduce is then exposed), 2 for smaller ones (every
    extra chunk costs ~0.1 ms of launch tails at the C3 size -- tools/probe_comm_single.py -- which a 51 MB message
    cannot win back), 1 below 8 MB (latency-bound collectives).  1 = no pipelining."""
    if world <= 1:
        return 1
    if "PMF_DIST_CHUNKS" in os.environ:
        return max(1, int(os.environ["PMF_DIST_CHUNKS"]))
    if not message_bytes:
        return 4
    b = int(message_bytes)
    if b < (8 << 20):
        return 1
    return min(max(b // (256 << 20), 4 if b >= (64 << 20) else 2), 32)


def item_message_bytes(ctx, gaussian):
    """Bytes of the item statistics one iteration all-reduces (the factor half-sweep's message)."""
    width = (ctx.cov_stride + ctx.kpad) if gaussian else 2 * ctx.kpad
    return ctx.n_items * width * np.dtype(ctx.np_dtype).itemsize
