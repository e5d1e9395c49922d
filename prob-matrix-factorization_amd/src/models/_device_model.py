"""Shared host logic of the CAVI model classes: DataFrame -> arrays, device
context ownership, the validation monitor and the early-stop bookkeeping.

The numerical work of `fit` / `predict` is done by libpmf_hip.so through
`pmf_hip.Context`; nothing here computes a factor update on the host."""
from __future__ import annotations

import os

import numpy as np

import pmf_hip
from pmf_hip import ITEM, USER


def engine_dtype(explicit=None):
    """Device storage type: constructor argument, else $PMF_HIP_DTYPE, else f32."""
    d = explicit or os.environ.get("PMF_HIP_DTYPE", "f32")
    if d not in ("f32", "f64"):
        raise ValueError(f"dtype must be 'f32' or 'f64', got {d!r}")
    return d


def engine_device(explicit=None):
    if explicit is not None:
        return int(explicit)
    return int(os.environ.get("PMF_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def frame_arrays(df):
    """(u int64, i int64, rating float64) from a DataFrame[u, i, rating]
    (reference: hpf_cavi.py:113-115)."""
    return (df["u"].to_numpy(dtype=int), df["i"].to_numpy(dtype=int),
            df["rating"].to_numpy(dtype=float))


class DeviceModel:
    """Base of the four CAVI classes.  Subclasses set `_uses_bias`."""

    _uses_bias = False
    _gaussian = False     # Gaussian models exchange [I x (Kp + Kpad)] statistics, the others [I x 2 Kpad]

    def __init__(self, config, dtype=None, device=None, comm=None, presharded=False):
        """`comm`: a `pmf_hip.dist.Comm` (the RCCL communicator inside libpmf_hip.so, one rank
        per GPU; `pmf_hip.dist.init_from_env()`) makes `fit` a user-sharded multi-GPU run: every
        rank passes the SAME frames, keeps the ratings of its user range, item statistics are
        all-reduced once per item half-sweep by the library, and every rank ends with the full
        factor matrices.

        `presharded=True` (with `comm`): `fit` is given only THIS RANK's training rows instead -- global
        user ids, rank r holding a contiguous id range that lies above rank r-1's -- so no rank ever
        holds the full frame (12 GB per rank at BASELINE config C4); dimensions and the ranges are
        agreed by small all-reduces.  `presharded=bounds` (world + 1 ascending user ids, bounds[0] = 0) states the
        ranges instead of having them derived from the rows: rank r owns users [bounds[r], bounds[r + 1]) -- the
        way to say where users WITHOUT training rows belong, which decides the rank that must be handed their
        validation rows (derived ranges give such users to the next rank above that has training rows).
        Rows outside their rank's range are refused on every rank alike.  Either way the initial state is drawn
        with the reference's RNG sequence and a rank keeps only its own users' rows of it."""
        self.config = config
        self._comm = comm if (comm is not None and comm.world > 1) else None
        explicit = presharded is not None and not isinstance(presharded, (bool, np.bool_))
        self._given_bounds = np.asarray(presharded, dtype=np.int64) if explicit else None
        self._presharded = (explicit or bool(presharded)) and self._comm is not None
        self._bounds = None
        self.n_users = None
        self.n_items = None
        self._dtype = engine_dtype(dtype)
        self._device = engine_device(device)
        self._ctx = None
        self._shard_ctx = None
        # structured per-iteration record next to the reference's stdout lines
        self.history_ = {"val_rmse": [], "val_macro_mae": [], "iterations": 0, "stopped_early": False,
                         "seconds": []}
        self._t_last = None

    # ---- dimensions (hpf_cavi.py:60-64) ----------------------------------
    def _infer_dimensions(self, train_df):
        self.n_users = int(train_df["u"].max()) + 1 if len(train_df) else 0
        self.n_items = int(train_df["i"].max()) + 1 if len(train_df) else 0
        if self._presharded:   # every rank sees its own rows only: the dimensions are the maxima over the ranks
            dims = self._comm.all_reduce_host([float(self.n_users), float(self.n_items)], op="max")
            self.n_users, self.n_items = int(dims[0]), int(dims[1])
        if self.config.verbose:
            print(f"Inferred n_users={self.n_users}, n_items={self.n_items}")
        self._plan_shards(train_df)

    def _plan_shards(self, train_df):
        """User ranges of a sharded fit, known before the initial state is drawn."""
        self._bounds = None
        if self._comm is None:
            return
        from pmf_hip import dist as pdist
        comm = self._comm
        if not self._presharded:
            self._bounds = pdist.shard_bounds(train_df["u"].to_numpy(dtype=int), self.n_users, comm.world)
            return
        u = train_df["u"].to_numpy(dtype=int)
        if self._given_bounds is not None:
            b = self._given_bounds
            if b.shape != (comm.world + 1,) or b[0] != 0 or (np.diff(b) < 0).any():
                raise ValueError(f"presharded bounds must be {comm.world + 1} ascending user ids starting at 0, got {b}")
            b = np.minimum(b, self.n_users)        # n_users = highest training id + 1, as in the reference
            b[-1] = self.n_users
            lo, hi = int(b[comm.rank]), int(b[comm.rank + 1])
            stray = comm.all_reduce_host([float(((u < lo) | (u >= hi)).sum())])[0]
            if stray:
                raise ValueError(f"presharded fit: {int(stray)} training row(s) lie outside their rank's user range {b}")
            empty = [r for r in range(comm.world) if b[r + 1] <= b[r]]
            if empty:
                raise ValueError(f"presharded fit: rank(s) {empty} hold no user range in {b} (n_users = {self.n_users})")
            self._bounds = b
            return
        top = np.zeros(comm.world)
        low = np.full(comm.world, 0.0)
        top[comm.rank] = float(u.max()) + 1 if len(u) else 0.0
        low[comm.rank] = float(u.min()) if len(u) else 0.0
        top, low = comm.all_reduce_host(top), comm.all_reduce_host(low)
        top = np.maximum.accumulate(top)                    # a rank without rows owns an empty range
        top[-1] = self.n_users
        bounds = np.concatenate([[0.0], top]).astype(np.int64)
        for r in range(comm.world):
            if low[r] < bounds[r] and top[r] > bounds[r]:
                raise ValueError(f"presharded fit: rank {r}'s user ids start at {int(low[r])}, inside rank {r - 1}'s range "
                                 f"(< {int(bounds[r])}); every rank must hold a contiguous user-id range above the previous rank's")
        empty = [r for r in range(comm.world) if bounds[r + 1] <= bounds[r]]
        if empty:      # the same on every rank (the bounds come from all-reduced values): all ranks raise together
            raise ValueError(f"presharded fit: rank(s) {empty} hold no user range (no training rows, and no user ids left "
                             f"above the previous rank's); use fewer ranks or give every rank at least one user")
        self._bounds = bounds

    def _user_rows(self, draw):
        """`draw(n)` -> the next n rows of a user-side initial array from the model's RNG.  Unsharded: all
        rows.  Sharded: the SAME stream is consumed (so every rank's state is the reference's), in blocks,
        and only this rank's rows are kept -- no rank materialises a full user-side array."""
        if self._comm is None:
            return draw(self.n_users)
        lo, hi = int(self._bounds[self._comm.rank]), int(self._bounds[self._comm.rank + 1])
        block = 1 << 18
        for at in range(0, lo, block):
            draw(min(block, lo - at))
        mine = draw(hi - lo)
        for at in range(hi, self.n_users, block):
            draw(min(block, self.n_users - at))
        return mine

    def _open_context(self, u, i, x):
        if self._ctx is not None:
            self._ctx.close()
        n_local = self.n_users
        from pmf_hip import dist as pdist
        if self._comm is not None:
            lo, hi = int(self._bounds[self._comm.rank]), int(self._bounds[self._comm.rank + 1])
            if self._presharded:
                u = np.asarray(u) - lo              # the frame IS the shard: ids become local to the range
            else:
                u, i, x = pdist.take_shard(u, i, x, self._bounds, self._comm.rank)
            n_local = hi - lo
        self._ctx = pmf_hip.Context(n_local, self.n_items, self.config.n_factors,
                                    dtype=self._dtype, device=self._device)
        if self._comm is not None:
            # from here on the library runs this context's ITEM half-sweeps as
            # accumulate -> all-reduce -> finalize, pipelined over the item row chunks
            self._comm.attach(self._ctx)
            self._ctx.set_row_chunks(pmf_hip.ITEM, pdist.default_item_chunks(
                self._comm.world, pdist.item_message_bytes(self._ctx, self._gaussian)))
        self._ctx.set_ratings(u, i, x)
        import time
        self._t_last = time.perf_counter()
        for key in ("val_rmse", "val_macro_mae", "seconds"):
            self.history_[key] = []
        self.history_["iterations"], self.history_["stopped_early"] = 0, False
        return self._ctx

    def _run_iteration(self, issue):
        """`issue()` makes the sweep calls of one iteration (asynchronous launches on the context's stream)."""
        return issue()

    # ---- multi-GPU helpers --------------------------------------------------
    def _mine(self, user_array):
        """This rank's rows of a user-side array: arrays made by `_user_rows` (or any array that already has
        this rank's row count) pass through, full-size ones are sliced."""
        if self._comm is None:
            return user_array
        lo, hi = int(self._bounds[self._comm.rank]), int(self._bounds[self._comm.rank + 1])
        if len(user_array) == hi - lo and hi - lo != self.n_users:
            return user_array
        return user_array[lo:hi]

    def _user_array(self, array_id, ctx=None):
        """A user-side state array for ALL users on the host: this context's rows, or -- after a
        sharded fit -- every rank's rows in user order (a collective: every rank must call it)."""
        ctx = ctx or self._ctx
        if self._comm is None:
            return ctx.get_array(USER, array_id)
        return ctx.gather_user_rows(array_id, self._bounds)

    def _finish_sharded(self, arrays):
        """After a sharded fit every rank holds the full factors on the host; give it a
        full-size context too, so predict / evaluate / top-k work as after a single-GPU fit.
        `arrays`: [(side, array_id, host_array)].  The training shard stays in `_shard_ctx`."""
        self._shard_ctx = self._ctx
        full = pmf_hip.Context(self.n_users, self.n_items, self.config.n_factors, dtype=self._dtype,
                               device=self._device)
        for side, array_id, host in arrays:
            full.set_array(side, array_id, host)
        self._ctx = full

    def _need_ctx(self):
        if self._ctx is None:
            raise RuntimeError(f"{type(self).__name__} has not been fitted")
        return self._ctx

    # ---- validation monitor ----------------------------------------------
    def _monitor_setup(self, val_df, offset=0.0, drop_unseen=False):
        """Put the validation pairs on the device once.  Returns a callable
        giving (rmse, macro_mae) for the current device state, or None."""
        if val_df is None:
            return None
        vu, vi, vy = frame_arrays(val_df)
        if drop_unseen:  # gaussian_mf_cavi_bias.py:323-331
            keep = (vu < self.n_users) & (vi < self.n_items)
            vu, vi, vy = vu[keep], vi[keep], vy[keep]
            if len(vy) == 0 and not (self._comm is not None and self._presharded):
                # (presharded: this rank's rows are not the whole validation set; an empty share must still take
                # part in the collectives of _sharded_monitor, which reports the globally empty case itself)
                def empty():
                    print("Warning: No valid (u,i) pairs.")
                    return float("nan"), float("nan")
                return empty
        y = vy + offset if drop_unseen else vy
        ctx = self._ctx
        if self._comm is not None:
            return self._sharded_monitor(vu, vi, y, offset, warn_empty=drop_unseen)
        if ctx.eval_set(vu, vi, y):
            return lambda: ctx.eval_run(self._uses_bias, offset)
        # too many distinct labels for the fused reduction: device predict + host metrics
        from src.evaluation.metrics import macro_mae, rmse

        def slow():
            p = ctx.predict(vu, vi, self._uses_bias, offset)
            return float(rmse(y, p)), float(macro_mae(y, p))
        return slow

    def _tick(self, it):
        """Iteration `it` has been issued: count it and note the wall time since the
        previous tick (kernels are asynchronous; with a validation monitor the tick
        follows its synchronising read-back, so the figure is the true step time)."""
        import time
        now = time.perf_counter()
        if self._t_last is not None:
            self.history_["seconds"].append(now - self._t_last)
        self._t_last = now
        self.history_["iterations"] = it

    def _sharded_monitor(self, vu, vi, y, offset, warn_empty=False):
        """Every rank scores the validation pairs of its own users (pairs with an unseen
        user go to the last rank, where the id stays out of range and predicts 0); the
        additive sums are all-reduced."""
        from pmf_hip import MAX_LABELS
        comm, ctx = self._comm, self._ctx
        labels = np.unique(y)
        if self._presharded:
            # every rank holds its own validation rows: the label set of metrics.macro_mae is their union
            slots = np.zeros(comm.world * (MAX_LABELS + 1))
            base = comm.rank * (MAX_LABELS + 1)
            slots[base] = len(labels)
            slots[base + 1:base + 1 + min(len(labels), MAX_LABELS)] = labels[:MAX_LABELS]
            slots = comm.all_reduce_host(slots).reshape(comm.world, MAX_LABELS + 1)
            if (slots[:, 0] > MAX_LABELS).any():
                raise NotImplementedError("sharded validation supports at most %d distinct ratings" % MAX_LABELS)
            labels = np.unique(np.concatenate([row[1:1 + int(row[0])] for row in slots]))
        if len(labels) > MAX_LABELS:
            raise NotImplementedError("sharded validation supports at most %d distinct ratings" % MAX_LABELS)
        lo, hi = int(self._bounds[comm.rank]), int(self._bounds[comm.rank + 1])
        last = comm.rank == comm.world - 1
        mine = (vu >= lo) & ((vu < hi) | last)
        if self._presharded:
            # a row handed to the wrong rank would silently drop out of the metric: refuse it, on every rank alike
            stray = comm.all_reduce_host([float((~mine).sum())])[0]
            if stray:
                raise ValueError(f"presharded fit: {int(stray)} validation row(s) were given to a rank that does not own "
                                 f"their user (ranges {self._bounds.tolist()}; ids >= n_users go to the last rank); pass "
                                 f"`presharded=bounds` to state the ranges")
        lu = np.where(vu[mine] < self.n_users, vu[mine] - lo, np.iinfo(np.int32).max)
        have = ctx.eval_set(lu, vi[mine], y[mine], labels=labels) if mine.any() else False

        def run():
            sums = comm.all_reduce_host(ctx.eval_sums(self._uses_bias, offset) if have else np.zeros(2 + 2 * MAX_LABELS))
            if sums[0] == 0:      # no rank holds a scorable pair
                if warn_empty:    # gaussian_mf_cavi_bias.py:326-328
                    print("Warning: No valid (u,i) pairs.")
                return float("nan"), float("nan")
            return ctx.metrics_from_sums(sums)
        return run

    def _record(self, rmse_v, mae_v):
        self.history_["val_rmse"].append(rmse_v)
        self.history_["val_macro_mae"].append(mae_v)

    def top_k_items(self, user_ids, k=10):
        """Extension (no reference counterpart): the k highest-scoring items per
        user under `predict`'s score, ties to the lower item id.  Returns
        (items [n, k] int32, scores [n, k] float64)."""
        return self._need_ctx().topk_items(np.asarray(user_ids, dtype=int), int(k), use_bias=self._uses_bias)

    def close(self):
        """Release the device context(s) (predict is unavailable afterwards)."""
        for name in ("_ctx", "_shard_ctx"):
            ctx = getattr(self, name, None)
            if ctx is not None:
                ctx.close()
                setattr(self, name, None)


__all__ = ["DeviceModel", "frame_arrays", "USER", "ITEM"]
