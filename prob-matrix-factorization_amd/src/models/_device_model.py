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

    def __init__(self, config, dtype=None, device=None):
        self.config = config
        self.n_users = None
        self.n_items = None
        self._dtype = engine_dtype(dtype)
        self._device = engine_device(device)
        self._ctx = None
        # structured per-iteration record next to the reference's stdout lines
        self.history_ = {"val_rmse": [], "val_macro_mae": [], "iterations": 0, "stopped_early": False,
                         "seconds": []}
        self._t_last = None

    # ---- dimensions (hpf_cavi.py:60-64) ----------------------------------
    def _infer_dimensions(self, train_df):
        self.n_users = int(train_df["u"].max()) + 1
        self.n_items = int(train_df["i"].max()) + 1
        if self.config.verbose:
            print(f"Inferred n_users={self.n_users}, n_items={self.n_items}")

    def _open_context(self, u, i, x):
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = pmf_hip.Context(self.n_users, self.n_items, self.config.n_factors,
                                    dtype=self._dtype, device=self._device)
        self._ctx.set_ratings(u, i, x)
        import time
        self._t_last = time.perf_counter()
        for key in ("val_rmse", "val_macro_mae", "seconds"):
            self.history_[key] = []
        self.history_["iterations"], self.history_["stopped_early"] = 0, False
        return self._ctx

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
            if len(vy) == 0:
                def empty():
                    print("Warning: No valid (u,i) pairs.")
                    return float("nan"), float("nan")
                return empty
        y = vy + offset if drop_unseen else vy
        ctx = self._ctx
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

    def _record(self, rmse_v, mae_v):
        self.history_["val_rmse"].append(rmse_v)
        self.history_["val_macro_mae"].append(mae_v)

    def top_k_items(self, user_ids, k=10):
        """Extension (no reference counterpart): the k highest-scoring items per
        user under `predict`'s score, ties to the lower item id.  Returns
        (items [n, k] int32, scores [n, k] float64)."""
        return self._need_ctx().topk_items(np.asarray(user_ids, dtype=int), int(k), use_bias=self._uses_bias)

    def close(self):
        """Release the device context (predict is unavailable afterwards)."""
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


__all__ = ["DeviceModel", "frame_arrays", "USER", "ITEM"]
