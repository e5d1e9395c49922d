"""Poisson matrix factorisation, CAVI on observed entries -- MI355X engine.

Drop-in for the reference's `src/models/poisson_mf_cavi.py`: same config
dataclass, same `fit` / `predict` / `evaluate_*` signatures, same attributes
(`a_theta, b_theta, a_beta, b_beta, E_theta, E_beta`, float64 NumPy), same
verbose output.  The two half-sweeps per iteration run as HIP kernels
(`pmf_gamma_sweep`)."""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from src.evaluation.metrics import macro_mae, rmse
from src.models._device_model import ITEM, USER, DeviceModel, frame_arrays
from pmf_hip import ARR_FACTOR, ARR_RATE, ARR_SHAPE, dist as pdist


@dataclass
class PoissonMFCAVIConfig:
    n_factors: int = 20
    a0: float = 0.3
    b0: float = 1.0
    max_iter: int = 100
    tol: Optional[float] = 1e-4
    random_state: int = 42
    verbose: bool = True


class PoissonMFCAVI(DeviceModel):
    """x_ij ~ Poisson(theta_i . beta_j), theta, beta ~ Gamma(a0, b0)."""

    def __init__(self, config: PoissonMFCAVIConfig, dtype=None, device=None, comm=None, presharded=False):
        super().__init__(config, dtype, device, comm, presharded)
        self.a_theta = self.b_theta = self.a_beta = self.b_beta = None
        self.E_theta = self.E_beta = None

    def _initialize_variational_params(self):
        """Reference draw order (poisson_mf_cavi.py:50-71): user shapes, item shapes."""
        cfg = self.config
        rng = np.random.default_rng(cfg.random_state)
        self.a_theta = cfg.a0 + self._user_rows(lambda n: rng.gamma(1.0, 0.1, size=(n, cfg.n_factors)))
        self.a_beta = cfg.a0 + rng.gamma(1.0, 0.1, size=(self.n_items, cfg.n_factors))
        self.b_theta = np.full(self.a_theta.shape, float(cfg.b0))
        self.b_beta = np.full((self.n_items, cfg.n_factors), float(cfg.b0))
        self.E_theta = self.a_theta / self.b_theta
        self.E_beta = self.a_beta / self.b_beta

    def _pull_state(self):
        ctx, g = self._ctx, self._user_array
        self.a_theta, self.b_theta = g(ARR_SHAPE), g(ARR_RATE)
        self.a_beta, self.b_beta = ctx.get_array(ITEM, ARR_SHAPE), ctx.get_array(ITEM, ARR_RATE)
        self.E_theta, self.E_beta = g(ARR_FACTOR), ctx.get_array(ITEM, ARR_FACTOR)
        if self._comm is not None:
            self._finish_sharded([(USER, ARR_FACTOR, self.E_theta), (ITEM, ARR_FACTOR, self.E_beta)])

    def fit(self, train_df, val_df=None):
        cfg = self.config
        self._infer_dimensions(train_df)
        self._initialize_variational_params()
        u, i, x = frame_arrays(train_df)
        ctx = self._open_context(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, self._mine(self.E_theta))
        ctx.set_array(ITEM, ARR_FACTOR, self.E_beta)
        prior = (cfg.a0, cfg.b0, False, 0.0, 0.0)
        monitor = self._monitor_setup(val_df)
        previous = None
        for it in range(1, cfg.max_iter + 1):
            if cfg.verbose:
                print(f"\nCAVI iteration {it}/{cfg.max_iter}")
            # users (poisson_mf_cavi.py:135-170) then items (:173-200); with a Comm the library runs
            # the item half-sweep as accumulate -> all-reduce -> finalize
            self._run_iteration(lambda: pdist.gamma_iteration(ctx, self._comm, None, prior, prior))
            self._tick(it)
            if monitor is None:
                continue
            val_rmse, val_macro_mae = monitor()
            self._record(val_rmse, val_macro_mae)
            if cfg.verbose:
                print(f"Validation RMSE: {val_rmse:.4f} | MacroMAE: {val_macro_mae:.4f}")
            if previous is not None:
                improvement = previous - val_rmse
                if cfg.verbose:
                    print(f"Improvement: {improvement:.6f}")
                if cfg.tol is not None and improvement < cfg.tol:  # poisson_mf_cavi.py:213
                    if cfg.verbose:
                        print("Early stopping.")
                    self.history_["stopped_early"] = True
                    break
            previous = val_rmse
        if self.history_["iterations"] > 0:
            self._pull_state()
        return self

    def predict(self, user_ids, item_ids):
        return self._need_ctx().predict(np.asarray(user_ids, dtype=int), np.asarray(item_ids, dtype=int))

    def evaluate_rmse(self, df):
        return rmse(df["rating"].to_numpy(), self.predict(df["u"].to_numpy(), df["i"].to_numpy()))

    def evaluate_macro_mae(self, df):
        return macro_mae(df["rating"].to_numpy(), self.predict(df["u"].to_numpy(), df["i"].to_numpy()))
