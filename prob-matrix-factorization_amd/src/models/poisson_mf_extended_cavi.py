"""Extended Poisson MF with per-user / per-item scalar factors,
x_ij ~ Poisson(phi_u psi_i theta_u . beta_i) -- MI355X engine.

Drop-in for the reference's `src/models/poisson_mf_extended_cavi.py`: same
config dataclass, `fit` / `predict` / `evaluate_rmse`, attributes
`a_theta, b_theta, a_beta, b_beta, a_phi, b_phi, a_psi, b_psi, E_theta, E_beta,
E_phi, E_psi`.  Each half-sweep (factor update, then the scalar update that uses
the row's NEW factors) is one `pmf_gamma_ext_sweep`."""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from src.evaluation.metrics import rmse
from src.models._device_model import ITEM, USER, DeviceModel, frame_arrays
from pmf_hip import (ARR_FACTOR, ARR_RATE, ARR_SCALE, ARR_SCALE_RATE, ARR_SCALE_SHAPE, ARR_SHAPE,
                     PREDICT_SCALE)


@dataclass
class PoissonMFExtendedCAVIConfig:
    n_factors: int = 20
    a0: float = 0.3
    b0: float = 1.0
    max_iter: int = 100
    tol: Optional[float] = 1e-4
    random_state: int = 42
    verbose: bool = True


class PoissonMFExtendedCAVI(DeviceModel):
    _uses_bias = PREDICT_SCALE   # predict / monitor flag: multiply by E_phi[u] E_psi[i]

    def __init__(self, config: PoissonMFExtendedCAVIConfig, dtype=None, device=None):
        super().__init__(config, dtype, device)
        for name in ("theta", "beta", "phi", "psi"):
            setattr(self, f"a_{name}", None)
            setattr(self, f"b_{name}", None)
            setattr(self, f"E_{name}", None)

    def _initialize_variational_params(self):
        """Reference draw order (poisson_mf_extended_cavi.py:57-75): a_theta, a_beta, a_phi, a_psi."""
        cfg = self.config
        K, N, M = cfg.n_factors, self.n_users, self.n_items
        rng = np.random.default_rng(cfg.random_state)
        self.a_theta = cfg.a0 + rng.gamma(1.0, 0.1, size=(N, K))
        self.a_beta = cfg.a0 + rng.gamma(1.0, 0.1, size=(M, K))
        self.a_phi = cfg.a0 + rng.gamma(1.0, 0.1, size=N)
        self.a_psi = cfg.a0 + rng.gamma(1.0, 0.1, size=M)
        self.b_theta, self.b_beta = np.full((N, K), float(cfg.b0)), np.full((M, K), float(cfg.b0))
        self.b_phi, self.b_psi = np.full(N, float(cfg.b0)), np.full(M, float(cfg.b0))
        for name in ("theta", "beta", "phi", "psi"):
            setattr(self, f"E_{name}", getattr(self, f"a_{name}") / getattr(self, f"b_{name}"))

    def _pull_state(self):
        ctx = self._ctx
        for side, factor, scalar in ((USER, "theta", "phi"), (ITEM, "beta", "psi")):
            setattr(self, f"a_{factor}", ctx.get_array(side, ARR_SHAPE))
            setattr(self, f"b_{factor}", ctx.get_array(side, ARR_RATE))
            setattr(self, f"E_{factor}", ctx.get_array(side, ARR_FACTOR))
            setattr(self, f"a_{scalar}", ctx.get_array(side, ARR_SCALE_SHAPE))
            setattr(self, f"b_{scalar}", ctx.get_array(side, ARR_SCALE_RATE))
            setattr(self, f"E_{scalar}", ctx.get_array(side, ARR_SCALE))

    def fit(self, train_df, val_df=None):
        cfg = self.config
        self._infer_dimensions(train_df)
        self._initialize_variational_params()
        u, i, x = frame_arrays(train_df)
        ctx = self._open_context(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, self.E_theta)
        ctx.set_array(ITEM, ARR_FACTOR, self.E_beta)
        ctx.set_array(USER, ARR_SCALE, self.E_phi)
        ctx.set_array(ITEM, ARR_SCALE, self.E_psi)
        monitor = self._monitor_setup(val_df)
        previous = None
        for it in range(1, cfg.max_iter + 1):
            if cfg.verbose:
                print(f"\nCAVI iteration {it}/{cfg.max_iter}")
            self._run_iteration(lambda: (ctx.gamma_ext_sweep(USER, cfg.a0, cfg.b0),     # poisson_mf_extended_cavi.py:108-160
                                         ctx.gamma_ext_sweep(ITEM, cfg.a0, cfg.b0)))    # poisson_mf_extended_cavi.py:163-215
            self._tick(it)
            if monitor is None:
                continue
            val_rmse, _ = monitor()
            self._record(val_rmse, float("nan"))
            if cfg.verbose:
                print(f"Validation RMSE: {val_rmse:.4f}")
            if previous is not None:
                improvement = previous - val_rmse
                if cfg.verbose:
                    print(f"Improvement: {improvement:.6f}")
                if cfg.tol is not None and improvement < cfg.tol:
                    if cfg.verbose:
                        print("Early stopping.")
                    self.history_["stopped_early"] = True
                    break
            previous = val_rmse
        if self.history_["iterations"] > 0:
            self._pull_state()
        return self

    def predict(self, user_ids, item_ids):
        return self._need_ctx().predict(np.asarray(user_ids, dtype=int), np.asarray(item_ids, dtype=int),
                                        use_bias=PREDICT_SCALE)

    def evaluate_rmse(self, df):
        return rmse(df["rating"].to_numpy(), self.predict(df["u"].to_numpy(), df["i"].to_numpy()))
