"""Hierarchical Poisson factorisation, CAVI on observed entries -- MI355X engine.

Drop-in for the reference's `src/models/hpf_cavi.py`: same config dataclass,
methods and attributes (`gamma_a_theta, gamma_b_theta, gamma_a_beta,
gamma_b_beta, gamma_a_xi (scalar), gamma_b_xi, gamma_a_eta (scalar),
gamma_b_eta, E_theta, E_beta, E_xi, E_eta`).  Each half-sweep, including the
xi / eta update that follows it, is one `pmf_gamma_sweep(hierarchical=1)`."""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from src.evaluation.metrics import macro_mae, rmse
from src.models._device_model import ITEM, USER, DeviceModel, frame_arrays
from pmf_hip import ARR_FACTOR, ARR_HYPER_RATE, ARR_PRIOR_RATE, ARR_RATE, ARR_SHAPE, dist as pdist


@dataclass
class HPF_CAVI_Config:
    n_factors: int = 20
    a: float = 0.3
    a_prime: float = 0.3
    b_prime: float = 1.0
    c: float = 0.3
    c_prime: float = 0.3
    d_prime: float = 1.0
    max_iter: int = 100
    tol: Optional[float] = 1e-4
    random_state: int = 42
    verbose: bool = True


class HPF_CAVI(DeviceModel):
    """x_ui ~ Poisson(theta_u . beta_i); theta_uk ~ Gamma(a, xi_u), xi_u ~ Gamma(a', b');
    beta_ik ~ Gamma(c, eta_i), eta_i ~ Gamma(c', d')."""

    def __init__(self, config: HPF_CAVI_Config, dtype=None, device=None, comm=None, presharded=False):
        super().__init__(config, dtype, device, comm, presharded)
        self.gamma_a_theta = self.gamma_b_theta = None
        self.gamma_a_beta = self.gamma_b_beta = None
        self.gamma_a_xi = self.gamma_b_xi = None
        self.gamma_a_eta = self.gamma_b_eta = None
        self.E_theta = self.E_beta = self.E_xi = self.E_eta = None

    def _initialize(self):
        """Reference draw order (hpf_cavi.py:66-89): a_theta, b_theta, a_beta, b_beta."""
        cfg = self.config
        K, N, M = cfg.n_factors, self.n_users, self.n_items
        rng = np.random.default_rng(cfg.random_state)
        users = lambda n: rng.gamma(1.0, 0.1, size=(n, K))      # (a sharded fit keeps this rank's rows only)
        noise = [self._user_rows(users), self._user_rows(users), rng.gamma(1.0, 0.1, size=(M, K)),
                 rng.gamma(1.0, 0.1, size=(M, K))]
        self.gamma_a_theta = cfg.a + noise[0]
        self.gamma_b_theta = cfg.b_prime + noise[1]
        self.gamma_a_beta = cfg.c + noise[2]
        self.gamma_b_beta = cfg.d_prime + noise[3]
        self.gamma_a_xi = cfg.a_prime + K * cfg.a
        self.gamma_b_xi = cfg.b_prime * np.ones(len(noise[0]))
        self.gamma_a_eta = cfg.c_prime + K * cfg.c
        self.gamma_b_eta = cfg.d_prime * np.ones(M)
        self._update_expectations()

    def _update_expectations(self):
        self.E_theta = self.gamma_a_theta / self.gamma_b_theta
        self.E_beta = self.gamma_a_beta / self.gamma_b_beta
        self.E_xi = self.gamma_a_xi / self.gamma_b_xi
        self.E_eta = self.gamma_a_eta / self.gamma_b_eta

    def _pull_state(self):
        ctx, g = self._ctx, self._user_array
        self.gamma_a_theta, self.gamma_b_theta = g(ARR_SHAPE), g(ARR_RATE)
        self.gamma_a_beta, self.gamma_b_beta = ctx.get_array(ITEM, ARR_SHAPE), ctx.get_array(ITEM, ARR_RATE)
        self.E_theta, self.E_beta = g(ARR_FACTOR), ctx.get_array(ITEM, ARR_FACTOR)
        self.gamma_b_xi, self.gamma_b_eta = g(ARR_HYPER_RATE), ctx.get_array(ITEM, ARR_HYPER_RATE)
        self.E_xi, self.E_eta = g(ARR_PRIOR_RATE), ctx.get_array(ITEM, ARR_PRIOR_RATE)
        if self._comm is not None:
            self._finish_sharded([(USER, ARR_FACTOR, self.E_theta), (ITEM, ARR_FACTOR, self.E_beta)])

    def fit(self, train_df, val_df=None):
        cfg = self.config
        self._infer_dimensions(train_df)
        self._initialize()
        u, i, x = frame_arrays(train_df)
        ctx = self._open_context(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, self._mine(self.E_theta))
        ctx.set_array(ITEM, ARR_FACTOR, self.E_beta)
        ctx.set_array(USER, ARR_PRIOR_RATE, self._mine(self.E_xi))
        ctx.set_array(ITEM, ARR_PRIOR_RATE, self.E_eta)
        user_prior = (cfg.a, 0.0, True, self.gamma_a_xi, cfg.b_prime)
        item_prior = (cfg.c, 0.0, True, self.gamma_a_eta, cfg.d_prime)
        monitor = self._monitor_setup(val_df)
        previous = None
        for it in range(1, cfg.max_iter + 1):
            if cfg.verbose:
                print(f"\nHPF_CAVI iteration {it}/{cfg.max_iter}")
            # theta then xi (hpf_cavi.py:126-159); beta then eta (hpf_cavi.py:162-193)
            self._run_iteration(lambda: pdist.gamma_iteration(ctx, self._comm, None, user_prior, item_prior))
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
                if cfg.tol is not None and improvement < cfg.tol:  # hpf_cavi.py:207
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
