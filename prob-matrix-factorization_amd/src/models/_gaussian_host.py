"""Host logic shared by the two Gaussian CAVI classes (with / without biases).

Iteration order, early-stop rule and verbose lines follow the reference
(`gaussian_mf_cavi_bias.py:91-286`, bias-free twin `gaussian_mf_cavi.py:81-200`);
each half-sweep is one C-ABI call (`pmf_gauss_factor_sweep`,
`pmf_gauss_bias_sweep`)."""
import numpy as np

from src.evaluation.metrics import macro_mae, rmse
from src.models._device_model import ITEM, USER, DeviceModel, frame_arrays
from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, dist as pdist


class GaussianHost(DeviceModel):
    def __init__(self, config, dtype=None, device=None, comm=None, presharded=False):
        super().__init__(config, dtype, device, comm, presharded)
        self.m_theta = self.m_beta = None
        self._V_theta = self._V_beta = None
        self.global_mean = 0.0
        if self._uses_bias:
            self.m_user_bias = self.m_item_bias = None

    # The covariance stacks are rows x K x K float64 on the host (32 GB at
    # 1M x 64 x 64): they stay packed on the device and are materialised only
    # when the attribute is read.
    def _train_ctx(self):
        return getattr(self, "_shard_ctx", None) or self._ctx

    @property
    def V_theta(self):
        """(n_users, K, K) float64.  After a SHARDED fit this is only THIS RANK's user range
        (rows `user_range`): an attribute read must not hide a collective (a rank-0-only read
        would deadlock the job, and the full stack is U x K x K x 8 bytes on every rank).  All ranks
        together call `gather_V_theta()` for the full stack."""
        if self._V_theta is None and self._ctx is not None:
            self._V_theta = self._train_ctx().get_array(USER, ARR_COV)
        return self._V_theta

    @property
    def user_range(self):
        """[lo, hi) of the users this rank trained (all users when not sharded)."""
        if self._comm is None or self._bounds is None:
            return 0, self.n_users
        return int(self._bounds[self._comm.rank]), int(self._bounds[self._comm.rank + 1])

    def gather_V_theta(self):
        """Collective after a sharded fit: every rank gets the full (n_users, K, K) covariance stack
        (broadcast in 64 MB row blocks).  Equals `V_theta` when not sharded."""
        if self._comm is None:
            return self.V_theta
        return self._user_array(ARR_COV, self._train_ctx())

    @V_theta.setter
    def V_theta(self, value):
        self._V_theta = value

    def V_theta_rows(self, user_ids):
        """`V_theta[user_ids]` (len x K x K float64) read straight from the device's packed storage
        (`pmf_get_array_rows`) -- what the reference's row indexing does (gaussian_mf_cavi_bias.py:157-162)
        without materialising the whole stack.  After a sharded fit the ids must lie in `user_range`."""
        ids = np.asarray(user_ids, dtype=np.int64).reshape(-1)
        if self._V_theta is not None:
            return np.asarray(self._V_theta)[ids - self.user_range[0]]
        lo, hi = self.user_range
        if len(ids) and (ids.min() < lo or ids.max() >= hi):
            raise IndexError(f"V_theta_rows: user ids outside this rank's range [{lo}, {hi})")
        return self._train_ctx().get_array_rows(USER, ARR_COV, ids - lo)

    def V_beta_rows(self, item_ids):
        """`V_beta[item_ids]` (len x K x K float64), same idea."""
        ids = np.asarray(item_ids, dtype=np.int64).reshape(-1)
        if self._V_beta is not None:
            return np.asarray(self._V_beta)[ids]
        return self._train_ctx().get_array_rows(ITEM, ARR_COV, ids)

    @property
    def V_beta(self):
        if self._V_beta is None and self._ctx is not None:
            self._V_beta = self._train_ctx().get_array(ITEM, ARR_COV)
        return self._V_beta

    @V_beta.setter
    def V_beta(self, value):
        self._V_beta = value

    def _initialize_variational_params(self):
        """Reference draw order (gaussian_mf_cavi_bias.py:52-67): user means, item means."""
        K = self.config.n_factors
        rng = np.random.default_rng(self.config.random_state)
        self.m_theta = 0.1 * self._user_rows(lambda n: rng.standard_normal((n, K)))
        self.m_beta = 0.1 * rng.standard_normal((self.n_items, K))
        if self._uses_bias:
            self.m_user_bias = np.zeros(len(self.m_theta))
            self.m_item_bias = np.zeros(self.n_items)
        self._V_theta = self._V_beta = None

    def _pull_state(self):
        ctx, g = self._ctx, self._user_array
        self.m_theta, self.m_beta = g(ARR_FACTOR), ctx.get_array(ITEM, ARR_FACTOR)
        arrays = [(USER, ARR_FACTOR, self.m_theta), (ITEM, ARR_FACTOR, self.m_beta)]
        if self._uses_bias:
            self.m_user_bias, self.m_item_bias = g(ARR_BIAS), ctx.get_array(ITEM, ARR_BIAS)
            arrays += [(USER, ARR_BIAS, self.m_user_bias), (ITEM, ARR_BIAS, self.m_item_bias)]
        self._V_theta = self._V_beta = None
        if self._comm is not None:
            self._finish_sharded(arrays)

    # ---- what one iteration is (the MAP / gradient subclass overrides these two) ----
    _iteration_label = "CAVI iteration"

    def _prepare(self, ctx):
        ctx.set_cov_identity(USER, 1.0)
        ctx.set_cov_identity(ITEM, 1.0)

    @staticmethod
    def _should_stop(improvement, tol):
        return improvement >= 0 and improvement < tol   # gaussian_mf_cavi_bias.py:279

    def _iterate(self, ctx):
        cfg = self.config
        pdist.gaussian_iteration(ctx, self._comm, None, None, cfg.sigma2, cfg.eta_theta2,
                                 cfg.eta_beta2, cfg.eta_bias2 if self._uses_bias else None)

    def fit(self, train_df, val_df=None, global_mean=0.0):
        cfg = self.config
        self.global_mean = global_mean
        self._infer_dimensions(train_df)
        self._initialize_variational_params()
        u, i, x = frame_arrays(train_df)
        ctx = self._open_context(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, self._mine(self.m_theta))
        ctx.set_array(ITEM, ARR_FACTOR, self.m_beta)
        if self._uses_bias:
            ctx.set_array(USER, ARR_BIAS, self._mine(self.m_user_bias))
            ctx.set_array(ITEM, ARR_BIAS, self.m_item_bias)
        self._prepare(ctx)
        monitor = self._monitor_setup(val_df, offset=global_mean, drop_unseen=True)
        previous = None
        for it in range(1, cfg.max_iter + 1):
            if cfg.verbose:
                print(f"\n{self._iteration_label} {it}/{cfg.max_iter}")
            self._run_iteration(lambda: self._iterate(ctx))
            self._tick(it)
            if monitor is None:
                continue
            val_rmse, val_macro_mae = monitor()
            self._record(val_rmse, val_macro_mae)
            if cfg.verbose:
                if self._uses_bias:
                    print(f"Validation RMSE: {val_rmse:.4f} | MacroMAE: {val_macro_mae:.4f}")
                else:
                    print(f"Validation RMSE: {val_rmse:.4f}")
            if previous is not None:
                improvement = previous - val_rmse
                if cfg.verbose:
                    print(f"Improvement: {improvement:.6f}")
                if self._should_stop(improvement, cfg.tol):
                    if cfg.verbose:
                        print("Early stopping: small improvement on validation.")
                    self.history_["stopped_early"] = True
                    break
            previous = val_rmse
        if self.history_["iterations"] > 0:
            self._pull_state()
        return self

    def predict(self, user_ids, item_ids, global_mean=0.0):
        return self._need_ctx().predict(np.asarray(user_ids, dtype=int), np.asarray(item_ids, dtype=int),
                                        use_bias=self._uses_bias, offset=global_mean)

    def _seen(self, df):
        keep = (df["u"] < self.n_users) & (df["i"] < self.n_items)
        return df[keep]

    def evaluate_rmse(self, df, global_mean):
        df = self._seen(df)
        if df.empty:
            print("Warning: No valid (u,i) pairs.")
            return np.nan
        y_true = df["rating"].to_numpy(dtype=float) + global_mean
        return rmse(y_true, self.predict(df["u"].to_numpy(), df["i"].to_numpy(), global_mean))

    def evaluate_macro_mae(self, df, global_mean):
        df = self._seen(df)
        if df.empty:
            return np.nan
        y_true = df["rating"].to_numpy(dtype=float) + global_mean
        return macro_mae(y_true, self.predict(df["u"].to_numpy(), df["i"].to_numpy(), global_mean))
