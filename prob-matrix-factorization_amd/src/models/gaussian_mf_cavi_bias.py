"""Gaussian matrix factorisation with user/item biases, mean-field CAVI with
full K x K posterior covariances -- MI355X engine.

Drop-in for the reference's `src/models/gaussian_mf_cavi_bias.py`: same config
dataclass, `fit(train_df, val_df=None, global_mean=0.0)`, `predict(user_ids,
item_ids, global_mean=0.0)`, `evaluate_rmse(df, global_mean)`,
`evaluate_macro_mae(df, global_mean)` and attributes `m_theta, m_beta,
V_theta, V_beta, m_user_bias, m_item_bias, global_mean`."""
from dataclasses import dataclass

from src.models._gaussian_host import GaussianHost


@dataclass
class GaussianMFCAVIConfig:
    n_factors: int = 10
    sigma2: float = 1.0
    eta_theta2: float = 1.0
    eta_beta2: float = 1.0
    eta_bias2: float = 1.0
    max_iter: int = 20
    tol: float = 1e-3
    random_state: int = 42
    verbose: bool = True


class GaussianMFCAVI(GaussianHost):
    """r_ij ~ N(mu + b_i + b_j + theta_i . beta_j, sigma2)."""
    _uses_bias = True
    _gaussian = True

    def __init__(self, config: GaussianMFCAVIConfig, dtype=None, device=None, comm=None, presharded=False):
        super().__init__(config, dtype, device, comm, presharded)
