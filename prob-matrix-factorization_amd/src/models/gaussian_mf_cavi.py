"""Gaussian matrix factorisation without biases -- MI355X engine.

Drop-in for the reference's `src/models/gaussian_mf_cavi.py` (same surface as
the bias model minus `eta_bias2`, the bias attributes and `evaluate_macro_mae`
in the per-iteration log)."""
from dataclasses import dataclass

from src.models._gaussian_host import GaussianHost


@dataclass
class GaussianMFCAVIConfig:
    n_factors: int = 10
    sigma2: float = 1.0
    eta_theta2: float = 1.0
    eta_beta2: float = 1.0
    max_iter: int = 20
    tol: float = 1e-3
    random_state: int = 42
    verbose: bool = True


class GaussianMFCAVI(GaussianHost):
    """r_ij ~ N(mu + theta_i . beta_j, sigma2)."""
    _uses_bias = False
    _gaussian = True

    def __init__(self, config: GaussianMFCAVIConfig, dtype=None, device=None, comm=None, presharded=False):
        super().__init__(config, dtype, device, comm, presharded)
