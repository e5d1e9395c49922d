"""Gaussian matrix factorisation with biases, MAP point estimates by stochastic
gradient steps -- MI355X engine.

**No reference counterpart**: the reference's Gaussian model is CAVI only
(`gaussian_mf_cavi_bias.py`); this class optimises the MAP objective of the same
model and exists because the project brief names a gradient mode.  Its parity is
therefore unpinned -- the tests hold it to this build's own CPU restatement of the
same definition (include/pmf_hip.h, `pmf_gauss_sgd_sweep`) and to the CAVI model's
validation RMSE.

Surface as `GaussianMFCAVI`: `fit(train_df, val_df=None, global_mean=0.0)`,
`predict`, `evaluate_rmse`, `evaluate_macro_mae`, attributes `m_theta, m_beta,
m_user_bias, m_item_bias` (no covariances: `V_theta` / `V_beta` are None)."""
from dataclasses import dataclass

from pmf_hip import dist as pdist
from src.models._gaussian_host import GaussianHost


@dataclass
class GaussianMFSGDConfig:
    n_factors: int = 10
    sigma2: float = 1.0
    eta_theta2: float = 1.0
    eta_beta2: float = 1.0
    eta_bias2: float = 1.0
    lr: float = 0.01          # step size of one rating's update
    max_iter: int = 20        # epochs (one pass over the users' ratings, one over the items')
    tol: float = 1e-3
    random_state: int = 42
    verbose: bool = True


class GaussianMFSGD(GaussianHost):
    """argmin 1/(2 sigma2) sum (r_ij - b_i - b_j - theta_i . beta_j)^2 + Gaussian priors."""
    _uses_bias = True
    _gaussian = False            # exchanged item statistics are [I x (Kpad + 4)], not covariances
    _iteration_label = "SGD epoch"

    def __init__(self, config: GaussianMFSGDConfig, dtype=None, device=None, comm=None, presharded=False):
        super().__init__(config, dtype, device, comm, presharded)

    V_theta = property(lambda self: None, lambda self, value: None)
    V_beta = property(lambda self: None, lambda self, value: None)

    @staticmethod
    def _should_stop(improvement, tol):
        # MAP point estimates over-fit once the validation error turns: unlike the CAVI rule a
        # negative improvement stops too (the rule the reference uses for Poisson / HPF)
        return improvement < tol

    def _prepare(self, ctx):
        pass

    def _iterate(self, ctx):
        cfg = self.config
        pdist.gaussian_sgd_iteration(ctx, self._comm, None, cfg.lr, cfg.sigma2, cfg.eta_theta2,
                                     cfg.eta_beta2, cfg.eta_bias2)
