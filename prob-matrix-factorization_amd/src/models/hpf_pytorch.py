"""HPF MAP estimation with softplus-reparameterised factors -- PyTorch (ROCm).

Drop-in for the reference's `src/models/hpf_pytorch.py` (same constructor,
parameters `theta_uncons / beta_uncons / xi_uncons / eta_uncons`, buffers
`user_scale / item_scale`, properties `theta / beta / xi / eta`, `forward`,
`loss`, `predict`).  This model stays a `torch.nn.Module`; on an MI355X it is
simply placed on `cuda` (the reference never moves it off the CPU).  The loss
evaluates each softplus table once per call instead of once per access; the
value and gradients are unchanged (pinned by tests/golden/hpf_torch.npz)."""
from dataclasses import dataclass

import sys
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_engine = sys.modules.get("pmf_hip")
if _engine is not None and _engine.loaded_before_torch() and not torch.cuda.is_available():
    # libpmf_hip.so mapped /opt/rocm's HIP runtime before torch could map its bundled one
    warnings.warn("torch was imported after the first pmf_hip engine call and finds no GPU: import torch "
                  "(or this module) before fitting a CAVI model, or set PMF_HIP_TORCH_PRELOAD=1; "
                  "HPF_PyTorch will run on the CPU in this process", RuntimeWarning, stacklevel=2)


@dataclass
class HPF_PyTorch_Config:
    n_factors: int = 20
    a: float = 0.3
    a_prime: float = 1.0
    b_prime: float = 1.0
    c: float = 0.3
    c_prime: float = 1.0
    d_prime: float = 1.0
    lr: float = 0.001
    batch_size: int = 1024
    epochs: int = 20
    device: str = "cpu"
    verbose: bool = True


class HPF_PyTorch(nn.Module):
    def __init__(self, n_users, n_items, user_counts, item_counts, config: HPF_PyTorch_Config):
        super().__init__()
        self.config = config
        self.n_users, self.n_items, self.K = n_users, n_items, config.n_factors
        # 1/(count + 1e-6): each rating of a row carries 1/count of the row's prior term
        self.register_buffer("user_scale", 1.0 / (torch.tensor(user_counts, dtype=torch.float32) + 1e-6))
        self.register_buffer("item_scale", 1.0 / (torch.tensor(item_counts, dtype=torch.float32) + 1e-6))
        # draw order as the reference (hpf_pytorch.py:37-48): theta, beta, xi, eta
        self.theta_uncons = nn.Parameter(torch.randn(n_users, self.K) * 0.1)
        self.beta_uncons = nn.Parameter(torch.randn(n_items, self.K) * 0.1)
        self.xi_uncons = nn.Parameter(torch.randn(n_users) * 0.1)
        self.eta_uncons = nn.Parameter(torch.randn(n_items) * 0.1)

    @property
    def theta(self):
        return F.softplus(self.theta_uncons)

    @property
    def beta(self):
        return F.softplus(self.beta_uncons)

    @property
    def xi(self):
        return F.softplus(self.xi_uncons)

    @property
    def eta(self):
        return F.softplus(self.eta_uncons)

    def forward(self, user_ids, item_ids):
        return (self.theta[user_ids] * self.beta[item_ids]).sum(dim=1)

    def loss(self, user_ids, item_ids, ratings):
        """Poisson negative log-likelihood of the batch + the Gamma log-priors of
        the rows that appear in it, each weighted by 1/row-count
        (reference hpf_pytorch.py:71-184)."""
        cfg = self.config
        th = self.theta[user_ids]          # (B, K)
        be = self.beta[item_ids]
        xi = self.xi[user_ids]             # (B,)
        eta = self.eta[item_ids]
        lam = torch.clamp((th * be).sum(dim=1), min=1e-6)
        nll = (lam - ratings * torch.log(lam)).sum()
        su, si = self.user_scale[user_ids], self.item_scale[item_ids]
        xi_c, eta_c = xi.unsqueeze(1), eta.unsqueeze(1)
        prior_theta = ((-cfg.a * torch.log(xi_c) + xi_c * th - (cfg.a - 1) * torch.log(th)).sum(dim=1) * su).sum()
        prior_beta = ((-cfg.c * torch.log(eta_c) + eta_c * be - (cfg.c - 1) * torch.log(be)).sum(dim=1) * si).sum()
        prior_xi = ((-(cfg.a_prime - 1) * torch.log(xi) + cfg.b_prime * xi) * su).sum()
        prior_eta = ((-(cfg.c_prime - 1) * torch.log(eta) + cfg.d_prime * eta) * si).sum()
        return nll + prior_theta + prior_beta + prior_xi + prior_eta

    def predict(self, user_ids, item_ids):
        dev = self.theta_uncons.device
        if isinstance(user_ids, np.ndarray):
            user_ids = torch.from_numpy(user_ids).long()
        if isinstance(item_ids, np.ndarray):
            item_ids = torch.from_numpy(item_ids).long()
        with torch.no_grad():
            out = self.forward(user_ids.to(dev), item_ids.to(dev))
        return out.cpu().numpy()
