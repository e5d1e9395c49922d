"""CPU checks of the host-side mirror: config dataclasses accept the
reference's hyper-parameter dictionaries, metrics match the goldens, the
synthetic generator is deterministic."""
import ast
import dataclasses
import os

import numpy as np
import pytest

# the four lines of the reference's best_hyperparams.txt (data, :3-6)
BEST = {
    "GaussianMF": "{'n_factors': 30, 'sigma2': 0.3, 'eta_theta2': 0.5, 'eta_beta2': 0.5, 'eta_bias2': 1.0, 'max_iter': 100, 'tol': 0.001, 'random_state': 42, 'verbose': True}",
    "PoissonMF": "{'n_factors': 40, 'a0': 0.1, 'b0': 0.5, 'max_iter': 150, 'tol': None, 'random_state': 42, 'verbose': True}",
    "HPF_CAVI": "{'n_factors': 20, 'a': 0.3, 'a_prime': 5.0, 'b_prime': 5.0, 'c': 0.3, 'c_prime': 5.0, 'd_prime': 5.0, 'max_iter': 100, 'tol': None, 'random_state': 42, 'verbose': True}",
}


def test_configs_are_field_compatible_with_reference_dicts():
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVIConfig
    from src.models.hpf_cavi import HPF_CAVI_Config
    from src.models.poisson_mf_cavi import PoissonMFCAVIConfig
    for name, cls in (("GaussianMF", GaussianMFCAVIConfig), ("PoissonMF", PoissonMFCAVIConfig),
                      ("HPF_CAVI", HPF_CAVI_Config)):
        d = ast.literal_eval(BEST[name])
        cfg = cls(**d)
        # config.txt is str(asdict(config)): same keys, same order, same repr
        assert str(dataclasses.asdict(cfg)) == BEST[name]


def test_default_configs_match_reference_defaults():
    from src.models.gaussian_mf_cavi import GaussianMFCAVIConfig as G0
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVIConfig as G1
    from src.models.hpf_cavi import HPF_CAVI_Config
    from src.models.poisson_mf_cavi import PoissonMFCAVIConfig
    assert dataclasses.asdict(G1()) == dict(n_factors=10, sigma2=1.0, eta_theta2=1.0, eta_beta2=1.0,
                                            eta_bias2=1.0, max_iter=20, tol=1e-3, random_state=42, verbose=True)
    assert dataclasses.asdict(G0()) == dict(n_factors=10, sigma2=1.0, eta_theta2=1.0, eta_beta2=1.0,
                                            max_iter=20, tol=1e-3, random_state=42, verbose=True)
    assert dataclasses.asdict(PoissonMFCAVIConfig()) == dict(n_factors=20, a0=0.3, b0=1.0, max_iter=100,
                                                             tol=1e-4, random_state=42, verbose=True)
    assert dataclasses.asdict(HPF_CAVI_Config()) == dict(n_factors=20, a=0.3, a_prime=0.3, b_prime=1.0, c=0.3,
                                                         c_prime=0.3, d_prime=1.0, max_iter=100, tol=1e-4,
                                                         random_state=42, verbose=True)


def test_metrics_match_reference_goldens(golden_dir):
    from src.evaluation.metrics import mae, macro_mae, rmse
    d = np.load(os.path.join(golden_dir, "metrics.npz"))
    assert rmse(d["y_true"], d["y_pred"]) == pytest.approx(float(d["rmse"]), rel=1e-14)
    assert mae(d["y_true"], d["y_pred"]) == pytest.approx(float(d["mae"]), rel=1e-14)
    assert macro_mae(d["y_true"], d["y_pred"]) == pytest.approx(float(d["macro_mae"]), rel=1e-14)
    assert macro_mae(d["y_true"] - 4.4, d["y_pred"] - 4.4) == pytest.approx(float(d["macro_mae_centered"]), rel=1e-14)


def test_initial_state_draw_order_matches_reference(golden_dir):
    """The host does the RNG initialisation; max_iter=0 goldens pin its order."""
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
    from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
    d = np.load(os.path.join(golden_dir, "hpf_s7_k16.npz"))
    m = HPF_CAVI(HPF_CAVI_Config(n_factors=16, a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0,
                                 random_state=7, verbose=False))
    m.n_users, m.n_items = d["it0_E_theta"].shape[0], d["it0_E_beta"].shape[0]
    m._initialize()
    for k in ("gamma_a_theta", "gamma_b_theta", "gamma_a_beta", "gamma_b_beta", "E_theta", "E_beta", "E_xi", "E_eta"):
        assert np.array_equal(getattr(m, k), d[f"it0_{k}"]), k
    d = np.load(os.path.join(golden_dir, "poisson_s42_k8.npz"))
    p = PoissonMFCAVI(PoissonMFCAVIConfig(n_factors=8, a0=0.1, b0=0.5, random_state=42, verbose=False))
    p.n_users, p.n_items = d["it0_E_theta"].shape[0], d["it0_E_beta"].shape[0]
    p._initialize_variational_params()
    for k in ("a_theta", "a_beta", "b_theta", "b_beta", "E_theta", "E_beta"):
        assert np.array_equal(getattr(p, k), d[f"it0_{k}"]), k
    d = np.load(os.path.join(golden_dir, "gauss_bias_s42_k8.npz"))
    g = GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=8, random_state=42, verbose=False))
    g.n_users, g.n_items = d["it0_m_theta"].shape[0], d["it0_m_beta"].shape[0]
    g._initialize_variational_params()
    assert np.array_equal(g.m_theta, d["it0_m_theta"]) and np.array_equal(g.m_beta, d["it0_m_beta"])
    assert not g.m_user_bias.any() and not g.m_item_bias.any()


def test_synthetic_generator_is_seeded_and_shaped():
    from pmf_hip.synth import synth_ratings, train_val_split
    a = synth_ratings(5000, 800, 60000, seed=3)
    b = synth_ratings(5000, 800, 60000, seed=3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    u, i, r = a
    assert u.dtype == np.int32 and i.dtype == np.int32 and r.dtype == np.float64
    assert 0 <= u.min() and u.max() < 5000 and 0 <= i.min() and i.max() < 800
    assert set(np.unique(r)) <= {0.0, 1.0, 2.0, 3.0, 4.0, 5.0}
    deg_i = np.bincount(i, minlength=800)
    assert deg_i.max() > 20 * np.median(deg_i)  # heavy-headed item popularity
    (tu, ti, tr), (vu, vi, vr) = train_val_split(u, i, r)
    assert len(tu) + len(vu) == len(u) and 0.88 < len(tu) / len(u) < 0.92
