"""The parity bar of BASELINE.md section 3, on the GPU:

  * BASELINE config #1 (10k x 2k, 200k ratings, K=16): the engine against vectors
    captured from the reference itself (tests/golden/medium_c1.npz);
  * long runs (100 iterations, K=64) in fp32 against the fp64 mode (which the
    golden tests tie to the reference at 1e-9): val RMSE |d| <= 1e-4, factor
    matrices max-rel <= 1e-3, identical top-10 per user.
"""
import os

import numpy as np
import pandas as pd
import pytest

from helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


def _frame(u, i, r):
    return pd.DataFrame({"u": u, "i": i, "rating": r})


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-9), ("f32", 2e-4)])
def test_baseline_config1_against_reference_vectors(dtype, tol):
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
    from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
    d = np.load(os.path.join(GOLDEN, "medium_c1.npz"))
    u, i, r = d["u"].astype(np.int64), d["i"].astype(np.int64), d["rating"].astype(np.float64)
    ru, ri = d["rows_u"], d["rows_i"]
    hpf = HPF_CAVI(HPF_CAVI_Config(n_factors=16, a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0,
                                   max_iter=2, tol=None, random_state=42, verbose=False), dtype=dtype).fit(_frame(u, i, r + 1))
    poi = PoissonMFCAVI(PoissonMFCAVIConfig(n_factors=16, a0=0.1, b0=0.5, max_iter=2, tol=None, random_state=42,
                                            verbose=False), dtype=dtype).fit(_frame(u, i, r))
    gm = float(d["gauss_bias_global_mean"])
    gau = GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=16, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0,
                                              max_iter=2, tol=0.0, random_state=42, verbose=False),
                         dtype=dtype).fit(_frame(u, i, r - gm), global_mean=gm)
    for kind, A, B in (("hpf", hpf.E_theta, hpf.E_beta), ("poisson", poi.E_theta, poi.E_beta)):
        assert rel_err(A[ru], d[f"{kind}_A_rows"]) <= tol and rel_err(B[ri], d[f"{kind}_B_rows"]) <= tol, kind
        assert A.sum() == pytest.approx(float(d[f"{kind}_A_sum"]), rel=tol)
        assert np.abs(B).sum() == pytest.approx(float(d[f"{kind}_B_abs"]), rel=tol)
    atol = 1e-9 if dtype == "f64" else 2e-4
    np.testing.assert_allclose(gau.m_theta[ru], d["gauss_bias_A_rows"], rtol=0, atol=atol)
    np.testing.assert_allclose(gau.m_beta[ri], d["gauss_bias_B_rows"], rtol=0, atol=atol)
    np.testing.assert_allclose(gau.m_user_bias.sum(), float(d["gauss_bias_bias_u_sum"]), rtol=0, atol=atol * 1e3)
    want_V = d["gauss_bias_Vtheta_rows"]
    got_V = gau._ctx.get_array(0, 5)[ru[:16]]
    assert np.max(np.abs(got_V - want_V)) / np.abs(want_V).max() <= atol


def _topk_agree(A32, B32, A64, B64, users, k=10, tie=1e-4):
    s32, s64 = A32[users] @ B32.T, A64[users] @ B64.T
    same = 0
    for a, b in zip(s32, s64):
        ta, tb = np.argsort(-a, kind="stable")[:k], np.argsort(-b, kind="stable")[:k]
        kth = b[tb[-1]]
        same += set(ta) == set(tb) or all(j in tb or abs(b[j] - kth) <= tie * abs(kth) for j in ta)
    return same / len(users)


@pytest.fixture(scope="module")
def midsize():
    from pmf_hip.synth import synth_ratings, train_val_split
    u, i, r = synth_ratings(20_000, 3_000, 440_000, seed=11)
    u[0], i[0] = 19_999, 2_999
    (tu, ti, tr), (vu, vi, vr) = train_val_split(u, i, r)
    tu[0], ti[0] = 19_999, 2_999
    return _frame(tu, ti, tr), _frame(vu, vi, vr)


@pytest.mark.parametrize("kind", ["hpf", "poisson", "gauss_bias"])
def test_fp32_vs_fp64_after_100_iterations_k64(kind, midsize):
    train, val = midsize
    fits = {}
    for dtype in ("f64", "f32"):
        if kind == "hpf":
            from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
            tr, va = train.copy(), val.copy()
            tr["rating"] += 1; va["rating"] += 1
            m = HPF_CAVI(HPF_CAVI_Config(n_factors=64, a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0,
                                         max_iter=100, tol=None, verbose=False), dtype=dtype).fit(tr, val_df=va)
            A, B = m.E_theta, m.E_beta
        elif kind == "poisson":
            from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
            m = PoissonMFCAVI(PoissonMFCAVIConfig(n_factors=64, a0=0.1, b0=0.5, max_iter=100, tol=None, verbose=False),
                              dtype=dtype).fit(train, val_df=val)
            A, B = m.E_theta, m.E_beta
        else:
            from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
            gm = float(train["rating"].mean())
            tr, va = train.copy(), val.copy()
            tr["rating"] -= gm; va["rating"] -= gm
            m = GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=64, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0,
                                                    max_iter=100, tol=-1.0, verbose=False), dtype=dtype).fit(
                tr, val_df=va, global_mean=gm)
            A, B = m.m_theta, m.m_beta
        assert m.history_["iterations"] == 100
        fits[dtype] = (A, B, np.array(m.history_["val_rmse"]))
        m.close()
    (A64, B64, r64), (A32, B32, r32) = fits["f64"], fits["f32"]
    assert np.max(np.abs(r32 - r64)) <= 1e-4                       # val RMSE trajectory, every iteration
    scale_a, scale_b = np.abs(A64).max(), np.abs(B64).max()
    assert np.max(np.abs(A32 - A64)) / scale_a <= 1e-3 and np.max(np.abs(B32 - B64)) / scale_b <= 1e-3
    users = np.random.default_rng(0).choice(len(A64), 500, replace=False)
    assert _topk_agree(A32, B32, A64, B64, users) >= 0.99
