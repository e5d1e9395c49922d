"""The HIP path against vectors the REFERENCE produced at the K values that matter: the benchmark's
K = 64 / 128 (BASELINE.json) and the reference's own configurations K = 30 (Gaussian), 40 (Poisson),
20 (HPF) (best_hyperparams.txt:3-5).  tests/golden/make_golden.py (`headline`) ran the reference on a
120 x 48 x 1600 problem with the usual edge cases; nothing here goes through the oracle."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, frames, load_case, max_abs, rel_err

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "hk_*.npz")))
GAMMA_KEYS = {"poisson": ["E_theta", "E_beta", "a_theta", "b_beta"],
              "hpf": ["E_theta", "E_beta", "E_xi", "E_eta", "gamma_a_theta", "gamma_b_beta", "gamma_b_xi", "gamma_b_eta"]}

# state after 3 iterations: relative (Poisson/HPF), absolute on O(0.1..1) values (Gaussian)
#   f64: summation order / LU-vs-sweep rounding only;  f32: storage + arithmetic in fp32
GAMMA_TOL = {"f64": 1e-11, "f32": 5e-5}
GAUSS_TOL = {"f64": 1e-10, "f32": 2e-4}
TRAJ_RTOL = {"f64": 1e-10, "f32": 2e-4}


def _model(kind, meta, max_iter, tol, dtype):
    kw = dict(meta["base_cfg"], n_factors=meta["K"], random_state=meta["seed"], max_iter=max_iter, tol=tol, verbose=False)
    if kind == "hpf":
        from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
        return HPF_CAVI(HPF_CAVI_Config(**kw), dtype=dtype)
    if kind == "poisson":
        from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
        return PoissonMFCAVI(PoissonMFCAVIConfig(**kw), dtype=dtype)
    if kind == "gauss_bias":
        from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    else:
        from src.models.gaussian_mf_cavi import GaussianMFCAVI, GaussianMFCAVIConfig
    return GaussianMFCAVI(GaussianMFCAVIConfig(**kw), dtype=dtype)


def test_fixture_set_covers_the_headline_configurations():
    assert set(CASES) == {"hk_gauss_bias_k30", "hk_gauss_bias_k64", "hk_gauss_bias_k128", "hk_gauss_k64",
                          "hk_poisson_k40", "hk_poisson_k64", "hk_hpf_k20", "hk_hpf_k64",
                          "hk_gauss_bias_k80", "hk_gauss_bias_k112"}     # (round 3: 5 and 7 tiles of the MFMA block sweep)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", CASES)
def test_state_after_three_iterations_matches_the_reference(case, dtype):
    d, meta = load_case(case)
    kind, gm = meta["kind"], float(d["global_mean"])
    train, val = frames(d)
    gauss = kind.startswith("gauss")
    m = _model(kind, meta, 3, 0.0 if gauss else None, dtype)
    if gauss:
        m.fit(train, global_mean=gm)
        tol = GAUSS_TOL[dtype]
        keys = ["m_theta", "m_beta"] + (["m_user_bias", "m_item_bias"] if kind == "gauss_bias" else [])
        for key in keys:
            assert max_abs(getattr(m, key), d[f"it3_{key}"]) <= tol, key
        for side in ("theta", "beta"):
            V = getattr(m, f"V_{side}")
            assert rel_err(np.einsum("nkk->nk", V), d[f"it3_V_{side}_diag"]) <= tol * 10, side
            want = d[f"it3_V_{side}_rows"]
            got = V[meta["cov_rows"]]
            assert np.max(np.abs(got - want)) <= tol * 10 * np.abs(want).max(), side
        pred = m.predict(d["pred_u"], d["pred_i"], gm)
    else:
        m.fit(train)
        for key in GAMMA_KEYS[kind]:
            assert rel_err(getattr(m, key), d[f"it3_{key}"]) <= GAMMA_TOL[dtype], key
        pred = m.predict(d["pred_u"], d["pred_i"])
    ptol = 1e-9 if dtype == "f64" else 3e-4
    np.testing.assert_allclose(pred, d["it3_predict"], rtol=ptol, atol=ptol)
    m.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", CASES)
def test_validation_trajectory_matches_the_reference(case, dtype):
    """Validation RMSE / MacroMAE after each of 5 iterations, as the reference's own evaluate_* gave them."""
    d, meta = load_case(case)
    kind, gm = meta["kind"], float(d["global_mean"])
    train, val = frames(d)
    gauss = kind.startswith("gauss")
    m = _model(kind, meta, meta["traj_iters"], -1.0 if gauss else None, dtype)
    if gauss:
        m.fit(train, val_df=val, global_mean=gm)
    else:
        m.fit(train, val_df=val)
    assert m.history_["iterations"] == meta["traj_iters"] and not m.history_["stopped_early"]
    np.testing.assert_allclose(m.history_["val_rmse"], d["traj_val_rmse"], rtol=TRAJ_RTOL[dtype])
    if kind != "gauss":
        np.testing.assert_allclose(m.history_["val_macro_mae"], d["traj_val_macro_mae"], rtol=TRAJ_RTOL[dtype])
    m.close()
