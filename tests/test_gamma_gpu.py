"""GPU parity of the Poisson MF / HPF path (C-ABI -> HIP kernels) against the
golden vectors captured from the reference and against the CPU oracle."""
import numpy as np
import pytest

from helpers import frames, load_case, rel_err, skewed_problem
from oracle import cavi_oracle as orc

pytestmark = pytest.mark.gpu

# Tolerances (relative, on every element of every state array):
#   f64 device arithmetic vs the fp64 reference: summation-order noise only
#   f32 device arithmetic: ~1e-6 per sweep, grows slowly with the iteration count
TOL = {"f64": {1: 1e-12, 3: 1e-11, 20: 1e-9}, "f32": {1: 2e-5, 3: 5e-5, 20: 5e-4}}

HPF_KEYS = ["gamma_a_theta", "gamma_b_theta", "gamma_a_beta", "gamma_b_beta", "gamma_b_xi",
            "gamma_b_eta", "E_theta", "E_beta", "E_xi", "E_eta"]
POI_KEYS = ["a_theta", "b_theta", "a_beta", "b_beta", "E_theta", "E_beta"]


def _make(kind, meta, max_iter, tol, dtype):
    from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
    from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
    kw = dict(meta["base_cfg"], n_factors=meta["K"], random_state=meta["seed"], max_iter=max_iter,
              tol=tol, verbose=False)
    if kind == "hpf":
        return HPF_CAVI(HPF_CAVI_Config(**kw), dtype=dtype)
    return PoissonMFCAVI(PoissonMFCAVIConfig(**kw), dtype=dtype)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["hpf_s42_k8", "hpf_s7_k16", "poisson_s42_k8", "poisson_s7_k16"])
def test_states_match_reference_goldens(case, dtype):
    d, meta = load_case(case)
    kind = meta["kind"]
    train, val = frames(d)
    keys = HPF_KEYS if kind == "hpf" else POI_KEYS
    for n_it in (0, 1, 3, 20):
        m = _make(kind, meta, n_it, None, dtype).fit(train)
        for key in keys:
            tol = 1e-15 if n_it == 0 else TOL[dtype][n_it]
            assert rel_err(getattr(m, key), d[f"it{n_it}_{key}"]) <= tol, (key, n_it)
        if n_it == 3:
            ptol = 1e-11 if dtype == "f64" else 1e-4
            np.testing.assert_allclose(m.predict(d["pred_u"], d["pred_i"]), d["it3_predict"], rtol=ptol, atol=1e-12)
            np.testing.assert_allclose(m.evaluate_rmse(val), float(d["it3_val_rmse"]), rtol=ptol)
            np.testing.assert_allclose(m.evaluate_macro_mae(val), float(d["it3_val_macro_mae"]), rtol=ptol)
        m.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["hpf_s42_k8", "hpf_s7_k16", "poisson_s42_k8", "poisson_s7_k16"])
def test_validation_trajectory_and_early_stop(case, dtype):
    d, meta = load_case(case)
    train, val = frames(d)
    m = _make(meta["kind"], meta, 40, meta["stop_tol"], dtype).fit(train, val_df=val)
    assert m.history_["iterations"] == int(d["stop_iterations_run"])
    assert m.history_["stopped_early"] == bool(d["stop_early"])
    rtol = 1e-10 if dtype == "f64" else 1e-4
    np.testing.assert_allclose(m.history_["val_rmse"], d["stop_val_rmse"], rtol=rtol)
    np.testing.assert_allclose(m.history_["val_macro_mae"], d["stop_val_macro_mae"], rtol=rtol)


def test_verbose_log_matches_reference(capsys):
    d, meta = load_case("hpf_s42_k8")
    train, val = frames(d)
    from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
    kw = dict(meta["base_cfg"], n_factors=meta["K"], random_state=meta["seed"], max_iter=40,
              tol=meta["stop_tol"], verbose=True)
    HPF_CAVI(HPF_CAVI_Config(**kw), dtype="f64").fit(train, val_df=val)
    assert capsys.readouterr().out == str(d["stop_log"])


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-11), ("f32", 3e-5)])
@pytest.mark.parametrize("K", [1, 3, 8, 20, 28, 40, 52, 64, 100, 128, 250, 256])
def test_half_sweeps_vs_oracle_skewed(K, dtype, tol):
    """Direct C-ABI calls on a skewed problem: rows far above one chunk (split
    rows), empty rows, every lane-group width."""
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ARR_HYPER_RATE, ARR_PRIOR_RATE, ARR_RATE, ARR_SHAPE, ITEM, USER
    U, I, N = 3000, 400, 60000
    u, i, x = skewed_problem(K, U, I, N)
    st = orc.init_hpf(U, I, K, 0.3, 5.0, 5.0, 0.3, 5.0, 5.0, seed=3)
    idx = (orc.group_positions(u, U), orc.group_positions(i, I))
    assert np.diff(idx[1][0]).max() > 2000 and (np.diff(idx[0][0]) == 0).any()
    with pmf_hip.Context(U, I, K, dtype=dtype) as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, st["E_theta"])
        ctx.set_array(ITEM, ARR_FACTOR, st["E_beta"])
        ctx.set_array(USER, ARR_PRIOR_RATE, st["E_xi"])
        ctx.set_array(ITEM, ARR_PRIOR_RATE, st["E_eta"])
        for _ in range(2):
            orc.hpf_iteration(st, idx, u, i, x, 0.3, 5.0, 0.3, 5.0, orc.gamma_half_sweep_segsum)
            ctx.gamma_sweep(USER, 0.3, 0.0, True, st["gamma_a_xi"], 5.0)
            ctx.gamma_sweep(ITEM, 0.3, 0.0, True, st["gamma_a_eta"], 5.0)
        got = {"gamma_a_theta": ctx.get_array(USER, ARR_SHAPE), "gamma_b_theta": ctx.get_array(USER, ARR_RATE),
               "gamma_a_beta": ctx.get_array(ITEM, ARR_SHAPE), "gamma_b_beta": ctx.get_array(ITEM, ARR_RATE),
               "E_theta": ctx.get_array(USER, ARR_FACTOR), "E_beta": ctx.get_array(ITEM, ARR_FACTOR),
               "E_xi": ctx.get_array(USER, ARR_PRIOR_RATE), "E_eta": ctx.get_array(ITEM, ARR_PRIOR_RATE),
               "gamma_b_xi": ctx.get_array(USER, ARR_HYPER_RATE), "gamma_b_eta": ctx.get_array(ITEM, ARR_HYPER_RATE)}
    for key, val in got.items():
        assert rel_err(val, st[key]) <= tol, key


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_poisson_rate_floor_and_zero_ratings(dtype):
    """rate clamp 1e-10 (hpf_cavi.py:141): all-zero factors give rate = floor;
    zero ratings contribute nothing to the shape."""
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ARR_RATE, ARR_SHAPE, ITEM, USER
    U, I, K = 6, 5, 4
    u = np.array([0, 0, 1, 2, 5, 5, 5]); i = np.array([0, 1, 1, 4, 0, 0, 3])
    x = np.array([3.0, 0.0, 2.0, 1.0, 4.0, 4.0, 0.0])
    Et = np.zeros((U, K)); Eb = np.abs(np.random.default_rng(0).normal(size=(I, K))) * 1e-3
    Et[5] = 1e-9
    idx = (orc.group_positions(u, U), orc.group_positions(i, I))
    a, b = orc.gamma_half_sweep_rows(Et, Eb, *idx[0], i, x, 0.1, 0.5)
    with pmf_hip.Context(U, I, K, dtype=dtype) as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, Et)
        ctx.set_array(ITEM, ARR_FACTOR, Eb)
        ctx.gamma_sweep(USER, 0.1, 0.5)
        tol = 1e-12 if dtype == "f64" else 1e-5
        assert rel_err(ctx.get_array(USER, ARR_SHAPE), a) <= tol
        assert rel_err(ctx.get_array(USER, ARR_RATE), b) <= tol
        assert rel_err(ctx.get_array(USER, ARR_FACTOR), a / b) <= tol


EXT_KEYS = ["a_theta", "b_theta", "a_beta", "b_beta", "a_phi", "b_phi", "a_psi", "b_psi",
            "E_theta", "E_beta", "E_phi", "E_psi"]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["poisson_ext_s42_k8", "poisson_ext_s7_k16"])
def test_extended_poisson_matches_reference_goldens(case, dtype):
    from src.models.poisson_mf_extended_cavi import PoissonMFExtendedCAVI, PoissonMFExtendedCAVIConfig
    d, meta = load_case(case)
    train, val = frames(d)
    kw = dict(meta["base_cfg"], n_factors=meta["K"], random_state=meta["seed"], verbose=False)
    for n_it in (0, 1, 3, 20):
        m = PoissonMFExtendedCAVI(PoissonMFExtendedCAVIConfig(max_iter=n_it, tol=None, **kw), dtype=dtype).fit(train)
        for key in EXT_KEYS:
            tol = 1e-15 if n_it == 0 else TOL[dtype][n_it] * (10 if dtype == "f64" else 4)
            assert rel_err(getattr(m, key), d[f"it{n_it}_{key}"]) <= tol, (key, n_it)
        if n_it == 3:
            ptol = 1e-10 if dtype == "f64" else 2e-4
            np.testing.assert_allclose(m.predict(d["pred_u"], d["pred_i"]), d["it3_predict"], rtol=ptol, atol=1e-12)
            np.testing.assert_allclose(m.evaluate_rmse(val), float(d["it3_val_rmse"]), rtol=ptol)
        m.close()
    m = PoissonMFExtendedCAVI(PoissonMFExtendedCAVIConfig(max_iter=40, tol=meta["stop_tol"], **kw),
                              dtype=dtype).fit(train, val_df=val)
    assert m.history_["iterations"] == int(d["stop_iterations_run"]) and m.history_["stopped_early"] == bool(d["stop_early"])
    np.testing.assert_allclose(m.history_["val_rmse"], d["stop_val_rmse"], rtol=1e-9 if dtype == "f64" else 2e-4)


def test_extended_poisson_split_rows_vs_oracle():
    """Heavy rows (several chunks) and empty rows through pmf_gamma_ext_sweep."""
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ARR_RATE, ARR_SCALE, ARR_SCALE_RATE, ARR_SCALE_SHAPE, ARR_SHAPE, ITEM, USER
    U, I, N, K = 2000, 150, 40000, 12
    u, i, x = skewed_problem(9, U, I, N)
    st = orc.init_poisson_ext(U, I, K, 0.3, 1.0, seed=4)
    idx = (orc.group_positions(u, U), orc.group_positions(i, I))
    assert np.diff(idx[1][0]).max() > 1000
    with pmf_hip.Context(U, I, K, dtype="f64") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, st["E_theta"]); ctx.set_array(ITEM, ARR_FACTOR, st["E_beta"])
        ctx.set_array(USER, ARR_SCALE, st["E_phi"]); ctx.set_array(ITEM, ARR_SCALE, st["E_psi"])
        for _ in range(2):
            orc.poisson_ext_iteration(st, idx, u, i, x, 0.3, 1.0)
            ctx.gamma_ext_sweep(USER, 0.3, 1.0)
            ctx.gamma_ext_sweep(ITEM, 0.3, 1.0)
        got = {"a_theta": ctx.get_array(USER, ARR_SHAPE), "b_theta": ctx.get_array(USER, ARR_RATE),
               "E_theta": ctx.get_array(USER, ARR_FACTOR), "E_phi": ctx.get_array(USER, ARR_SCALE),
               "a_phi": ctx.get_array(USER, ARR_SCALE_SHAPE), "b_phi": ctx.get_array(USER, ARR_SCALE_RATE),
               "a_beta": ctx.get_array(ITEM, ARR_SHAPE), "b_beta": ctx.get_array(ITEM, ARR_RATE),
               "E_beta": ctx.get_array(ITEM, ARR_FACTOR), "E_psi": ctx.get_array(ITEM, ARR_SCALE),
               "a_psi": ctx.get_array(ITEM, ARR_SCALE_SHAPE), "b_psi": ctx.get_array(ITEM, ARR_SCALE_RATE)}
    for key, val in got.items():
        assert rel_err(val, st[key]) <= 1e-10, key


def test_bad_ids_are_rejected():
    import pmf_hip
    with pmf_hip.Context(10, 10, 8) as ctx:
        with pytest.raises(pmf_hip.PmfError, match="outside"):
            ctx.set_ratings([0, 10], [0, 1], [1.0, 2.0])
        with pytest.raises(pmf_hip.PmfError, match="has not been set"):
            ctx.set_ratings([0, 9], [0, 1], [1.0, 2.0])
            ctx.gamma_sweep(0, 0.3, 1.0)
