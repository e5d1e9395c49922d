"""GPU parity of the Gaussian MF path against the golden vectors captured from
the reference and against the CPU oracle."""
import numpy as np
import pytest

from helpers import frames, load_case, max_abs, rel_err, skewed_problem
from oracle import cavi_oracle as orc

pytestmark = pytest.mark.gpu

# Absolute tolerances on means / biases (values are O(0.1 .. 1)) and relative
# tolerance (w.r.t. the largest entry of a row's covariance) on covariances.
#   f64: device fp64 vs the reference's fp64 LU inverse -- rounding only
#   f32: fp32 storage and arithmetic; the normal matrices of heavy rows have
#        condition numbers up to ~1e3 here, so ~1e-4 after 20 sweeps
TOL = {"f64": {1: 1e-11, 3: 1e-10, 20: 1e-8}, "f32": {1: 2e-5, 3: 1e-4, 20: 2e-3}}


def _make(kind, meta, max_iter, tol, dtype, verbose=False):
    kw = dict(meta["base_cfg"], n_factors=meta["K"], random_state=meta["seed"], max_iter=max_iter,
              tol=tol, verbose=verbose)
    if kind == "gauss_bias":
        from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    else:
        from src.models.gaussian_mf_cavi import GaussianMFCAVI, GaussianMFCAVIConfig
    return GaussianMFCAVI(GaussianMFCAVIConfig(**kw), dtype=dtype)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["gauss_bias_s42_k8", "gauss_bias_s7_k16", "gauss_s42_k8", "gauss_s7_k16"])
def test_states_match_reference_goldens(case, dtype):
    d, meta = load_case(case)
    kind = meta["kind"]
    train, val = frames(d)
    gm = float(d["global_mean"])
    keys = ["m_theta", "m_beta"] + (["m_user_bias", "m_item_bias"] if kind == "gauss_bias" else [])
    for n_it in (0, 1, 3, 20):
        m = _make(kind, meta, n_it, 0.0, dtype).fit(train, global_mean=gm)
        tol = 1e-15 if n_it == 0 else TOL[dtype][n_it]
        for key in keys:
            assert max_abs(getattr(m, key), d[f"it{n_it}_{key}"]) <= tol, (key, n_it)
        for side in ("V_theta", "V_beta"):
            got = getattr(m, side)
            if f"it{n_it}_{side}" in d:
                want = d[f"it{n_it}_{side}"]
                scale = np.abs(want).max(axis=(1, 2), keepdims=True)
                assert np.max(np.abs(got - want) / scale) <= max(tol, 1e-15), (side, n_it)
                assert np.array_equal(got, np.swapaxes(got, 1, 2))
            else:
                want = d[f"it{n_it}_{side}_diag"]
                assert rel_err(np.einsum("nkk->nk", got), want) <= max(tol * 10, 1e-15), (side, n_it)
        if n_it == 3:
            ptol = 1e-9 if dtype == "f64" else 2e-4
            np.testing.assert_allclose(m.predict(d["pred_u"], d["pred_i"], gm), d["it3_predict"], rtol=ptol, atol=ptol)
            np.testing.assert_allclose(m.evaluate_rmse(val, gm), float(d["it3_val_rmse"]), rtol=ptol)
            if kind == "gauss_bias":
                np.testing.assert_allclose(m.evaluate_macro_mae(val, gm), float(d["it3_val_macro_mae"]), rtol=ptol)
        m.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", ["gauss_bias_s42_k8", "gauss_bias_s7_k16", "gauss_s42_k8", "gauss_s7_k16"])
def test_validation_trajectory_and_early_stop(case, dtype):
    d, meta = load_case(case)
    train, val = frames(d)
    gm = float(d["global_mean"])
    m = _make(meta["kind"], meta, 40, meta["stop_tol"], dtype).fit(train, val_df=val, global_mean=gm)
    assert m.history_["iterations"] == int(d["stop_iterations_run"])
    assert m.history_["stopped_early"] == bool(d["stop_early"])
    rtol = 1e-9 if dtype == "f64" else 2e-4
    np.testing.assert_allclose(m.history_["val_rmse"], d["stop_val_rmse"], rtol=rtol)
    if meta["kind"] == "gauss_bias":
        np.testing.assert_allclose(m.history_["val_macro_mae"], d["stop_val_macro_mae"], rtol=rtol)


@pytest.mark.parametrize("case", ["gauss_bias_s42_k8", "gauss_s42_k8"])
def test_verbose_log_matches_reference(case, capsys):
    d, meta = load_case(case)
    train, val = frames(d)
    _make(meta["kind"], meta, 40, meta["stop_tol"], "f64", verbose=True).fit(
        train, val_df=val, global_mean=float(d["global_mean"]))
    assert capsys.readouterr().out == str(d["stop_log"])


_ORACLE_CACHE = {}


def _oracle_states(K, bias, U, I, N, iters):
    """The oracle's state after `iters` iterations of the problem below: computed once per problem, whatever
    the number of device variants (dtypes, kernel switches) compared with it -- it is the slow side (O(N K^2))."""
    key = (K, bias, U, I, N, iters)
    if key not in _ORACLE_CACHE:
        u, i, x = skewed_problem(100 + K, U, I, N, rating_kind="centered")
        st = orc.init_gaussian(U, I, K, seed=5, bias=bias)
        idx = (orc.group_positions(u, U), orc.group_positions(i, I))
        for _ in range(iters):
            orc.gaussian_iteration(st, idx, u, i, x, 0.3, 0.5, 0.5, 1.0 if bias else None, vectorised=True)
        _ORACLE_CACHE[key] = st
    return _ORACLE_CACHE[key]


def _oracle_vs_device(K, dtype, bias, U=1500, I=300, N=40000, iters=2, env=None, with_oracle=True):
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER
    u, i, x = skewed_problem(100 + K, U, I, N, rating_kind="centered")
    init = orc.init_gaussian(U, I, K, seed=5, bias=bias)
    idx = (orc.group_positions(u, U), orc.group_positions(i, I))
    assert np.diff(idx[1][0]).max() > 2 * 512, "need rows split over several accumulate tasks"
    with pmf_hip.Context(U, I, K, dtype=dtype) as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, init["m_theta"]); ctx.set_array(ITEM, ARR_FACTOR, init["m_beta"])
        ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
        if bias:
            ctx.set_array(USER, ARR_BIAS, init["m_user_bias"]); ctx.set_array(ITEM, ARR_BIAS, init["m_item_bias"])
        for _ in range(iters):
            ctx.gauss_factor_sweep(USER, 0.3, 0.5)
            ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
            if bias:
                ctx.gauss_bias_sweep(USER, 0.3, 1.0)
                ctx.gauss_bias_sweep(ITEM, 0.3, 1.0)
        got = {"m_theta": ctx.get_array(USER, ARR_FACTOR), "m_beta": ctx.get_array(ITEM, ARR_FACTOR),
               "V_theta": ctx.get_array(USER, ARR_COV), "V_beta": ctx.get_array(ITEM, ARR_COV)}
        if bias:
            got["m_user_bias"] = ctx.get_array(USER, ARR_BIAS)
            got["m_item_bias"] = ctx.get_array(ITEM, ARR_BIAS)
    return got, (_oracle_states(K, bias, U, I, N, iters) if with_oracle else None)


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-10), ("f32", 2e-4)])
@pytest.mark.parametrize("K", [1, 5, 8, 16, 30, 32, 45, 48, 50, 64, 72, 88, 100, 128])
def test_half_sweeps_vs_oracle_skewed(K, dtype, tol):
    """Direct C-ABI calls on a skewed problem: split rows, empty rows, every
    solver width (register kernels up to 64, LDS kernel above)."""
    sizes = dict(N=40000, I=300) if K <= 64 else dict(N=7000, I=50, U=800)   # the oracle is O(N K^2)
    got, st = _oracle_vs_device(K, dtype, bias=True, **sizes)
    for key in ("m_theta", "m_beta", "m_user_bias", "m_item_bias"):
        assert max_abs(got[key], st[key]) <= tol, key
    for key in ("V_theta", "V_beta"):
        scale = np.abs(st[key]).max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(got[key] - st[key]) / scale) <= tol, key


@pytest.mark.parametrize("K", [5, 16, 20, 30, 32, 33, 40, 45, 48, 49, 50, 52, 64, 70, 79, 81, 88, 95, 100, 120, 128])
def test_mfma_kernel_matches_generic_kernel(K, monkeypatch):
    """The fp32 fast paths (K <= 64: one wavefront per task; 64 < K <= 128: two) (MFMA outer products, fused solve) against the
    generic accumulate kernel + standalone solve on the same inputs.  Above 64 also the un-fused launches
    (PMF_GAUSS_UNFUSED: accumulate-only kernel + `gauss_solve_pair_kernel`), i.e. both homes of the two-wave MFMA block
    sweep: 5 tiles of 16 rows (K = 70, 79), 6 (81, 88, 95), 7 (100) and 8 (120, 128), with K not a multiple of 4 (a
    pivot block that reaches into the identity padding: 70, 79, 81, 95) and not a multiple of 16 (ADVICE r2 asked for
    K = 79 / 81 / 88 / 95 when these sizes still ran the VALU row splits)."""
    small = dict(N=20000) if K <= 64 else dict(N=7000, I=50, U=800)
    fast, _ = _oracle_vs_device(K, "f32", bias=True, iters=1, with_oracle=False, **small)
    unfused = None
    if K > 64:
        monkeypatch.setenv("PMF_GAUSS_UNFUSED", "1")
        unfused, _ = _oracle_vs_device(K, "f32", bias=True, iters=1, with_oracle=False, **small)
        monkeypatch.delenv("PMF_GAUSS_UNFUSED")
    monkeypatch.setenv("PMF_GAUSS_GENERIC", "1")
    slow, _ = _oracle_vs_device(K, "f32", bias=True, iters=1, with_oracle=False, **small)
    for key in fast:
        assert max_abs(fast[key], slow[key]) <= 3e-5, key
        if unfused is not None:
            assert max_abs(unfused[key], slow[key]) <= 3e-5, (key, "unfused")


def test_empty_rows_keep_initial_state():
    """gaussian_mf_cavi_bias.py:134-135: rows without ratings are skipped."""
    import pmf_hip
    from pmf_hip import ARR_COV, ARR_FACTOR, ITEM, USER
    U, I, K = 40, 30, 8
    rng = np.random.default_rng(1)
    u = rng.integers(0, U, 300); i = rng.integers(0, I, 300)
    u[u == 7] = 8; i[i == 3] = 4; u[0], i[0] = U - 1, I - 1
    x = rng.normal(size=300)
    m0, b0 = rng.normal(size=(U, K)), rng.normal(size=(I, K))
    with pmf_hip.Context(U, I, K, dtype="f64") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, m0); ctx.set_array(ITEM, ARR_FACTOR, b0)
        ctx.set_cov_identity(USER, 1.0); ctx.set_cov_identity(ITEM, 1.0)
        ctx.gauss_factor_sweep(USER, 0.3, 0.5); ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
        assert np.array_equal(ctx.get_array(USER, ARR_FACTOR)[7], m0[7])
        assert np.array_equal(ctx.get_array(ITEM, ARR_FACTOR)[3], b0[3])
        assert np.array_equal(ctx.get_array(USER, ARR_COV)[7], np.eye(K))
        assert not np.allclose(ctx.get_array(USER, ARR_FACTOR)[8], m0[8])


@pytest.mark.parametrize("K,dtype,tol", [(136, "f32", 3e-4), (256, "f32", 6e-4), (150, "f64", 1e-9)])
def test_gaussian_beyond_128_factors(K, dtype, tol):
    """The reference has no K limit (its grids stop at 70); the context's limit is 256.  K > 128 runs the generic
    accumulate kernel and the block-per-row sweep -- matrix in LDS while it fits (fp32 K <= 200, fp64 K <= 141), in a
    per-block global scratch slice beyond -- against the oracle on a small skewed problem."""
    sizes = dict(U=400, I=60, N=4500) if K < 200 else dict(U=200, I=12, N=3000)      # the oracle is O(N K^2)
    got, st = _oracle_vs_device(K, dtype, bias=True, iters=2, **sizes)
    for key in ("m_theta", "m_beta", "m_user_bias", "m_item_bias"):
        assert max_abs(got[key], st[key]) <= tol, key
    for key in ("V_theta", "V_beta"):
        scale = np.abs(st[key]).max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(got[key] - st[key]) / scale) <= tol, key
