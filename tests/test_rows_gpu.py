"""Row-subset exchange of device state (pmf_get_array_rows / pmf_set_array_rows): what the reference's
row indexing does (`V_beta[j_idx]`, `V_theta[i] = ...`, gaussian_mf_cavi_bias.py:146-162) without moving
the whole stack."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("K", [1, 5, 20, 64, 70])
def test_row_reads_equal_the_full_download(dtype, K):
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ARR_RATE, ITEM, USER
    rng = np.random.default_rng(K)
    U, I = 700, 45
    with pmf_hip.Context(U, I, K, dtype=dtype) as ctx:
        full = {}
        for side, rows in ((USER, U), (ITEM, I)):
            A = rng.standard_normal((rows, K, K))
            full[side, ARR_COV] = A @ A.transpose(0, 2, 1)          # symmetric, every row different
            full[side, ARR_FACTOR] = rng.standard_normal((rows, K))
            full[side, ARR_RATE] = rng.random((rows, K))
            full[side, ARR_BIAS] = rng.standard_normal(rows)
        for (side, arr), a in full.items():
            ctx.set_array(side, arr, a)
        for (side, arr), a in full.items():
            rows = ctx.rows(side)
            ids = np.concatenate([[rows - 1, 0, rows - 1], rng.integers(0, rows, 40)])   # any order, repeats
            got = ctx.get_array_rows(side, arr, ids)
            np.testing.assert_array_equal(got, ctx.get_array(side, arr)[ids])
            assert got.shape == (len(ids),) + a.shape[1:]
        assert ctx.get_array_rows(USER, ARR_COV, []).shape == (0, K, K)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_row_writes_land_in_their_rows_only(dtype):
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, USER
    rng = np.random.default_rng(3)
    U, I, K = 300, 20, 12
    with pmf_hip.Context(U, I, K, dtype=dtype) as ctx:
        for arr, shape in ((ARR_COV, (U, K, K)), (ARR_FACTOR, (U, K)), (ARR_BIAS, (U,))):
            base = rng.standard_normal(shape)
            if arr == ARR_COV:
                base = base + base.transpose(0, 2, 1)
            ctx.set_array(USER, arr, base)
            before = ctx.get_array(USER, arr)
            ids = rng.choice(U, 25, replace=False)
            new = rng.standard_normal((25,) + shape[1:])
            if arr == ARR_COV:
                new = new + new.transpose(0, 2, 1)
            ctx.set_array_rows(USER, arr, ids, new)
            after = ctx.get_array(USER, arr)
            want = before.copy()
            want[ids] = new.astype(ctx.np_dtype).astype(np.float64)
            np.testing.assert_array_equal(after, want)


def test_row_ids_are_validated_and_unset_arrays_refused():
    import pmf_hip
    from pmf_hip import ARR_COV, ARR_FACTOR, ITEM, USER
    with pmf_hip.Context(50, 10, 4) as ctx:
        ctx.set_array(USER, ARR_FACTOR, np.zeros((50, 4)))
        with pytest.raises(pmf_hip.PmfError, match="outside"):
            ctx.get_array_rows(USER, ARR_FACTOR, [0, 50])
        with pytest.raises(pmf_hip.PmfError, match="outside"):
            ctx.set_array_rows(USER, ARR_FACTOR, [-1], np.zeros((1, 4)))
        with pytest.raises(pmf_hip.PmfError, match="has not been set"):
            ctx.get_array_rows(ITEM, ARR_COV, [0])
        with pytest.raises(ValueError):
            ctx.set_array_rows(USER, ARR_FACTOR, [0, 1], np.zeros((3, 4)))


def test_a_sweep_from_row_written_covariances_matches_numpy():
    """distinct covariances written row by row are what the next half-sweep gathers"""
    import pmf_hip
    from pmf_hip import ARR_COV, ARR_FACTOR, ITEM, USER
    rng = np.random.default_rng(11)
    U, I, K, N = 60, 25, 9, 900
    u, i = rng.integers(0, U, N), rng.integers(0, I, N)
    x = rng.standard_normal(N)
    m_beta = rng.standard_normal((I, K))
    A = rng.standard_normal((I, K, K))
    V_beta = A @ A.transpose(0, 2, 1) / K + np.eye(K)
    with pmf_hip.Context(U, I, K, dtype="f64") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, np.zeros((U, K))); ctx.set_array(ITEM, ARR_FACTOR, m_beta)
        ctx.set_cov_identity(USER, 1.0); ctx.set_cov_identity(ITEM, 1.0)
        order = rng.permutation(I)
        ctx.set_array_rows(ITEM, ARR_COV, order, V_beta[order])
        ctx.gauss_factor_sweep(USER, 0.3, 0.5)
        rows = np.arange(U)
        got_V, got_m = ctx.get_array_rows(USER, ARR_COV, rows), ctx.get_array_rows(USER, ARR_FACTOR, rows)
    for r in range(U):
        sel = np.nonzero(u == r)[0]
        if len(sel) == 0:
            continue
        mo = m_beta[i[sel]]
        V = np.linalg.inv(np.eye(K) / 0.5 + (V_beta[i[sel]].sum(axis=0) + mo.T @ mo) / 0.3)
        np.testing.assert_allclose(got_V[r], V, atol=1e-11)
        np.testing.assert_allclose(got_m[r], V @ (mo * x[sel][:, None]).sum(axis=0) / 0.3, atol=1e-11)


def test_model_covariance_rows_equal_the_full_attribute():
    """`V_theta_rows(ids)` / `V_beta_rows(ids)` read the packed device rows directly; the full attributes materialise the
    whole (rows, K, K) stack -- same values, and asking for rows first must not materialise the stack."""
    import pandas as pd
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    rng = np.random.default_rng(2)
    U, I, N, K = 400, 90, 6000, 12
    df = pd.DataFrame({"u": rng.integers(0, U, N), "i": rng.integers(0, I, N), "rating": rng.normal(size=N)})
    m = GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=K, max_iter=2, tol=0.0, verbose=False), dtype="f64").fit(df)
    uids, iids = rng.integers(0, m.n_users, 17), rng.integers(0, m.n_items, 9)
    rows_u, rows_i = m.V_theta_rows(uids), m.V_beta_rows(iids)
    assert m._V_theta is None and m._V_beta is None            # nothing was materialised
    assert rows_u.shape == (17, K, K) and rows_i.shape == (9, K, K)
    np.testing.assert_array_equal(rows_u, m.V_theta[uids])
    np.testing.assert_array_equal(rows_i, m.V_beta[iids])
    np.testing.assert_array_equal(m.V_theta_rows(uids), rows_u)  # (served from the materialised stack now)
    m.close()
