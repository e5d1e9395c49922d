"""Top-k items (dense reconstruction on the MFMA pipe + selection) against a
NumPy ranking of the same factors, and the "identical top-k rankings" claim:
fp32 engine vs the fp64 reference state on the golden problems."""
import numpy as np
import pytest

from helpers import frames, load_case

pytestmark = pytest.mark.gpu


def _rank(scores, k):
    # score descending, item id ascending on ties
    order = np.lexsort((np.arange(scores.shape[1])[None, :].repeat(len(scores), 0), -scores), axis=1)
    return order[:, :k]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("K,bias", [(8, False), (20, True), (64, False), (100, True)])
def test_topk_matches_numpy_ranking(K, bias, dtype):
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_FACTOR, ITEM, USER
    rng = np.random.default_rng(K)
    U, I, k = 333, 1999, 10
    A, B = rng.gamma(0.5, 1.0, (U, K)), rng.gamma(0.5, 1.0, (I, K))
    B[17] = B[5]          # exact ties: identical item rows
    B[1500] = B[5]
    bu, bi = rng.normal(size=U), rng.normal(size=I)
    bi[17] = bi[1500] = bi[5]
    users = rng.permutation(U)[:200]
    with pmf_hip.Context(U, I, K, dtype=dtype) as ctx:
        ctx.set_array(USER, ARR_FACTOR, A); ctx.set_array(ITEM, ARR_FACTOR, B)
        if bias:
            ctx.set_array(USER, ARR_BIAS, bu); ctx.set_array(ITEM, ARR_BIAS, bi)
        items, scores = ctx.topk_items(users, k, use_bias=bias)
        # what the device holds (fp32 contexts round the tables)
        Ad, Bd = ctx.get_array(USER, ARR_FACTOR), ctx.get_array(ITEM, ARR_FACTOR)
        full = Ad[users] @ Bd.T
        if bias:
            full = ctx.get_array(USER, ARR_BIAS)[users][:, None] + ctx.get_array(ITEM, ARR_BIAS)[None, :] + full
        pred = ctx.predict(np.repeat(users, k), items.reshape(-1), use_bias=bias).reshape(-1, k)
    assert items.shape == (200, k) and (items >= 0).all()
    tol = 1e-12 if dtype == "f64" else 3e-6
    np.testing.assert_allclose(scores, np.take_along_axis(full, items, axis=1), rtol=tol, atol=tol)
    np.testing.assert_allclose(scores, pred, rtol=tol, atol=tol)
    # scores are sorted, ties resolved towards the lower id
    assert (np.diff(scores, axis=1) <= 0).all()
    tie = np.diff(scores, axis=1) == 0
    assert (np.diff(items, axis=1)[tie] > 0).all()
    want = _rank(full, k)
    if dtype == "f64":
        assert np.array_equal(items, want)
    else:
        # fp32 accumulation order differs from NumPy's: allow swaps between near-equal scores only
        kth = np.take_along_axis(full, want[:, -1:], axis=1)
        got_scores = np.take_along_axis(full, items, axis=1)
        assert (got_scores >= kth - 1e-5 * np.abs(kth)).all()
        assert np.mean(items == want) > 0.98


def test_topk_all_users_batched_and_small_k_edge():
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ITEM, USER
    rng = np.random.default_rng(0)
    U, I, K = 5000, 70, 16
    A, B = rng.normal(size=(U, K)), rng.normal(size=(I, K))
    with pmf_hip.Context(U, I, K, dtype="f64") as ctx:
        ctx.set_array(USER, ARR_FACTOR, A); ctx.set_array(ITEM, ARR_FACTOR, B)
        items, scores = ctx.topk_items(np.arange(U), I)          # k = n_items: a full ranking
        assert np.array_equal(items, _rank(A @ B.T, I))
        with pytest.raises(pmf_hip.PmfError, match="outside"):
            ctx.topk_items([0], I + 1)
        with pytest.raises(pmf_hip.PmfError, match="outside"):
            ctx.topk_items([U], 3)


@pytest.mark.parametrize("case", ["hpf_s7_k16", "poisson_s42_k8", "gauss_bias_s7_k16"])
def test_fp32_engine_gives_the_reference_top10(case):
    """North-star check: after 20 iterations the fp32 engine ranks the same top-10
    items per user as the reference's fp64 state (golden vectors).  A difference
    only counts when the two items' reference scores differ by more than 1e-4
    relative (fp32 storage cannot separate closer scores)."""
    d, meta = load_case(case)
    kind = meta["kind"]
    train, _ = frames(d)
    kw = dict(meta["base_cfg"], n_factors=meta["K"], random_state=meta["seed"], max_iter=20, verbose=False)
    if kind == "hpf":
        from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
        m = HPF_CAVI(HPF_CAVI_Config(tol=None, **kw), dtype="f32").fit(train)
        ref = d["it20_E_theta"] @ d["it20_E_beta"].T
    elif kind == "poisson":
        from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
        m = PoissonMFCAVI(PoissonMFCAVIConfig(tol=None, **kw), dtype="f32").fit(train)
        ref = d["it20_E_theta"] @ d["it20_E_beta"].T
    else:
        from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
        m = GaussianMFCAVI(GaussianMFCAVIConfig(tol=0.0, **kw), dtype="f32").fit(train, global_mean=float(d["global_mean"]))
        ref = d["it20_m_user_bias"][:, None] + d["it20_m_item_bias"][None, :] + d["it20_m_theta"] @ d["it20_m_beta"].T
    users = np.arange(m.n_users)
    items, _ = m.top_k_items(users, 10)
    want = _rank(ref, 10)
    exact = np.all(items == want, axis=1)
    for uu in np.nonzero(~exact)[0]:
        a, b = ref[uu, items[uu]], ref[uu, want[uu]]
        assert np.allclose(a, b, rtol=1e-4, atol=1e-6), (uu, items[uu], want[uu])
    assert exact.mean() > 0.97
