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


@pytest.mark.parametrize("K,k,mode,signed", [(64, 10, 0, False), (64, 64, 1, False), (128, 10, 2, False), (20, 7, 1, False),
                                              (12, 33, 0, False),
                                              (100, 64, 1, False),   # stage buffers + 64-entry lists > 64 KB of LDS
                                              (64, 10, 0, True), (32, 50, 1, True), (128, 24, 0, True)])
def test_fused_topk_equals_the_two_phase_path(K, k, mode, signed, monkeypatch):
    """The fused kernel (MFMA score tiles + running k best in LDS, item range cut into segments and merged) against
    the two-phase path (score matrix in HBM, then select): same items, scores equal up to the summation order
    of the two kernels; exact ties and a NaN row included.  `signed`: Gaussian factors, so scores of both signs -- the
    lists are kept as integer keys whose order must be the floats' -- and, for the all-zero user, zeros of both signs
    (0 * -x = -0.0), which rank as equal."""
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_SCALE, ITEM, USER
    rng = np.random.default_rng(K + k)
    U, I = 700, 20011                       # not multiples of 32; 20011 items -> several segments
    if signed:
        A, B = rng.standard_normal((U, K)), rng.standard_normal((I, K))
    else:
        A, B = rng.gamma(0.5, 1.0, (U, K)), rng.gamma(0.5, 1.0, (I, K))
    B[I - 1] = B[11]; B[4000] = B[11]       # exact ties across segments
    B[77] = np.nan                          # an item whose scores are NaN never ranks
    A[5] = 0.0                              # a user whose scores are all equal (0): k lowest item ids win
    cu, ci = rng.gamma(1.0, 1.0, U) + 0.1, rng.gamma(1.0, 1.0, I) + 0.1
    if signed:
        cu, ci = rng.standard_normal(U), rng.standard_normal(I)
    ci[I - 1] = ci[4000] = ci[11]
    users = rng.permutation(U)[:333]
    users[0] = 5
    out = {}
    for two_phase in (False, True):
        if two_phase:
            monkeypatch.setenv("PMF_TOPK_TWO_PHASE", "1")
        with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
            ctx.set_array(USER, ARR_FACTOR, A); ctx.set_array(ITEM, ARR_FACTOR, B)
            if mode:
                arr = ARR_BIAS if mode == 1 else ARR_SCALE
                ctx.set_array(USER, arr, cu); ctx.set_array(ITEM, arr, ci)
            out[two_phase] = ctx.topk_items(users, k, use_bias=mode)
    (fi_, fs), (ti_, ts) = out[False], out[True]
    assert fi_.shape == (333, k) and (fi_ >= 0).all() and not (fi_ == 77).any()
    np.testing.assert_allclose(fs, ts, rtol=3e-6, atol=1e-6)
    assert (np.diff(fs, axis=1) <= 0).all()
    tie = np.diff(fs, axis=1) == 0
    assert (np.diff(fi_, axis=1)[tie] > 0).all()
    # user 5: all scores equal (0, or the item bias order for mode 1) -> deterministic ids
    if mode != 1:
        want = [i for i in range(k + 1) if i != 77][:k]
        assert list(fi_[0]) == want and list(ti_[0]) == want
    same = fi_ == ti_
    assert same.mean() > 0.995
    for r, c in zip(*np.nonzero(~same)):    # only near-equal scores may swap
        assert abs(fs[r, c] - ts[r, c]) <= 3e-6 * max(1.0, abs(ts[r, c]))


def test_extended_poisson_top_k_follows_its_own_predict():
    """PoissonMFExtendedCAVI ranks under its scaled score E_phi[u] E_psi[i] theta_u . beta_i."""
    from src.models.poisson_mf_extended_cavi import PoissonMFExtendedCAVI, PoissonMFExtendedCAVIConfig
    d, meta = load_case("poisson_ext_s7_k16")
    train, _ = frames(d)
    m = PoissonMFExtendedCAVI(PoissonMFExtendedCAVIConfig(n_factors=meta["K"], random_state=meta["seed"], max_iter=5,
                                                          tol=None, verbose=False, **meta["base_cfg"]), dtype="f64").fit(train)
    users = np.arange(m.n_users)
    items, scores = m.top_k_items(users, 10)
    full = (m.E_phi[:, None] * m.E_psi[None, :]) * (m.E_theta @ m.E_beta.T)
    assert np.array_equal(items, _rank(full, 10))
    pred = m.predict(np.repeat(users, 10), items.reshape(-1)).reshape(-1, 10)
    np.testing.assert_allclose(scores, pred, rtol=1e-12)
    m32 = PoissonMFExtendedCAVI(PoissonMFExtendedCAVIConfig(n_factors=meta["K"], random_state=meta["seed"], max_iter=5,
                                                            tol=None, verbose=False, **meta["base_cfg"]), dtype="f32").fit(train)
    it32, _ = m32.top_k_items(users, 10)      # the fused fp32 kernel, scale mode
    assert np.mean(it32 == items) > 0.97


@pytest.mark.parametrize("max_blocks", [0, 2])
@pytest.mark.parametrize("U,I,K,k", [(1, 1, 1, 1), (5, 7, 3, 7), (40, 31, 16, 31), (33, 32, 8, 1), (100, 33, 20, 33),
                                      (64, 64, 64, 64), (130, 95, 128, 50), (257, 1000, 100, 64), (700, 63, 64, 10),
                                      (515, 97, 40, 24), (129, 4097, 12, 3)])
def test_fused_topk_edge_shapes_rank_like_numpy(U, I, K, k, max_blocks, monkeypatch):
    """The fused fp32 kernel where its special paths begin and end: fewer items than one 32-item tile (only the partial
    stage), exactly one tile, one item past a tile, k = the item count, k = 1 and k = 64, one user, user counts around the
    32 / 128 tile edges; Gaussian factors (scores of both signs).  `max_blocks` = 2 makes two blocks walk all user tiles
    (the persistent loop with many tiles per block).  Checked against a NumPy ranking of the device's own fp32 tables."""
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ITEM, USER
    if max_blocks:
        monkeypatch.setenv("PMF_TOPK_MAX_BLOCKS", str(max_blocks))
    rng = np.random.default_rng(U * 1000 + I)
    A, B = rng.standard_normal((U, K)), rng.standard_normal((I, K))
    if I > 8:
        B[I - 1] = B[2]                       # an exact tie with the last item
    users = rng.permutation(U)
    with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
        ctx.set_array(USER, ARR_FACTOR, A); ctx.set_array(ITEM, ARR_FACTOR, B)
        items, scores = ctx.topk_items(users, k)
        full = ctx.get_array(USER, ARR_FACTOR)[users] @ ctx.get_array(ITEM, ARR_FACTOR).T
    assert items.shape == (U, k) and (items >= 0).all() and (items < I).all()
    assert all(len(set(row)) == k for row in items)                       # no item twice
    # (signed products cancel: the fp32 sum's error is relative to sum |a_k b_k|, not to the score)
    tol = 3e-7 * (np.abs(A[users]) @ np.abs(B).T).max() + 1e-6
    np.testing.assert_allclose(scores, np.take_along_axis(full, items, axis=1), rtol=0, atol=tol)
    assert (np.diff(scores, axis=1) <= 0).all()
    tie = np.diff(scores, axis=1) == 0
    assert (np.diff(items, axis=1)[tie] > 0).all()
    want = _rank(full, k)
    kth = np.take_along_axis(full, want[:, -1:], axis=1)
    assert (np.take_along_axis(full, items, axis=1) >= kth - 2 * tol).all()
    if k == I:
        assert np.array_equal(np.sort(items, axis=1), np.tile(np.arange(I), (U, 1)))
