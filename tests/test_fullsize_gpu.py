"""BASELINE-size runs (1M users x 100k items, 50M ratings, K = 64) checked
through size-independent properties and through the oracle on sampled rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U, I, N, K = 1_000_000, 100_000, 50_000_000, 64


@pytest.fixture(scope="module")
def ratings():
    from pmf_hip.synth import synth_ratings
    return synth_ratings(U, I, N, seed=99)


def _positions(ids, rows):
    """{row: positions of its ratings} without a 50M-element argsort"""
    return {int(r): np.nonzero(ids == r)[0] for r in rows}


def _gathered_sums(ctx, side_other, ids, block=4096):
    """sum_j V_other[o_j] and the mean rows m_other[o_j] of one row's ratings, read through
    pmf_get_array_rows (float32 storage -> float64), in blocks so that a 100k-rating row stays small
    on the host.  The covariances are the DEVICE's own rows: a gather that addressed another row would
    sum other matrices than these."""
    from pmf_hip import ARR_COV, ARR_FACTOR
    K = ctx.K
    S = np.zeros((K, K))
    for at in range(0, len(ids), block):
        S += ctx.get_array_rows(side_other, ARR_COV, ids[at:at + block]).sum(axis=0)
    return S, ctx.get_array_rows(side_other, ARR_FACTOR, ids)


def _row_update(S_cov, m_rows, resid, sigma2, eta2):
    """gaussian_mf_cavi_bias.py:146-162 for one row"""
    K = m_rows.shape[1]
    V = np.linalg.inv(np.eye(K) / eta2 + (S_cov + m_rows.T @ m_rows) / sigma2)
    return V, V @ (m_rows * resid[:, None]).sum(axis=0) / sigma2


def _check_rows(ctx, side, rows, pos, other_ids, x, b_self, b_other, sigma2, eta2, tol=2e-3):
    """sampled rows of `side` after a factor half-sweep against NumPy, from the gathered rows' device state"""
    from pmf_hip import ARR_COV, ARR_FACTOR
    other = 1 - side
    got_m = ctx.get_array_rows(side, ARR_FACTOR, rows)
    got_V = ctx.get_array_rows(side, ARR_COV, rows)
    checked = 0
    for k, rr in enumerate(rows):
        sel = pos[int(rr)]
        if len(sel) == 0:
            continue
        o = other_ids[sel].astype(np.int64)
        S_cov, m_rows = _gathered_sums(ctx, other, o)
        V, m = _row_update(S_cov, m_rows, x[sel] - b_self[rr] - b_other[o], sigma2, eta2)
        assert np.max(np.abs(got_V[k] - V)) <= tol * np.abs(V).max(), (side, int(rr), len(sel))
        assert np.max(np.abs(got_m[k] - m)) <= tol * max(np.abs(m).max(), 1e-3), (side, int(rr), len(sel))
        checked += 1
    return checked



def test_hpf_full_size_invariants_and_sampled_rows(ratings):
    """(1) allocation conservation: sum_k (shape[r,k] - prior) = sum of row r's
    ratings whenever no rate was clamped, so the grand total equals sum(x);
    (2) rate sums: sum_r (rate[r,k] - prior_r) = sum_j FACTOR_other[o_j, k];
    (3) sampled rows (incl. the heaviest, split over thousands of chunks)
    against the oracle's per-row update."""
    import pmf_hip
    from oracle import cavi_oracle as orc
    from pmf_hip import ARR_FACTOR, ARR_PRIOR_RATE, ARR_RATE, ARR_SHAPE, ITEM, USER
    u, i, r = ratings
    x = r + 1.0
    rng = np.random.default_rng(0)
    Et = (0.3 + rng.gamma(1.0, 0.1, (U, K))) / (5.0 + rng.gamma(1.0, 0.1, (U, K)))
    Eb = (0.3 + rng.gamma(1.0, 0.1, (I, K))) / (5.0 + rng.gamma(1.0, 0.1, (I, K)))
    xi = 1.0 + rng.random(I)
    with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, Et); ctx.set_array(ITEM, ARR_FACTOR, Eb)
        ctx.set_array(ITEM, ARR_PRIOR_RATE, xi)
        ctx.gamma_sweep(ITEM, 0.3, 0.0, True, 5.0 + K * 0.3, 5.0)
        shape, rate = ctx.get_array(ITEM, ARR_SHAPE), ctx.get_array(ITEM, ARR_RATE)
        factor = ctx.get_array(ITEM, ARR_FACTOR)
    assert np.sum(shape - 0.3) == pytest.approx(np.sum(x), rel=2e-6)
    row_tot = np.bincount(i, weights=x, minlength=I)
    np.testing.assert_allclose((shape - 0.3).sum(axis=1), row_tot, rtol=5e-5, atol=1e-3)
    deg_u = np.bincount(u, minlength=U).astype(np.float64)
    np.testing.assert_allclose((rate - xi[:, None]).sum(axis=0), deg_u @ Et.astype(np.float32).astype(np.float64),
                               rtol=2e-6)
    np.testing.assert_allclose(factor, shape / rate, rtol=2e-6)
    deg_i = np.bincount(i, minlength=I)
    rows = np.concatenate([[int(np.argmax(deg_i))], rng.choice(I, 12, replace=False)])
    order = np.argsort(i, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(deg_i)])
    for rr in rows:
        sel = order[ptr[rr]:ptr[rr + 1]]
        a, b = orc.gamma_half_sweep_rows(Eb[rr:rr + 1], Et, np.array([0, len(sel)]), np.arange(len(sel)),
                                         u[sel].astype(np.int64), x[sel], 0.3, xi[rr])
        np.testing.assert_allclose(shape[rr], a[0], rtol=3e-4)
        np.testing.assert_allclose(rate[rr], b[0], rtol=3e-4)


def test_gaussian_full_size_sampled_rows(ratings):
    """One user half-sweep at full size; sampled users (heaviest included)
    against the oracle's normal equations built from the same item state."""
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER
    u, i, r = ratings
    x = r - r.mean()
    rng = np.random.default_rng(1)
    m_beta = 0.1 * rng.standard_normal((I, K))
    b_item = 0.05 * rng.standard_normal(I)
    b_user = 0.05 * rng.standard_normal(U)
    v_scale = 0.5
    with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, np.zeros((U, K))); ctx.set_array(ITEM, ARR_FACTOR, m_beta)
        ctx.set_cov_identity(USER, 1.0); ctx.set_cov_identity(ITEM, v_scale)
        ctx.set_array(USER, ARR_BIAS, b_user); ctx.set_array(ITEM, ARR_BIAS, b_item)
        ctx.gauss_factor_sweep(USER, 0.3, 0.5)
        m_theta = ctx.get_array(USER, ARR_FACTOR)
        ctx.gauss_bias_sweep(USER, 0.3, 1.0)
        b_new = ctx.get_array(USER, ARR_BIAS)
        deg_u = np.bincount(u, minlength=U)
        rows = np.concatenate([[int(np.argmax(deg_u))], rng.choice(U, 10, replace=False)])
    order = np.argsort(u, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(deg_u)])
    mb32 = m_beta.astype(np.float32).astype(np.float64)
    for rr in rows:
        sel = order[ptr[rr]:ptr[rr + 1]]
        if len(sel) == 0:
            assert not m_theta[rr].any()
            continue
        mo = mb32[i[sel]]
        S = len(sel) * v_scale * np.eye(K) + mo.T @ mo
        V = np.linalg.inv(np.eye(K) / 0.5 + S / 0.3)
        resid = x[sel] - b_user[rr] - b_item[i[sel]]
        want = V @ (mo * resid[:, None]).sum(axis=0) / 0.3
        np.testing.assert_allclose(m_theta[rr], want, rtol=2e-3, atol=2e-5)
        res_b = x[sel] - b_item[i[sel]] - mo @ m_theta[rr]
        var = 1.0 / (1.0 + len(sel) / 0.3)
        assert b_new[rr] == pytest.approx(var / 0.3 * res_b.sum(), rel=2e-3, abs=2e-5)


def test_gaussian_full_size_item_side_heaviest_rows(ratings):
    """Item half-sweep at full size from a synthetic user state (V_theta = c I, random
    means): the 1M-rating item goes through ~2,100 partial chunks, the slot-ordered
    combine and the standalone solve; light items through the fused kernel."""
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER
    u, i, r = ratings
    x = r - r.mean()
    rng = np.random.default_rng(2)
    m_theta = 0.3 * rng.standard_normal((U, K))
    b_user = 0.05 * rng.standard_normal(U)
    b_item = 0.05 * rng.standard_normal(I)
    c_scale = 0.25
    with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, m_theta); ctx.set_array(ITEM, ARR_FACTOR, np.zeros((I, K)))
        ctx.set_cov_identity(USER, c_scale); ctx.set_cov_identity(ITEM, 1.0)
        ctx.set_array(USER, ARR_BIAS, b_user); ctx.set_array(ITEM, ARR_BIAS, b_item)
        ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
        m_beta = ctx.get_array(ITEM, ARR_FACTOR)
        V_beta = ctx.get_array(ITEM, ARR_COV)
    deg_i = np.bincount(i, minlength=I)
    heavy = np.argsort(-deg_i)[:3]
    rows = np.concatenate([heavy, rng.choice(I, 8, replace=False)])
    assert deg_i[heavy[0]] > 500_000
    order = np.argsort(i, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(deg_i)])
    mt32 = m_theta.astype(np.float32).astype(np.float64)
    for rr in rows:
        sel = order[ptr[rr]:ptr[rr + 1]]
        if len(sel) == 0:
            assert not m_beta[rr].any() and np.array_equal(V_beta[rr], np.eye(K))
            continue
        mo = mt32[u[sel]]
        P = np.eye(K) / 0.5 + (len(sel) * c_scale * np.eye(K) + mo.T @ mo) / 0.3
        V = np.linalg.inv(P)
        resid = x[sel] - b_item[rr] - b_user[u[sel]]
        want = V @ (mo * resid[:, None]).sum(axis=0) / 0.3
        # fp32 sums over up to 1M ratings: relative to the row's scale
        assert np.max(np.abs(V_beta[rr] - V)) <= 2e-3 * np.abs(V).max(), (rr, len(sel))
        assert np.max(np.abs(m_beta[rr] - want)) <= 2e-3 * max(np.abs(want).max(), 1e-3), (rr, len(sel))


def test_gaussian_full_size_every_gathered_covariance_differs(ratings):
    """ITEM -> USER -> ITEM factor half-sweeps at full size (gaussian_mf_cavi_bias.py:132-201).  After the first
    sweep every covariance row a kernel gathers is a DIFFERENT matrix, so a gather that addressed another row --
    a 32-bit offset, a wrong stride: the user table is 8.3 GB, rows past 516k start beyond 2^32 bytes -- cannot
    pass, which the c * I tables of the tests above would let through.  Sampled rows of both sides against NumPy
    built from the gathered rows' own device state, read with pmf_get_array_rows."""
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER
    u, i, r = ratings
    x = r - r.mean()
    rng = np.random.default_rng(5)
    m_theta = 0.3 * rng.standard_normal((U, K))
    b_user = 0.05 * rng.standard_normal(U)
    b_item = 0.05 * rng.standard_normal(I)
    deg_u, deg_i = np.bincount(u, minlength=U), np.bincount(i, minlength=I)
    users = np.concatenate([[int(np.argmax(deg_u))], rng.choice(U, 10, replace=False)])
    by_deg = np.argsort(-deg_i)
    heavy = int(by_deg[np.nonzero(deg_i[by_deg] <= 150_000)[0][0]])        # ~300 partial chunks, 100k+ distinct gathers
    split = int(by_deg[np.nonzero(deg_i[by_deg] <= 3_000)[0][0]])          # a split row of a few chunks
    items = np.concatenate([[heavy, split], rng.choice(np.nonzero((deg_i > 0) & (deg_i <= 512))[0], 9, replace=False)])
    pos_u, pos_i = _positions(u, users), _positions(i, items)
    b32 = lambda a: a.astype(np.float32).astype(np.float64)   # what the device holds
    with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
        ctx.set_ratings(u, i, x)
        ctx.set_array(USER, ARR_FACTOR, m_theta); ctx.set_array(ITEM, ARR_FACTOR, np.zeros((I, K)))
        ctx.set_cov_identity(USER, 0.25); ctx.set_cov_identity(ITEM, 1.0)
        ctx.set_array(USER, ARR_BIAS, b_user); ctx.set_array(ITEM, ARR_BIAS, b_item)
        ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)                 # V_beta: one matrix per item from here on
        some = ctx.get_array_rows(ITEM, ARR_COV, items)
        assert len({some[k].tobytes() for k in range(len(items))}) == len(items)
        ctx.gauss_factor_sweep(USER, 0.3, 0.5)                 # gathers 50M distinct-by-item covariance rows
        n_u = _check_rows(ctx, USER, users, pos_u, i, x, b32(b_user), b32(b_item), 0.3, 0.5)
        ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)                 # gathers the users' new covariances from the 8.3 GB table
        n_i = _check_rows(ctx, ITEM, items, pos_i, u, x, b32(b_item), b32(b_user), 0.3, 0.5)
        stride_bytes = ctx.cov_stride * 4
    assert n_u >= 10 and n_i == len(items)
    far = max(int(u[pos_i[int(rr)]].max()) for rr in items) * stride_bytes
    assert far > 2 ** 32                                       # the checked items gathered rows beyond 4 GB


def test_gaussian_k128_shard_size_item_side_fused_and_sharded_paths():
    """BASELINE config C4's per-GPU shard (K = 128, 1.25M users x 1M items, 62.5M ratings; 78 GB of device state).
    ITEM sweep from V_theta = c I (checked against closed-form normal equations), then a USER sweep that gathers
    the resulting per-item covariances out of the 33 GB item table, then the ITEM sweep again -- gathering 62.5M
    DISTINCT rows of the 41 GB user table -- two ways: the fused launch (`gauss_accum_mfma128_kernel<17, fused>`)
    and the path a C4 rank takes (accumulate into the 33.5 GB statistics buffer -> RCCL collective per item chunk ->
    finalize, here over a one-rank communicator).  Sampled rows of every step against NumPy from the gathered
    rows' device state (pmf_get_array_rows); rows past id 260,111 start beyond 2^31 ELEMENTS of their table, the
    sizes at which a 32-bit offset or a wrong stride would show (gaussian_mf_cavi_bias.py:132-201)."""
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER, dist as pdist
    from pmf_hip.engine import Context
    from pmf_hip.synth import synth_ratings
    U2, I2, N2, K2 = 1_250_000, 1_000_000, 62_500_000, 128
    u, i, r = synth_ratings(U2, I2, N2, seed=7)
    x = r - r.mean()
    rng = np.random.default_rng(3)
    m_theta = (0.3 * rng.standard_normal((U2, K2))).astype(np.float32).astype(np.float64)
    b_user = 0.05 * rng.standard_normal(U2)
    b_item = 0.05 * rng.standard_normal(I2)
    b32 = lambda a: a.astype(np.float32).astype(np.float64)
    c_scale = 0.25
    deg_u, deg_i = np.bincount(u, minlength=U2), np.bincount(i, minlength=I2)
    by_deg = np.argsort(-deg_i)
    heavy2 = by_deg[:2]
    mid = int(by_deg[np.nonzero(deg_i[by_deg] <= 20_000)[0][0]])           # ~40 partial chunks of distinct gathers
    light = rng.choice(np.nonzero((deg_i > 0) & (deg_i <= 512))[0], 8, replace=False)
    items = np.concatenate([[mid], light])
    users = np.concatenate([[int(np.argmax(deg_u))], rng.choice(np.nonzero(deg_u > 0)[0], 9, replace=False)])
    first_rows = np.concatenate([heavy2, light[:4]])
    pos_first, pos_i, pos_u = _positions(i, first_rows), _positions(i, items), _positions(u, users)
    comm = pdist.Comm(0, 1, 0, Context.comm_unique_id(), "rccl")
    try:
        with pmf_hip.Context(U2, I2, K2, dtype="f32") as ctx:
            ctx.set_ratings(u, i, x)
            ctx.set_array(USER, ARR_FACTOR, m_theta)
            ctx.set_array(USER, ARR_BIAS, b_user); ctx.set_array(ITEM, ARR_BIAS, b_item)
            ctx.set_cov_identity(USER, c_scale)
            ctx.set_array(ITEM, ARR_FACTOR, np.zeros((I2, K2)))
            ctx.set_cov_identity(ITEM, 1.0)
            # 1. item side from identical user covariances: closed form, the two heaviest items included
            ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
            got = ctx.get_array_rows(ITEM, ARR_FACTOR, first_rows)
            for k, rr in enumerate(first_rows):
                sel = pos_first[int(rr)]
                mo = m_theta[u[sel]]
                P = np.eye(K2) / 0.5 + (len(sel) * c_scale * np.eye(K2) + mo.T @ mo) / 0.3
                want = np.linalg.solve(P, (mo * (x[sel] - b_item[rr] - b_user[u[sel]])[:, None]).sum(axis=0)) / 0.3
                assert np.max(np.abs(got[k] - want)) <= 2e-3 * max(np.abs(want).max(), 1e-3), (int(rr), len(sel))
            # 2. user side: every gathered item covariance is a different matrix now
            ctx.gauss_factor_sweep(USER, 0.3, 0.5)
            n_u = _check_rows(ctx, USER, users, pos_u, i, x, b32(b_user), b32(b_item), 0.3, 0.5)
            # 3. item side again, gathering the users' new covariances: fused, then the C4 rank's three stages
            ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
            n_i = _check_rows(ctx, ITEM, items, pos_i, u, x, b32(b_item), b32(b_user), 0.3, 0.5)
            m_fused = ctx.get_array(ITEM, ARR_FACTOR)
            ctx.set_array(ITEM, ARR_FACTOR, np.zeros((I2, K2)))
            ctx.set_cov_identity(ITEM, 1.0)
            comm.attach(ctx)
            ctx.set_row_chunks(ITEM, 8)
            ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
            n_s = _check_rows(ctx, ITEM, items, pos_i, u, x, b32(b_item), b32(b_user), 0.3, 0.5)
            m_shard = ctx.get_array(ITEM, ARR_FACTOR)
            assert ctx.device_bytes() > 100e9          # the statistics buffer of the sharded path is there
    finally:
        comm.close()
    assert n_u == len(users) and n_i == len(items) and n_s == len(items)
    # same sums, same solve arithmetic up to the separate-launch solve's order of operations
    assert np.max(np.abs(m_fused - m_shard)) <= 2e-4 * max(1.0, np.abs(m_fused).max())
    # the checked rows gathered from beyond 2^31 elements (and 2^32 bytes) of both tables
    stride = K2 * (K2 + 1) // 2
    assert max(int(u[pos_i[int(rr)]].max()) for rr in items) * stride > 2 ** 31
    assert max(int(i[pos_u[int(rr)]].max()) for rr in users) * stride > 2 ** 31
