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


def test_gaussian_k128_shard_size_item_side_fused_and_sharded_paths():
    """BASELINE config C4's per-GPU shard (K = 128, 1.25M users x 1M items, 62.5M ratings; 78 GB of device state):
    one item half-sweep (gaussian_mf_cavi_bias.py:170-201) two ways -- the fused launch
    (`gauss_accum_mfma128_kernel<17, fused>`) and the multi-GPU path a C4 rank takes (accumulate into the 33.5 GB
    statistics buffer -> RCCL all-reduce per item chunk -> finalize, here over a one-rank communicator) -- against
    each other and, on sampled items (the heaviest included), against the NumPy normal equations."""
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_FACTOR, ITEM, USER, dist as pdist
    from pmf_hip.engine import Context
    from pmf_hip.synth import synth_ratings
    U2, I2, N2, K2 = 1_250_000, 1_000_000, 62_500_000, 128
    u, i, r = synth_ratings(U2, I2, N2, seed=7)
    x = r - r.mean()
    rng = np.random.default_rng(3)
    m_theta = (0.3 * rng.standard_normal((U2, K2))).astype(np.float32).astype(np.float64)
    b_user = 0.05 * rng.standard_normal(U2)
    b_item = 0.05 * rng.standard_normal(I2)
    c_scale = 0.25
    comm = pdist.Comm(0, 1, 0, Context.comm_unique_id(), "rccl")
    try:
        with pmf_hip.Context(U2, I2, K2, dtype="f32") as ctx:
            ctx.set_ratings(u, i, x)
            ctx.set_array(USER, ARR_FACTOR, m_theta)
            ctx.set_array(USER, ARR_BIAS, b_user); ctx.set_array(ITEM, ARR_BIAS, b_item)
            ctx.set_cov_identity(USER, c_scale)
            out = []
            for sharded in (False, True):
                ctx.set_array(ITEM, ARR_FACTOR, np.zeros((I2, K2)))
                ctx.set_cov_identity(ITEM, 1.0)
                if sharded:
                    comm.attach(ctx)
                    ctx.set_row_chunks(ITEM, 8)
                ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
                out.append(ctx.get_array(ITEM, ARR_FACTOR))
            assert ctx.device_bytes() > 100e9          # the statistics buffer of the sharded path is there
    finally:
        comm.close()
    m_fused, m_shard = out
    # same sums, same solve arithmetic up to the separate-launch solve's order of operations
    assert np.max(np.abs(m_fused - m_shard)) <= 2e-4 * max(1.0, np.abs(m_fused).max())
    deg_i = np.bincount(i, minlength=I2)
    heavy = np.argsort(-deg_i)[:2]
    rows = np.concatenate([heavy, rng.choice(I2, 6, replace=False)])
    pos = {int(rr): np.nonzero(i == rr)[0] for rr in rows}      # (no 62.5M-element argsort needed for 8 rows)
    for rr in rows:
        sel = pos[int(rr)]
        if len(sel) == 0:
            assert not m_fused[rr].any() and not m_shard[rr].any()
            continue
        mo = m_theta[u[sel]]
        P = np.eye(K2) / 0.5 + (len(sel) * c_scale * np.eye(K2) + mo.T @ mo) / 0.3
        want = np.linalg.solve(P, (mo * (x[sel] - b_item[rr] - b_user[u[sel]])[:, None]).sum(axis=0)) / 0.3
        for got in (m_fused, m_shard):
            assert np.max(np.abs(got[rr] - want)) <= 2e-3 * max(np.abs(want).max(), 1e-3), (rr, len(sel))
