"""Single-GPU check of the multi-GPU kernels: W logical user-range shards on
one device, item statistics summed the way the RCCL all-reduce would, must
reproduce the unsharded (fused-kernel) result."""
import numpy as np
import pytest

from helpers import gamma_stats, gauss_stats, max_abs, rel_err, skewed_problem

pytestmark = pytest.mark.gpu


def _shards(u, i, x, U, W):
    from pmf_hip import dist as pdist
    b = pdist.shard_bounds(u, U, W)
    return b, [pdist.take_shard(u, i, x, b, g) for g in range(W)]


def _allreduce(stats):
    import torch
    torch.cuda.synchronize()   # the contexts run on their own streams
    total = stats[0].tensor.clone()
    for s in stats[1:]:
        total += s.tensor
    for s in stats:
        s.tensor.copy_(total)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-11), ("f32", 3e-5)])
@pytest.mark.parametrize("K", [16, 64])
def test_hpf_logical_shards_equal_single_context(K, dtype, tol):
    import torch
    import pmf_hip
    from oracle import cavi_oracle as orc
    from pmf_hip import ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER, dist as pdist
    U, I, N, W = 2000, 300, 50000, 3
    u, i, x = skewed_problem(3, U, I, N)
    st = orc.init_hpf(U, I, K, 0.3, 5.0, 5.0, 0.3, 5.0, 5.0, seed=3)
    up = (0.3, 0.0, True, st["gamma_a_xi"], 5.0)
    ip = (0.3, 0.0, True, st["gamma_a_eta"], 5.0)
    dev = torch.device("cuda", 0)

    def new_ctx(n_users, uu, ii, xx, lo, hi):
        c = pmf_hip.Context(n_users, I, K, dtype=dtype)
        c.set_ratings(uu, ii, xx)
        c.set_array(USER, ARR_FACTOR, st["E_theta"][lo:hi]); c.set_array(ITEM, ARR_FACTOR, st["E_beta"])
        c.set_array(USER, ARR_PRIOR_RATE, st["E_xi"][lo:hi]); c.set_array(ITEM, ARR_PRIOR_RATE, st["E_eta"])
        return c

    ref = new_ctx(U, u, i, x, 0, U)
    b, parts = _shards(u, i, x, U, W)
    ctxs = [new_ctx(int(b[g + 1] - b[g]), *parts[g], int(b[g]), int(b[g + 1])) for g in range(W)]
    stats = [gamma_stats(c, dev) for c in ctxs]
    for _ in range(3):
        pdist.gamma_iteration(ref, None, None, up, ip)
        for c in ctxs:
            c.gamma_sweep(USER, *up)
        for c, s in zip(ctxs, stats):
            c.gamma_accumulate(ITEM, s.ptr)
        torch.cuda.synchronize()
        _allreduce(stats)
        for c, s in zip(ctxs, stats):
            c.gamma_finalize(ITEM, s.ptr, *ip)
    want_u, want_i = ref.get_array(USER, ARR_FACTOR), ref.get_array(ITEM, ARR_FACTOR)
    got_u = np.concatenate([c.get_array(USER, ARR_FACTOR) for c in ctxs])
    assert rel_err(got_u, want_u) <= tol
    for c in ctxs:
        assert rel_err(c.get_array(ITEM, ARR_FACTOR), want_i) <= tol
        assert rel_err(c.get_array(ITEM, ARR_PRIOR_RATE), ref.get_array(ITEM, ARR_PRIOR_RATE)) <= tol
    assert np.array_equal(ctxs[0].get_array(ITEM, ARR_FACTOR), ctxs[1].get_array(ITEM, ARR_FACTOR))


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-10), ("f32", 2e-4)])
@pytest.mark.parametrize("K", [16, 64, 96, 128])
def test_gaussian_logical_shards_equal_single_context(K, dtype, tol):
    import torch
    import pmf_hip
    from oracle import cavi_oracle as orc
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER, dist as pdist
    U, I, N, W = 1500, 200, 30000, 3
    u, i, x = skewed_problem(4, U, I, N, rating_kind="centered")
    st = orc.init_gaussian(U, I, K, seed=2, bias=True)
    dev = torch.device("cuda", 0)

    def new_ctx(n_users, uu, ii, xx, lo, hi):
        c = pmf_hip.Context(n_users, I, K, dtype=dtype)
        c.set_ratings(uu, ii, xx)
        c.set_array(USER, ARR_FACTOR, st["m_theta"][lo:hi]); c.set_array(ITEM, ARR_FACTOR, st["m_beta"])
        c.set_cov_identity(USER); c.set_cov_identity(ITEM)
        c.set_array(USER, ARR_BIAS, np.zeros(n_users)); c.set_array(ITEM, ARR_BIAS, np.zeros(I))
        return c

    ref = new_ctx(U, u, i, x, 0, U)
    b, parts = _shards(u, i, x, U, W)
    ctxs = [new_ctx(int(b[g + 1] - b[g]), *parts[g], int(b[g]), int(b[g + 1])) for g in range(W)]
    stats = [gauss_stats(c, dev) for c in ctxs]
    for _ in range(2):
        pdist.gaussian_iteration(ref, None, None, None, 0.3, 0.5, 0.5, 1.0)
        for c in ctxs:
            c.gauss_factor_sweep(USER, 0.3, 0.5)
        for c, (s, _) in zip(ctxs, stats):
            c.gauss_factor_accumulate(ITEM, s.ptr)
        torch.cuda.synchronize()
        _allreduce([s for s, _ in stats])
        for c, (s, _) in zip(ctxs, stats):
            c.gauss_factor_finalize(ITEM, s.ptr, 0.3, 0.5)
        for c in ctxs:
            c.gauss_bias_sweep(USER, 0.3, 1.0)
        for c, (_, sb) in zip(ctxs, stats):
            c.gauss_bias_accumulate(ITEM, sb.ptr)
        torch.cuda.synchronize()
        _allreduce([sb for _, sb in stats])
        for c, (_, sb) in zip(ctxs, stats):
            c.gauss_bias_finalize(ITEM, sb.ptr, 0.3, 1.0)
    got_u = np.concatenate([c.get_array(USER, ARR_FACTOR) for c in ctxs])
    assert max_abs(got_u, ref.get_array(USER, ARR_FACTOR)) <= tol
    assert max_abs(np.concatenate([c.get_array(USER, ARR_BIAS) for c in ctxs]), ref.get_array(USER, ARR_BIAS)) <= tol
    for c in ctxs:
        assert max_abs(c.get_array(ITEM, ARR_FACTOR), ref.get_array(ITEM, ARR_FACTOR)) <= tol
        assert max_abs(c.get_array(ITEM, ARR_BIAS), ref.get_array(ITEM, ARR_BIAS)) <= tol
        assert max_abs(c.get_array(ITEM, ARR_COV), ref.get_array(ITEM, ARR_COV)) <= tol


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("K", [16, 64])
def test_hpf_item_row_chunks_do_not_change_the_result(K, dtype):
    """accumulate / finalize chunk by chunk (the pipelined multi-GPU item half-sweep) is
    bit-identical to the unchunked accumulate / finalize, heavy (split) rows included."""
    import torch
    import pmf_hip
    from oracle import cavi_oracle as orc
    from pmf_hip import ARR_FACTOR, ARR_HYPER_RATE, ARR_PRIOR_RATE, ARR_RATE, ARR_SHAPE, ITEM, USER, dist as pdist
    U, I, N = 3000, 110, 60000
    u, i, x = skewed_problem(5, U, I, N)
    assert np.bincount(i, minlength=I).max() > 600          # rows split over several tasks
    st = orc.init_hpf(U, I, K, 0.3, 5.0, 5.0, 0.3, 5.0, 5.0, seed=3)
    up = (0.3, 0.0, True, st["gamma_a_xi"], 5.0)
    ip = (0.3, 0.0, True, st["gamma_a_eta"], 5.0)
    dev = torch.device("cuda", 0)
    out = []
    for chunks in (1, 4, 7):
        c = pmf_hip.Context(U, I, K, dtype=dtype)
        c.set_ratings(u, i, x)
        c.set_row_chunks(ITEM, chunks)                       # after set_ratings: lists are rebuilt
        c.set_array(USER, ARR_FACTOR, st["E_theta"]); c.set_array(ITEM, ARR_FACTOR, st["E_beta"])
        c.set_array(USER, ARR_PRIOR_RATE, st["E_xi"]); c.set_array(ITEM, ARR_PRIOR_RATE, st["E_eta"])
        s = gamma_stats(c, dev)
        for _ in range(2):
            c.gamma_sweep(USER, *up)
            for k in range(c.n_chunks[ITEM]):
                c.select_chunk(ITEM, k)
                c.gamma_accumulate(ITEM, s.ptr)
            for k in reversed(range(c.n_chunks[ITEM])):
                c.select_chunk(ITEM, k)
                c.gamma_finalize(ITEM, s.ptr, *ip)
            c.select_chunk(ITEM, -1)
        out.append([c.get_array(sd, a) for sd in (USER, ITEM)
                    for a in (ARR_FACTOR, ARR_SHAPE, ARR_RATE, ARR_PRIOR_RATE, ARR_HYPER_RATE)])
        lo, hi = c.chunk_rows(ITEM, c.n_chunks[ITEM] - 1)
        assert hi == I and 0 <= lo < hi
        with pytest.raises(pmf_hip.PmfError):
            c.select_chunk(ITEM, c.n_chunks[ITEM])
        c.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("K", [16, 64, 96, 128])
def test_gaussian_item_row_chunks_do_not_change_the_result(K, dtype):
    import torch
    import pmf_hip
    from oracle import cavi_oracle as orc
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER, dist as pdist
    U, I, N = 3000, 110, 60000
    u, i, x = skewed_problem(6, U, I, N, rating_kind="centered")
    assert np.bincount(i, minlength=I).max() > 600
    st = orc.init_gaussian(U, I, K, seed=2, bias=True)
    dev = torch.device("cuda", 0)
    out = []
    for chunks in (1, 4, 7):
        c = pmf_hip.Context(U, I, K, dtype=dtype)
        c.set_row_chunks(ITEM, chunks)                       # before set_ratings
        c.set_ratings(u, i, x)
        c.set_array(USER, ARR_FACTOR, st["m_theta"]); c.set_array(ITEM, ARR_FACTOR, st["m_beta"])
        c.set_cov_identity(USER); c.set_cov_identity(ITEM)
        c.set_array(USER, ARR_BIAS, np.zeros(U)); c.set_array(ITEM, ARR_BIAS, np.zeros(I))
        s, sb = gauss_stats(c, dev)
        for _ in range(2):
            c.gauss_factor_sweep(USER, 0.3, 0.5)
            for k in range(c.n_chunks[ITEM]):
                c.select_chunk(ITEM, k)
                c.gauss_factor_accumulate(ITEM, s.ptr)
            for k in range(c.n_chunks[ITEM]):
                c.select_chunk(ITEM, k)
                c.gauss_factor_finalize(ITEM, s.ptr, 0.3, 0.5)
            c.gauss_bias_sweep(USER, 0.3, 1.0)
            for k in range(c.n_chunks[ITEM]):
                c.select_chunk(ITEM, k)
                c.gauss_bias_accumulate(ITEM, sb.ptr)
            for k in range(c.n_chunks[ITEM]):
                c.select_chunk(ITEM, k)
                c.gauss_bias_finalize(ITEM, sb.ptr, 0.3, 1.0)
            c.select_chunk(ITEM, -1)
        out.append([c.get_array(sd, a) for sd in (USER, ITEM) for a in (ARR_FACTOR, ARR_COV, ARR_BIAS)])
        c.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)
