"""Property tests (hypothesis) of the host-side sharding logic in pmf_hip.dist and the model base class:
no GPU, no library calls."""
import os
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "prob-matrix-factorization_amd")]


@st.composite
def rating_users(draw):
    n_users = draw(st.integers(1, 60))
    world = draw(st.integers(1, min(8, n_users)))
    n = draw(st.integers(0, 300))
    skew = draw(st.sampled_from([1.0, 2.0, 4.0]))
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    u = np.floor(n_users * rng.random(n) ** skew).astype(np.int64)
    return n_users, world, u, rng


@settings(max_examples=300, deadline=None)
@given(rating_users())
def test_shard_bounds_cover_the_users_with_nonempty_ranges_and_balance_ratings(case):
    from pmf_hip import dist as pdist
    n_users, world, u, _ = case
    b = pdist.shard_bounds(u, n_users, world)
    assert b.shape == (world + 1,) and b[0] == 0 and b[-1] == n_users
    assert (np.diff(b) >= 1).all()                         # every rank owns at least one user
    counts = np.bincount(u, minlength=n_users)
    per_rank = np.array([counts[b[r]:b[r + 1]].sum() for r in range(world)])
    assert per_rank.sum() == len(u)
    # balance: a shard exceeds the ideal share by at most one row's ratings, unless the minimum-one-user
    # rule forced the cut (then the shard is a single user)
    heaviest = counts.max() if len(u) else 0
    for r in range(world):
        assert per_rank[r] <= len(u) / world + heaviest or b[r + 1] - b[r] == 1 or b[r + 1] == n_users - (world - 1 - r)


@settings(max_examples=200, deadline=None)
@given(rating_users())
def test_take_shard_partitions_the_ratings_and_localises_the_ids(case):
    from pmf_hip import dist as pdist
    n_users, world, u, rng = case
    i = rng.integers(0, 17, len(u))
    x = rng.random(len(u))
    b = pdist.shard_bounds(u, n_users, world)
    seen = 0
    for r in range(world):
        lu, li, lx = pdist.take_shard(u, i, x, b, r)
        assert ((lu >= 0) & (lu < b[r + 1] - b[r])).all()
        sel = (u >= b[r]) & (u < b[r + 1])
        # original order kept (the per-row summation order depends on it)
        assert np.array_equal(lu + b[r], u[sel]) and np.array_equal(li, i[sel]) and np.array_equal(lx, x[sel])
        seen += len(lu)
    assert seen == len(u)
    with pytest.raises(ValueError):
        pdist.shard_bounds(u, n_users, n_users + 1)        # fewer users than ranks: refused, on every rank alike


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 2000), st.integers(1, 7), st.integers(0, 2**31 - 1), st.integers(1, 5))
def test_blockwise_rng_consumption_equals_slicing_the_full_draw(n_users, world, seed, width):
    """DeviceModel._user_rows: a rank draws the reference's RNG stream in blocks and keeps only its own users'
    rows -- the rows it keeps, and the state the stream is left in, are those of the full-size draw."""
    from src.models._device_model import DeviceModel
    world = min(world, n_users)
    cuts = np.sort(np.random.default_rng(seed).choice(np.arange(1, n_users), size=world - 1, replace=False)) \
        if world > 1 else np.array([], dtype=np.int64)
    bounds = np.concatenate([[0], cuts, [n_users]]).astype(np.int64)

    class Comm:
        def __init__(self, rank):
            self.rank, self.world = rank, world

    full_rng = np.random.default_rng(seed)
    full = full_rng.gamma(1.0, 0.1, size=(n_users, width))
    after = full_rng.standard_normal(3)
    for rank in range(world):
        m = DeviceModel.__new__(DeviceModel)
        m._comm, m._bounds, m.n_users = (Comm(rank) if world > 1 else None), bounds, n_users
        rng = np.random.default_rng(seed)
        mine = m._user_rows(lambda n: rng.gamma(1.0, 0.1, size=(n, width)))
        lo, hi = (bounds[rank], bounds[rank + 1]) if world > 1 else (0, n_users)
        assert np.array_equal(mine, full[lo:hi])
        assert np.array_equal(rng.standard_normal(3), after)     # the item-side draws that follow are the reference's


@given(st.integers(1, 64), st.integers(0, 2**40))
def test_default_item_chunks_is_bounded_and_single_for_small_messages(world, message_bytes):
    from pmf_hip import dist as pdist
    os.environ.pop("PMF_DIST_CHUNKS", None)
    n = pdist.default_item_chunks(world, message_bytes)
    assert 1 <= n <= 32
    if world <= 1 or 0 < message_bytes < (8 << 20):
        assert n == 1
    if world > 1 and message_bytes >= (64 << 20):
        assert n >= 4
