"""Seeded synthetic rating generator of SURVEY.md section 8(d): power-law user
and item degrees, ids decorrelated from popularity by a fixed permutation,
duplicates kept, ratings 0..5 = 50/50 blend of the reference data's empirical
mix and a planted rank-16 Poisson signal.  Host NumPy only (data preparation,
not part of the timed path)."""
import numpy as np

RATING_MIX = np.array([0.032, 0.006, 0.012, 0.036, 0.142, 0.772])
BASE_SEED = 20251226


def synth_ratings(n_users, n_items, nnz, seed=BASE_SEED, rank=16, chunk=4_000_000, item_seed=None):
    """Returns (u int32, i int32, rating float64 in 0..5).

    `item_seed` fixes the item side (popularity permutation, planted item factors)
    independently of `seed`: the shards of a multi-GPU run draw different users and
    ratings (seed + rank) over the SAME item catalogue."""
    rng = np.random.default_rng(seed)
    rng_items = rng if item_seed is None else np.random.default_rng(item_seed)
    perm_u = rng.permutation(n_users).astype(np.int32)
    perm_i = rng_items.permutation(n_items).astype(np.int32)
    theta = rng.gamma(0.3, 1.0, size=(n_users, rank)).astype(np.float32)
    beta = rng_items.gamma(0.3, 1.0, size=(n_items, rank)).astype(np.float32)
    # scale so the planted Poisson mean is 4.4 on average
    scale = 4.4 / (float(theta.mean()) * float(beta.mean()) * rank)
    u = np.empty(nnz, dtype=np.int32)
    i = np.empty(nnz, dtype=np.int32)
    r = np.empty(nnz, dtype=np.float64)
    for at in range(0, nnz, chunk):
        n = min(chunk, nnz - at)
        uu = perm_u[np.floor(n_users * rng.random(n) ** 2.0).astype(np.int64)]
        ii = perm_i[np.floor(n_items * rng.random(n) ** 3.0).astype(np.int64)]
        lam = np.einsum("nk,nk->n", theta[uu], beta[ii]) * scale
        planted = np.clip(rng.poisson(lam), 0, 5)
        mix = rng.choice(6, size=n, p=RATING_MIX)
        r[at:at + n] = np.where(rng.random(n) < 0.5, planted, mix)
        u[at:at + n] = uu
        i[at:at + n] = ii
    return u, i, r


def train_val_split(u, i, r, seed=BASE_SEED + 1, train_frac=0.9):
    keep = np.random.default_rng(seed).random(len(u)) < train_frac
    return (u[keep], i[keep], r[keep]), (u[~keep], i[~keep], r[~keep])
