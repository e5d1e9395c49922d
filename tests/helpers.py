"""Shared helpers for the GPU parity tests."""
import json
import os

import numpy as np
import pandas as pd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return d, json.loads(str(d["meta"]))


def frames(d):
    train = pd.DataFrame({"u": d["train_u"], "i": d["train_i"], "rating": d["train_rating"]})
    val = pd.DataFrame({"u": d["val_u"], "i": d["val_i"], "rating": d["val_rating"]})
    return train, val


def skewed_problem(seed, U, I, N, rating_kind="count"):
    """Power-law rows (SURVEY.md section 8d generator): heavy head rows exercise
    the split-row path, ids never drawn give empty rows."""
    rng = np.random.default_rng(seed)
    u = rng.permutation(U)[np.floor(U * rng.random(N) ** 2.0).astype(np.int64)]
    i = rng.permutation(I)[np.floor(I * rng.random(N) ** 3.0).astype(np.int64)]
    # ids that never occur -> empty rows (the last id stays: it defines the dimension)
    for dead in (1, U // 2, U - 2):
        u[u == dead] = 0
    for dead in (2, I // 3):
        i[i == dead] = 0
    u[0], i[0] = U - 1, I - 1
    r = rng.choice(6, size=N, p=[0.032, 0.006, 0.012, 0.036, 0.142, 0.772]).astype(np.float64)
    if rating_kind == "count":
        r += 1.0
    elif rating_kind == "centered":
        r -= r.mean()
    return u, i, r


def rel_err(a, b, floor=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor))) if a.size else 0.0


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0


# ---- caller-owned device statistics buffers (the bring-your-own-collective form of the C-ABI:
# pmf_*_accumulate / pmf_*_finalize) -- torch is test infrastructure here, the product path
# (pmf_hip.dist.Comm) keeps its buffers inside libpmf_hip.so
class DeviceStats:
    def __init__(self, n_elems, np_dtype, device):
        import torch
        tdt = torch.float64 if np_dtype == np.float64 else torch.float32
        self.tensor = torch.zeros(int(n_elems), dtype=tdt, device=device)
        self.ptr = self.tensor.data_ptr()


def gamma_stats(ctx, device):
    return DeviceStats(ctx.n_items * 2 * ctx.kpad, ctx.np_dtype, device)


def sgd_stats(ctx, device):
    return DeviceStats(ctx.n_items * ctx.sgd_stats_width, ctx.np_dtype, device)


def gauss_stats(ctx, device):
    return (DeviceStats(ctx.n_items * (ctx.cov_stride + ctx.kpad), ctx.np_dtype, device),
            DeviceStats(ctx.n_items * 2, ctx.np_dtype, device))


def recipe_standin(n_users=11_780, n_items=13_000, nnz=320_000):
    """The synthetic stand-in for the processed Food.com data (BASELINE config #5 shape; the real CSVs cannot be
    fetched offline): (train, validation, test) frames.  Deterministic -- tests/golden/config5_poisson.npz holds
    what the REFERENCE computes on exactly these frames."""
    from pmf_hip.synth import synth_ratings
    u, i, r = synth_ratings(n_users, n_items, nnz, seed=5)
    u[0], i[0] = n_users - 1, n_items - 1
    part = np.random.default_rng(0).choice(3, size=len(u), p=[0.85, 0.075, 0.075])
    part[0] = 0
    return tuple(pd.DataFrame({"u": u[part == k], "i": i[part == k], "rating": r[part == k]}) for k in range(3))
