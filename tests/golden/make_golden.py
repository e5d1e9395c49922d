#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

This script and its sibling `live_reference.py` (the live-comparison helper of
tests/test_live_reference_cpu.py) are the only places in the repo that import the reference
(`/root/reference/src/models/*.py`).  It runs in the build container only --
`/root/reference` does not exist on the GPU box -- and writes small `.npz`
fixtures (inputs + expected outputs, no code) that pin

  * the RNG draw order of every model's initialisation (`max_iter=0`),
  * the full variational state after 1, 3 and 20 CAVI iterations,
  * the per-iteration validation RMSE / MacroMAE trajectory and the
    early-stop iteration for a `tol` that triggers,
  * `predict` on a fixed id list that contains out-of-range ids,
  * `metrics.rmse / mae / macro_mae` on fixed vectors,
  * `HPF_PyTorch.loss` value + gradients on a fixed batch.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
"""
import contextlib
import io
import json
import os
import re
import sys

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.path.insert(0, REF)
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI as RefGaussBias  # noqa: E402
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVIConfig as RefGaussBiasCfg  # noqa: E402
from src.models.gaussian_mf_cavi import GaussianMFCAVI as RefGauss  # noqa: E402
from src.models.gaussian_mf_cavi import GaussianMFCAVIConfig as RefGaussCfg  # noqa: E402
from src.models.poisson_mf_cavi import PoissonMFCAVI as RefPoisson  # noqa: E402
from src.models.poisson_mf_cavi import PoissonMFCAVIConfig as RefPoissonCfg  # noqa: E402
from src.models.hpf_cavi import HPF_CAVI as RefHPF, HPF_CAVI_Config as RefHPFCfg  # noqa: E402
from src.models.poisson_mf_extended_cavi import PoissonMFExtendedCAVI as RefExt  # noqa: E402
from src.models.poisson_mf_extended_cavi import PoissonMFExtendedCAVIConfig as RefExtCfg  # noqa: E402
from src.evaluation import metrics as ref_metrics  # noqa: E402


def tiny_problem(seed, n_users=300, n_items=80, nnz=4000):
    """Skewed tiny rating set with the edge cases the reference code handles:
    empty user rows, empty item rows (max id defines the dimension), duplicate
    (u,i) pairs, validation ids beyond the training dimensions."""
    rng = np.random.default_rng(seed)
    u = np.floor(n_users * rng.random(nnz) ** 2.0).astype(np.int64)
    i = np.floor(n_items * rng.random(nnz) ** 3.0).astype(np.int64)
    pu = rng.permutation(n_users)
    pi = rng.permutation(n_items)
    u, i = pu[u], pi[i]
    # force the dimensions, then punch holes: some ids below the max never occur
    u[0], i[0] = n_users - 1, n_items - 1
    for dead_u in (3, 17, 111):
        u[u == dead_u] = (dead_u + 1) % (n_users - 1)
    for dead_i in (5, 41):
        i[i == dead_i] = (dead_i + 1) % (n_items - 1)
    # duplicates: repeat the first 25 pairs at the end with other ratings
    u[-25:], i[-25:] = u[:25], i[:25]
    mix = np.array([0.032, 0.006, 0.012, 0.036, 0.142, 0.772])
    r = rng.choice(6, size=nnz, p=mix).astype(np.float64)
    is_val = rng.random(nnz) < 0.1
    is_val[0] = False  # keep the dimension-defining row in train
    is_val[-25:] = False
    tr = ~is_val
    train = pd.DataFrame({"u": u[tr], "i": i[tr], "rating": r[tr]})
    val = pd.DataFrame({"u": u[is_val], "i": i[is_val], "rating": r[is_val]})
    # validation rows whose ids lie outside the training dimensions
    extra = pd.DataFrame({"u": [n_users, 5, n_users + 7], "i": [2, n_items, n_items + 1],
                          "rating": [4.0, 5.0, 1.0]})
    val = pd.concat([val, extra], ignore_index=True)
    assert train["u"].max() == n_users - 1 and train["i"].max() == n_items - 1
    return train, val


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


VAL_RE = re.compile(r"Validation RMSE: ([0-9.]+)(?: \| MacroMAE: ([0-9.]+))?")

PRED_U = np.array([0, 1, 2, 299, 300, 5, 17, 111, 250, 1000], dtype=np.int64)
PRED_I = np.array([0, 79, 80, 3, 2, 5, 41, 7, 100, 1], dtype=np.int64)


def state_of(model, kind):
    if kind in ("gauss_bias", "gauss"):
        d = {"m_theta": model.m_theta, "m_beta": model.m_beta,
             "V_theta": model.V_theta, "V_beta": model.V_beta}
        if kind == "gauss_bias":
            d["m_user_bias"] = model.m_user_bias
            d["m_item_bias"] = model.m_item_bias
        return d
    if kind == "poisson":
        return {"a_theta": model.a_theta, "b_theta": model.b_theta,
                "a_beta": model.a_beta, "b_beta": model.b_beta,
                "E_theta": model.E_theta, "E_beta": model.E_beta}
    if kind == "poisson_ext":
        return {k: getattr(model, k) for k in ("a_theta", "b_theta", "a_beta", "b_beta", "a_phi", "b_phi",
                                               "a_psi", "b_psi", "E_theta", "E_beta", "E_phi", "E_psi")}
    if kind == "hpf":
        return {"gamma_a_theta": model.gamma_a_theta, "gamma_b_theta": model.gamma_b_theta,
                "gamma_a_beta": model.gamma_a_beta, "gamma_b_beta": model.gamma_b_beta,
                "gamma_b_xi": model.gamma_b_xi, "gamma_b_eta": model.gamma_b_eta,
                "E_theta": model.E_theta, "E_beta": model.E_beta,
                "E_xi": model.E_xi, "E_eta": model.E_eta,
                "gamma_a_xi": np.float64(model.gamma_a_xi),
                "gamma_a_eta": np.float64(model.gamma_a_eta)}
    raise ValueError(kind)


def make(kind, cfg_kwargs):
    if kind == "gauss_bias":
        return RefGaussBias(RefGaussBiasCfg(**cfg_kwargs))
    if kind == "gauss":
        return RefGauss(RefGaussCfg(**cfg_kwargs))
    if kind == "poisson":
        return RefPoisson(RefPoissonCfg(**cfg_kwargs))
    if kind == "hpf":
        return RefHPF(RefHPFCfg(**cfg_kwargs))
    if kind == "poisson_ext":
        return RefExt(RefExtCfg(**cfg_kwargs))
    raise ValueError(kind)


def preprocess(kind, train, val):
    """Same pre-processing the reference drivers apply before fit
    (compare_models.py:54-65 centring, :180-185 +1 shift)."""
    gm = 0.0
    train, val = train.copy(), val.copy()
    if kind in ("gauss_bias", "gauss"):
        gm = float(train["rating"].mean())
        train["rating"] -= gm
        val["rating"] -= gm
    elif kind == "hpf":
        train["rating"] += 1
        val["rating"] += 1
    return train, val, gm


BASE_CFG = {
    "gauss_bias": dict(sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0),
    "gauss": dict(sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5),
    "poisson": dict(a0=0.1, b0=0.5),
    "hpf": dict(a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0),
    "poisson_ext": dict(a0=0.3, b0=1.0),
}
# tolerances chosen so that the early-stop rule of each model fires mid-run
STOP_TOL = {"gauss_bias": 2e-3, "gauss": 2e-3, "poisson": 2e-3, "hpf": 2e-3, "poisson_ext": 2e-3}


def gen_model_case(kind, seed, K):
    train_raw, val_raw = tiny_problem(1000 + seed)
    train, val, gm = preprocess(kind, train_raw, val_raw)
    out = {
        "train_u": train["u"].to_numpy(), "train_i": train["i"].to_numpy(),
        "train_rating": train["rating"].to_numpy(dtype=float),
        "val_u": val["u"].to_numpy(), "val_i": val["i"].to_numpy(),
        "val_rating": val["rating"].to_numpy(dtype=float),
        "global_mean": np.float64(gm),
        "pred_u": PRED_U, "pred_i": PRED_I,
    }
    is_gauss = kind in ("gauss_bias", "gauss")
    base = dict(BASE_CFG[kind], n_factors=K, random_state=seed, verbose=False)
    meta = {"kind": kind, "seed": seed, "K": K, "base_cfg": BASE_CFG[kind],
            "stop_tol": STOP_TOL[kind], "iters": [0, 1, 3, 20]}
    for n_it in (0, 1, 3, 20):
        # tol that can never trigger: Gaussian stops on 0<=imp<tol, Poisson/HPF on imp<tol
        cfg = dict(base, max_iter=n_it, tol=(0.0 if is_gauss else None))
        m = make(kind, cfg)
        if is_gauss:
            m.fit(train, global_mean=gm)
        else:
            m.fit(train)
        for name, arr in state_of(m, kind).items():
            if name.startswith("V_") and not (K == 8 and n_it in (0, 1, 3)):
                # covariance stacks are large: keep them for K=8 only, plus a trace
                out[f"it{n_it}_{name}_diag"] = np.einsum("nkk->nk", arr)
                continue
            out[f"it{n_it}_{name}"] = np.asarray(arr)
        if n_it == 3:
            if is_gauss:
                out["it3_predict"] = m.predict(PRED_U, PRED_I, gm)
                out["it3_val_rmse"] = np.float64(m.evaluate_rmse(val, gm))
                if kind == "gauss_bias":
                    out["it3_val_macro_mae"] = np.float64(m.evaluate_macro_mae(val, gm))
            else:
                out["it3_predict"] = m.predict(PRED_U, PRED_I)
                out["it3_val_rmse"] = np.float64(m.evaluate_rmse(val))
                if hasattr(m, "evaluate_macro_mae"):
                    out["it3_val_macro_mae"] = np.float64(m.evaluate_macro_mae(val))
    # validation trajectory + early stop (full-precision values re-derived by
    # re-running fit with max_iter = t; the printed ones have 4 decimals)
    cfg = dict(base, max_iter=40, tol=STOP_TOL[kind], verbose=True)
    m = make(kind, cfg)
    if is_gauss:
        _, log = quiet(m.fit, train, val_df=val, global_mean=gm)
    else:
        _, log = quiet(m.fit, train, val_df=val)
    n_done = len(VAL_RE.findall(log))
    stopped = "Early stopping" in log
    traj_rmse, traj_mae = [], []
    for t in range(1, n_done + 1):
        mt = make(kind, dict(base, max_iter=t, tol=(0.0 if is_gauss else None)))
        if is_gauss:
            mt.fit(train, global_mean=gm)
            traj_rmse.append(mt.evaluate_rmse(val, gm))
            traj_mae.append(mt.evaluate_macro_mae(val, gm) if kind == "gauss_bias" else np.nan)
        else:
            mt.fit(train)
            traj_rmse.append(mt.evaluate_rmse(val))
            traj_mae.append(mt.evaluate_macro_mae(val) if hasattr(mt, "evaluate_macro_mae") else np.nan)
    out["stop_val_rmse"] = np.array(traj_rmse)
    out["stop_val_macro_mae"] = np.array(traj_mae)
    out["stop_iterations_run"] = np.int64(n_done)
    out["stop_early"] = np.bool_(stopped)
    out["stop_log"] = np.array(log)
    for name, arr in state_of(m, kind).items():
        if name.startswith("V_"):
            continue
        out[f"stop_{name}"] = np.asarray(arr)
    out["meta"] = np.array(json.dumps(meta))
    return out


# ---- headline / reference-config K ---------------------------------------------------------------
# BASELINE.json quotes K = 64 and K = 128; the reference's own configurations are K = 30 (Gaussian),
# 40 (Poisson) and 20 (HPF) (best_hyperparams.txt:3-5).  A smaller problem (120 users x 48 items,
# 1600 ratings, same edge cases) keeps the reference's (n, K, K) stacks and the fixtures small:
# state after 3 iterations (means / expectations in full, covariances as diagonals plus one full matrix
# per side), predict, and the validation RMSE / MacroMAE after each of 5 iterations (which pins the
# whole state at every iteration through the reference's own evaluate_* functions).
HEADLINE = [("gauss_bias", 30), ("gauss_bias", 64), ("gauss_bias", 128), ("gauss", 64),
            ("poisson", 40), ("poisson", 64), ("hpf", 20), ("hpf", 64),
            # round 3: the two odd tile counts of the 64 < K <= 128 MFMA block sweep (5 and 7 tiles of 16 rows; K = 128 is 8)
            ("gauss_bias", 80), ("gauss_bias", 112)]
HEADLINE_KEYS = {"gauss_bias": ["m_theta", "m_beta", "m_user_bias", "m_item_bias"], "gauss": ["m_theta", "m_beta"],
                 "poisson": ["E_theta", "E_beta", "a_theta", "b_beta"],
                 "hpf": ["E_theta", "E_beta", "E_xi", "E_eta", "gamma_a_theta", "gamma_b_beta", "gamma_b_xi", "gamma_b_eta"]}
HL_PRED_U = np.array([0, 1, 2, 119, 120, 5, 17, 111, 100, 1000], dtype=np.int64)
HL_PRED_I = np.array([0, 47, 48, 3, 2, 5, 41, 7, 100, 1], dtype=np.int64)


def gen_headline_case(kind, K, seed=11):
    train_raw, val_raw = tiny_problem(2000 + K, n_users=120, n_items=48, nnz=1600)
    train, val, gm = preprocess(kind, train_raw, val_raw)
    is_gauss = kind in ("gauss_bias", "gauss")
    out = {"train_u": train["u"].to_numpy(), "train_i": train["i"].to_numpy(),
           "train_rating": train["rating"].to_numpy(dtype=float),
           "val_u": val["u"].to_numpy(), "val_i": val["i"].to_numpy(), "val_rating": val["rating"].to_numpy(dtype=float),
           "global_mean": np.float64(gm), "pred_u": HL_PRED_U, "pred_i": HL_PRED_I}
    base = dict(BASE_CFG[kind], n_factors=K, random_state=seed, verbose=False)
    meta = {"kind": kind, "seed": seed, "K": K, "base_cfg": BASE_CFG[kind], "iters": [3], "traj_iters": 5, "cov_rows": [7]}
    traj_rmse, traj_mae = [], []
    for n_it in range(1, 6):
        m = make(kind, dict(base, max_iter=n_it, tol=(0.0 if is_gauss else None)))
        if is_gauss:
            m.fit(train, global_mean=gm)
            traj_rmse.append(m.evaluate_rmse(val, gm))
            traj_mae.append(m.evaluate_macro_mae(val, gm) if kind == "gauss_bias" else np.nan)
        else:
            m.fit(train)
            traj_rmse.append(m.evaluate_rmse(val))
            traj_mae.append(m.evaluate_macro_mae(val))
        if n_it == 3:
            for key in HEADLINE_KEYS[kind]:
                out[f"it{n_it}_{key}"] = np.asarray(getattr(m, key))
            if is_gauss:
                for side in ("theta", "beta"):
                    V = getattr(m, f"V_{side}")
                    out[f"it{n_it}_V_{side}_diag"] = np.einsum("nkk->nk", V)
                    out[f"it{n_it}_V_{side}_rows"] = V[meta["cov_rows"]]
        if n_it == 3:
            out["it3_predict"] = m.predict(HL_PRED_U, HL_PRED_I, gm) if is_gauss else m.predict(HL_PRED_U, HL_PRED_I)
    out["traj_val_rmse"], out["traj_val_macro_mae"] = np.array(traj_rmse), np.array(traj_mae)
    out["meta"] = np.array(json.dumps(meta))
    return out


def gen_metrics():
    rng = np.random.default_rng(5)
    y_true = rng.choice(6, size=500, p=[0.032, 0.006, 0.012, 0.036, 0.142, 0.772]).astype(float)
    y_pred = y_true + rng.normal(0, 0.8, size=500)
    return {
        "y_true": y_true, "y_pred": y_pred,
        "rmse": np.float64(ref_metrics.rmse(y_true, y_pred)),
        "mae": np.float64(ref_metrics.mae(y_true, y_pred)),
        "macro_mae": np.float64(ref_metrics.macro_mae(y_true, y_pred)),
        # shifted labels, as the HPF drivers produce them (preds - 1 vs raw ratings)
        "macro_mae_centered": np.float64(ref_metrics.macro_mae(y_true - 4.4, y_pred - 4.4)),
    }


def gen_medium():
    """BASELINE config #1 shape (10k x 2k, 200k ratings, K=16), 2 iterations:
    checksums + 512 sampled rows per factor matrix."""
    rng = np.random.default_rng(20251226)
    U, I, N, K = 10_000, 2_000, 200_000, 16
    u = rng.permutation(U)[np.floor(U * rng.random(N) ** 2.0).astype(np.int64)]
    i = rng.permutation(I)[np.floor(I * rng.random(N) ** 3.0).astype(np.int64)]
    r = rng.choice(6, size=N, p=[0.032, 0.006, 0.012, 0.036, 0.142, 0.772]).astype(float)
    out = {"u": u.astype(np.int32), "i": i.astype(np.int32), "rating": r.astype(np.float32)}
    train = pd.DataFrame({"u": u, "i": i, "rating": r})
    rows_u = rng.choice(train["u"].max() + 1, 512, replace=False)
    rows_i = rng.choice(train["i"].max() + 1, 512, replace=False)
    out["rows_u"], out["rows_i"] = rows_u, rows_i
    for kind in ("hpf", "poisson", "gauss_bias"):
        tr, _, gm = preprocess(kind, train, train.iloc[:10])
        is_gauss = kind == "gauss_bias"
        m = make(kind, dict(BASE_CFG[kind], n_factors=K, random_state=42, verbose=False,
                            max_iter=2, tol=(0.0 if is_gauss else None)))
        if is_gauss:
            m.fit(tr, global_mean=gm)
            A, B = m.m_theta, m.m_beta
            out[f"{kind}_bias_u_sum"] = np.float64(m.m_user_bias.sum())
            out[f"{kind}_bias_i_sum"] = np.float64(m.m_item_bias.sum())
            out[f"{kind}_Vtheta_rows"] = m.V_theta[rows_u[:16]]
        else:
            m.fit(tr)
            A, B = m.E_theta, m.E_beta
        out[f"{kind}_A_rows"], out[f"{kind}_B_rows"] = A[rows_u], B[rows_i]
        out[f"{kind}_A_sum"], out[f"{kind}_B_sum"] = np.float64(A.sum()), np.float64(B.sum())
        out[f"{kind}_A_abs"], out[f"{kind}_B_abs"] = np.float64(np.abs(A).sum()), np.float64(np.abs(B).sum())
        out[f"{kind}_global_mean"] = np.float64(gm)
    return out


def gen_hpf_torch():
    import torch
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    train, _ = tiny_problem(1042)
    train = train.copy()
    train["rating"] += 1
    U, I = int(train["u"].max()) + 1, int(train["i"].max()) + 1
    uc = np.bincount(train["u"], minlength=U)
    ic = np.bincount(train["i"], minlength=I)
    cfg = HPF_PyTorch_Config(n_factors=8, a=0.3, a_prime=1.5, b_prime=2.0, c=0.4,
                             c_prime=2.5, d_prime=0.7, verbose=False)
    torch.manual_seed(0)
    m = HPF_PyTorch(U, I, uc, ic, cfg)
    bu = torch.from_numpy(train["u"].to_numpy()[:256])
    bi = torch.from_numpy(train["i"].to_numpy()[:256])
    br = torch.from_numpy(train["rating"].to_numpy(dtype=np.float32)[:256])
    loss = m.loss(bu, bi, br)
    loss.backward()
    return {
        "user_counts": uc, "item_counts": ic, "n_users": np.int64(U), "n_items": np.int64(I),
        "cfg": np.array(json.dumps(dict(n_factors=8, a=0.3, a_prime=1.5, b_prime=2.0, c=0.4,
                                        c_prime=2.5, d_prime=0.7))),
        "theta_uncons": m.theta_uncons.detach().numpy(), "beta_uncons": m.beta_uncons.detach().numpy(),
        "xi_uncons": m.xi_uncons.detach().numpy(), "eta_uncons": m.eta_uncons.detach().numpy(),
        "batch_u": bu.numpy(), "batch_i": bi.numpy(), "batch_r": br.numpy(),
        "loss": np.float64(loss.item()),
        "grad_theta": m.theta_uncons.grad.numpy(), "grad_beta": m.beta_uncons.grad.numpy(),
        "grad_xi": m.xi_uncons.grad.numpy(), "grad_eta": m.eta_uncons.grad.numpy(),
        "predict": m.predict(PRED_U[:4].copy() % U, PRED_I[:4].copy() % I),
    }


def gen_config5():
    """BASELINE config #5: the reference's Poisson MF, K = 64, a0 = 0.1, b0 = 0.5
    (best_hyperparams.txt:4), 150 iterations, on the recipe-shaped stand-in (tests/helpers.py:recipe_standin,
    train + validation rows as the full-training driver uses them).  About a minute of the reference's loop.
    Stored: test-set predictions and RMSE, 300 sampled users' factor rows and top-11 item lists with scores,
    checksums of both factor matrices."""
    here = os.path.dirname(OUT)
    sys.path.append(here)                                                        # tests/helpers.py
    sys.path.append(os.path.join(os.path.dirname(here), "prob-matrix-factorization_amd"))   # pmf_hip.synth only: appended, so `src` stays the reference's
    from helpers import recipe_standin
    train, val, test = recipe_standin()
    tr = pd.concat([train, val])
    cfg = dict(n_factors=64, a0=0.1, b0=0.5, max_iter=150, tol=None, random_state=42, verbose=False)
    m, _ = quiet(lambda: make("poisson", cfg).fit(tr))
    pred = m.predict(test["u"].to_numpy(), test["i"].to_numpy())
    users = np.random.default_rng(1).choice(m.n_users, 300, replace=False)
    early, _ = quiet(lambda: make("poisson", dict(cfg, max_iter=20)).fit(tr))      # a cheaper pin for the CPU oracle test
    scores = m.E_theta[users] @ m.E_beta.T
    top = np.argsort(-scores, axis=1, kind="stable")[:, :11]
    return {"cfg": np.array(json.dumps({k: v for k, v in cfg.items() if k != "verbose"})),
            "n_train_rows": np.int64(len(tr)), "n_test_rows": np.int64(len(test)),
            "test_pred": pred, "test_rmse": np.float64(ref_metrics.rmse(test["rating"].to_numpy(dtype=float), pred)),
            "users": users, "E_theta_rows": m.E_theta[users], "top11": top.astype(np.int32),
            "top11_scores": np.take_along_axis(scores, top, axis=1),
            "E_theta_sum": np.float64(m.E_theta.sum()), "E_beta_sum": np.float64(m.E_beta.sum()),
            "test_pred_it20": early.predict(test["u"].to_numpy(), test["i"].to_numpy()),
            "E_theta_rows_it20": early.E_theta[users]}


def main():
    only = sys.argv[1:]
    if "config5" in only:
        path = os.path.join(OUT, "config5_poisson.npz")
        np.savez_compressed(path, **gen_config5())
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")
        return
    if not only or "headline" in only:
        for kind, K in HEADLINE:
            if "new" in only and os.path.exists(os.path.join(OUT, f"hk_{kind}_k{K}.npz")):
                continue        # `headline new`: only the cases that have no file yet (existing fixtures stay byte-identical)
            path = os.path.join(OUT, f"hk_{kind}_k{K}.npz")
            np.savez_compressed(path, **gen_headline_case(kind, K))
            print("wrote", path, os.path.getsize(path) // 1024, "KiB")
        if only:
            return
    for kind in ("hpf", "poisson", "gauss_bias", "gauss", "poisson_ext"):
        if only and kind not in only:
            continue
        for seed, K in ((42, 8), (7, 16)):
            path = os.path.join(OUT, f"{kind}_s{seed}_k{K}.npz")
            np.savez_compressed(path, **gen_model_case(kind, seed, K))
            print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    if only:
        return
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **gen_metrics())
    np.savez_compressed(os.path.join(OUT, "medium_c1.npz"), **gen_medium())
    np.savez_compressed(os.path.join(OUT, "hpf_torch.npz"), **gen_hpf_torch())
    print("done")


if __name__ == "__main__":
    main()
