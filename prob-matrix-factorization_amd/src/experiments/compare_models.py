"""Fit the four models with validation monitoring, time `fit`, and tabulate
train / val / test RMSE and MacroMAE (reference: src/experiments/compare_models.py).
Also home of `load_best_hyperparams`, which the full-training drivers import.
Figures are out of scope; the result table is written to
`model_comparison_results.csv` and the configs to `model_comparison_params.txt`
(same text layout as the reference, compare_models.py:427-432)."""
import ast
import os
import time
import traceback
from dataclasses import asdict

import numpy as np
import torch  # before the first engine call: pmf_hip.load() explains the load order (reference: compare_models.py:20 imports it at the top too)
import pandas as pd

from src.data.load_data import load_all_splits
from src.evaluation.metrics import macro_mae, rmse
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig


def load_best_hyperparams(filepath="best_hyperparams.txt"):
    """`Name: {python dict}` per line; blank lines, lines starting with '=' and
    lines without ':' are skipped (reference compare_models.py:25-47)."""
    if not os.path.exists(filepath):
        print(f"Warning: {filepath} not found. Using default hyperparameters.")
        return {}
    configs = {}
    with open(filepath) as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("=") or ":" not in line:
                continue
            name, _, text = line.partition(":")
            try:
                configs[name.strip()] = ast.literal_eval(text.strip())
            except Exception as exc:
                print(f"Error parsing config for {name.strip()}: {exc}")
    print(f"Loaded hyperparameters from {filepath}")
    return configs


def _row(label, scores, seconds, config):
    (tr, tm), (vr, vm), (sr, sm) = scores
    return {"Model": label, "Train RMSE": tr, "Val RMSE": vr, "Test RMSE": sr, "Train MacroMAE": tm,
            "Val MacroMAE": vm, "Test MacroMAE": sm, "Time (s)": seconds, "Config": str(asdict(config))}


def _timed(tag, fit, note=""):
    # `note`: the reference's lines carry fixed texts here, whatever the configuration says (:86, :301)
    print(f"     [{tag}] Starting training{note}...", flush=True)
    t0 = time.time()
    fit()
    dt = time.time() - t0
    print(f"     [{tag}] Training finished in {dt:.1f}s", flush=True)
    return dt


def run_gaussian_mf(train_df, val_df, test_df, config_dict=None, verbose=False):
    print("  -> Initializing Gaussian MF (Bias)...", flush=True)
    print("     [GaussianMF] Centering data...", flush=True)
    global_mean = train_df["rating"].mean()       # TRAIN mean for all three splits (:54-65)
    centred = []
    for df in (train_df, val_df, test_df):
        df = df.copy()
        df["rating"] -= global_mean
        centred.append(df)
    if config_dict:
        print(f"     [GaussianMF] Using loaded config: {config_dict}")
        config = GaussianMFCAVIConfig(**config_dict)
    else:
        print("     [GaussianMF] Using default config")
        config = GaussianMFCAVIConfig(n_factors=20, sigma2=0.5, eta_theta2=0.1, eta_beta2=0.01, eta_bias2=0.01,
                                      max_iter=100, tol=1e-8, random_state=42, verbose=verbose)
    model = GaussianMFCAVI(config)
    dt = _timed("GaussianMF", lambda: model.fit(centred[0], val_df=centred[1], global_mean=global_mean), " (max_iter=100)")
    print("     [GaussianMF] Evaluating...", flush=True)
    scores = []
    for raw, cen in zip((train_df, val_df, test_df), centred):
        preds = model.predict(cen["u"].to_numpy(), cen["i"].to_numpy(), global_mean)
        scores.append((model.evaluate_rmse(cen, global_mean), macro_mae(raw["rating"].to_numpy(), preds)))
    return _row("Gaussian MF (CAVI)", scores, dt, config)


def run_poisson_mf(train_df, val_df, test_df, config_dict=None, verbose=False):
    print("  -> Initializing Poisson MF (CAVI)...", flush=True)
    if config_dict:
        print(f"     [PoissonMF] Using loaded config: {config_dict}")
        config = PoissonMFCAVIConfig(**config_dict)
    else:
        print("     [PoissonMF] Using default config")
        config = PoissonMFCAVIConfig(n_factors=100, a0=0.1, b0=1.0, max_iter=50, tol=1e-4, random_state=42,
                                     verbose=verbose)
    model = PoissonMFCAVI(config)
    dt = _timed("PoissonMF", lambda: model.fit(train_df, val_df=val_df))
    scores = []
    for df in (train_df, val_df, test_df):
        preds = model.predict(df["u"].to_numpy(), df["i"].to_numpy())
        scores.append((model.evaluate_rmse(df), macro_mae(df["rating"].to_numpy(), preds)))
    return _row("Poisson MF (CAVI)", scores, dt, config)


def _plus_one(*frames):
    out = []
    for df in frames:
        df = df.copy()
        df["rating"] += 1
        out.append(df)
    return out


def _unshifted_scores(predict, frames):
    """Models trained on rating+1: compare on the original scale (:217-221)."""
    scores = []
    for df in frames:
        preds = predict(df["u"].to_numpy(), df["i"].to_numpy())
        y = df["rating"].to_numpy() - 1
        scores.append((rmse(y, preds - 1), macro_mae(y, preds - 1)))
    return scores


def run_hpf_cavi(train_df, val_df, test_df, config_dict=None, verbose=False):
    print("  -> Initializing HPF (CAVI)...", flush=True)
    print("     [HPF_CAVI] Shifting ratings...", flush=True)
    shifted = _plus_one(train_df, val_df, test_df)
    if config_dict:
        print(f"     [HPF_CAVI] Using loaded config: {config_dict}")
        config = HPF_CAVI_Config(**config_dict)
    else:
        print("     [HPF_CAVI] Using default config")
        config = HPF_CAVI_Config(n_factors=50, a=1.0, a_prime=1.0, b_prime=1.0, c=1.0, c_prime=1.0, d_prime=1.0,
                                 max_iter=100, tol=1e-4, random_state=42, verbose=verbose)
    model = HPF_CAVI(config)
    dt = _timed("HPF_CAVI", lambda: model.fit(shifted[0], val_df=shifted[1]))
    return _row("HPF (CAVI)", _unshifted_scores(model.predict, shifted), dt, config)


def run_hpf_pytorch(train_df, val_df, test_df, config_dict=None, verbose=False):
    from src.experiments._full_training import row_counts
    from src.experiments.train_hpf_pytorch_full import adam_epochs, pick_device
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    print("  -> Initializing HPF (PyTorch)...", flush=True)
    print("     [HPF_PyTorch] Shifting ratings...", flush=True)
    shifted = _plus_one(train_df, val_df, test_df)
    # dimensions over all three splits here (:251-252), counts from train only
    n_users = int(max(df["u"].max() for df in (train_df, val_df, test_df))) + 1
    n_items = int(max(df["i"].max() for df in (train_df, val_df, test_df))) + 1
    user_counts = row_counts(train_df["u"].to_numpy(), n_users)
    item_counts = row_counts(train_df["i"].to_numpy(), n_items)
    if config_dict:
        fields = HPF_PyTorch_Config.__annotations__.keys()
        kept = {k: v for k, v in config_dict.items() if k in fields}
        print(f"     [HPF_PyTorch] Using loaded config: {kept}")
        config = HPF_PyTorch_Config(**kept)
    else:
        print("     [HPF_PyTorch] Using default config")
        config = HPF_PyTorch_Config(n_factors=20, a=1.0, a_prime=1.0, b_prime=1.0, c=1.0, c_prime=1.0, d_prime=1.0,
                                    lr=0.01, epochs=50, verbose=verbose)
    device = pick_device(config.device)
    model = HPF_PyTorch(n_users, n_items, user_counts, item_counts, config).to(device)
    u = torch.from_numpy(shifted[0]["u"].to_numpy()).long().to(device)
    i = torch.from_numpy(shifted[0]["i"].to_numpy()).long().to(device)
    r = torch.from_numpy(shifted[0]["rating"].to_numpy(dtype=np.float32)).to(device)
    # the comparison script uses a fixed batch of 4096, not the config's (:299)
    dt = _timed("HPF_PyTorch", lambda: adam_epochs(model, u, i, r, config.lr, 4096, config.epochs, verbose, 5,
                                                   prefix="     [HPF_PyTorch] "), " (epochs=50)")
    model.eval()
    return _row("HPF (PyTorch)", _unshifted_scores(model.predict, shifted), dt, config)


RUNNERS = (("GaussianMF", run_gaussian_mf), ("PoissonMF", run_poisson_mf), ("HPF_CAVI", run_hpf_cavi),
           ("HPF_PyTorch", run_hpf_pytorch))
COLUMNS = ["Model", "Train RMSE", "Val RMSE", "Test RMSE", "Train MacroMAE", "Val MacroMAE", "Test MacroMAE",
           "Time (s)"]


def save_results(results_df):
    results_df[COLUMNS].to_csv("model_comparison_results.csv", index=False)
    with open("model_comparison_params.txt", "w") as fh:
        for _, row in results_df.iterrows():
            fh.write(f"=== {row['Model']} ===\n{row['Config']}\n\n")
    print("Parameters saved to model_comparison_params.txt", flush=True)


def main():
    print("Loading Data (using load_all_splits)...", flush=True)
    try:
        train_df, val_df, test_df = load_all_splits()
    except Exception as exc:
        print(f"Error loading data: {exc}")
        return
    hyperparams = load_best_hyperparams("best_hyperparams.txt")
    results = []
    for key, runner in RUNNERS:
        try:
            results.append(runner(train_df, val_df, test_df, config_dict=hyperparams.get(key), verbose=True))
        except Exception as exc:  # one failing model does not abort the comparison
            print(f"{key} failed: {exc}")
            traceback.print_exc()
    results_df = pd.DataFrame(results)
    print("\n=== FINAL RESULTS ===", flush=True)
    print("\n=== FINAL RESULTS ===", flush=True)          # (twice, as the reference does: compare_models.py:482-483)
    if len(results_df):
        print(results_df[COLUMNS])
        save_results(results_df)


if __name__ == "__main__":
    main()
