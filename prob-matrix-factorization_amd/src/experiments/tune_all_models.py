"""Random search over each model's hyper-parameters on a 50k/10k subsample,
scored by validation MacroMAE; writes `best_hyperparams.txt` in the format
`load_best_hyperparams` reads (reference: src/experiments/tune_all_models.py).

A trial is one small, launch-latency-bound `fit`; the CAVI trials of a model can
therefore run concurrently, each on its own engine context / HIP stream
(`--workers`; ctypes releases the GIL inside every engine call).  The search
spaces, fixed settings (max_iter, tol, epochs) and the selection rule are the
reference's; like the reference the draw is unseeded unless `--seed` is given."""
import argparse
import random
from concurrent.futures import ThreadPoolExecutor
from dataclasses import asdict

import numpy as np
import torch  # before the first engine call: pmf_hip.load() explains the load order (reference: compare_models.py:20 imports it at the top too)

from src.data.load_data import load_all_splits
from src.evaluation.metrics import macro_mae, rmse
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig

_rng = random.Random()
_workers = 1


def load_data():
    print("Loading Data (using load_all_splits)...")
    train_df, val_df, _ = load_all_splits()
    print("Subsampling for tuning speed...")
    return (train_df.sample(n=min(50000, len(train_df)), random_state=42),
            val_df.sample(n=min(10000, len(val_df)), random_state=42))


def _search(title, label, n_trials, draw, trial, describe):
    """Draw all configurations first (sequential, so the sample does not depend on
    scheduling), evaluate them (possibly concurrently), keep the lowest MacroMAE."""
    print(f"\n=== Tuning {title} ===")
    configs = [draw() for _ in range(n_trials)]

    def safe(cfg):
        try:
            return trial(cfg)
        except Exception as exc:
            return exc

    if _workers > 1:
        with ThreadPoolExecutor(max_workers=_workers) as pool:
            results = list(pool.map(safe, configs))
    else:
        results = [safe(c) for c in configs]
    best_score, best_config = float("inf"), None
    for k, (cfg, res) in enumerate(zip(configs, results), start=1):
        if isinstance(res, Exception):
            print(f"Trial {k} failed: {res}")
            continue
        mm, rm = res
        print(f"Trial {k}/{n_trials}: MacroMAE={mm:.4f} (RMSE={rm:.4f}) | {describe(cfg)}")
        if mm < best_score and not np.isnan(mm):
            best_score, best_config = mm, cfg
    print(f"Best {label} MacroMAE: {best_score:.4f}")
    return best_config


def _shift(df, by):
    out = df.copy()
    out["rating"] += by
    return out


def tune_gaussian_mf(train_df, val_df, n_trials=10, verbose=False):
    global_mean = train_df["rating"].mean()
    train_c, val_c = _shift(train_df, -global_mean), _shift(val_df, -global_mean)
    factors, sigma2s, regs = [30, 50, 70], [0.3, 0.5, 0.7], [0.5, 1.0, 2.0]

    def draw():
        return GaussianMFCAVIConfig(n_factors=_rng.choice(factors), sigma2=_rng.choice(sigma2s),
                                    eta_theta2=_rng.choice(regs), eta_beta2=_rng.choice(regs),
                                    eta_bias2=_rng.choice(regs), max_iter=50, tol=1e-3, verbose=verbose,
                                    random_state=42)

    def trial(cfg):
        model = GaussianMFCAVI(cfg).fit(train_c, val_df=val_c, global_mean=global_mean)
        preds = model.predict(val_df["u"].to_numpy(), val_df["i"].to_numpy(), global_mean)
        model.close()
        y = val_df["rating"].to_numpy()
        return macro_mae(y, preds), rmse(y, preds)

    return _search("Gaussian MF (CAVI)", "Gaussian MF", n_trials, draw, trial,
                   lambda c: f"factors={c.n_factors}, s2={c.sigma2}, reg={c.eta_theta2}/{c.eta_beta2}/{c.eta_bias2}")


def tune_poisson_mf(train_df, val_df, n_trials=10, verbose=False):
    def draw():
        return PoissonMFCAVIConfig(n_factors=_rng.choice([10, 20, 40]), a0=_rng.choice([0.05, 0.1, 0.2]),
                                   b0=_rng.choice([0.1, 0.3, 0.5]), max_iter=30, tol=1e-3, verbose=verbose,
                                   random_state=42)

    def trial(cfg):
        model = PoissonMFCAVI(cfg).fit(train_df, val_df=val_df)
        preds = model.predict(val_df["u"].to_numpy(), val_df["i"].to_numpy())
        model.close()
        y = val_df["rating"].to_numpy()
        return macro_mae(y, preds), rmse(y, preds)

    return _search("Poisson MF (CAVI)", "Poisson MF", n_trials, draw, trial,
                   lambda c: f"factors={c.n_factors}, a0={c.a0}, b0={c.b0}")


def tune_hpf_cavi(train_df, val_df, n_trials=10, verbose=False):
    train_s, val_s = _shift(train_df, 1), _shift(val_df, 1)

    def draw():
        k = _rng.choice([10, 20, 30])
        a = _rng.choice([0.1, 0.3, 0.5])
        prime = _rng.choice([3.0, 5.0, 7.0])
        return HPF_CAVI_Config(n_factors=k, a=a, a_prime=prime, b_prime=prime, c=a, c_prime=prime, d_prime=prime,
                               max_iter=50, tol=1e-3, verbose=verbose)

    def trial(cfg):
        model = HPF_CAVI(cfg).fit(train_s, val_df=val_s)
        preds = model.predict(val_s["u"].to_numpy(), val_s["i"].to_numpy()) - 1
        model.close()
        y = val_s["rating"].to_numpy() - 1
        return macro_mae(y, preds), rmse(y, preds)

    return _search("HPF (CAVI)", "HPF CAVI", n_trials, draw, trial,
                   lambda c: f"factors={c.n_factors}, a={c.a}, prime={c.a_prime}")


def tune_hpf_pytorch(train_df, val_df, n_trials=10, verbose=False):
    from src.experiments._full_training import row_counts
    from src.experiments.train_hpf_pytorch_full import adam_epochs, pick_device
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    train_s, val_s = _shift(train_df, 1), _shift(val_df, 1)
    n_users = int(max(train_s["u"].max(), val_s["u"].max())) + 1
    n_items = int(max(train_s["i"].max(), val_s["i"].max())) + 1
    user_counts = row_counts(train_s["u"].to_numpy(), n_users)
    item_counts = row_counts(train_s["i"].to_numpy(), n_items)
    device = pick_device("cpu")
    u = torch.from_numpy(train_s["u"].to_numpy()).long().to(device)
    i = torch.from_numpy(train_s["i"].to_numpy()).long().to(device)
    r = torch.from_numpy(train_s["rating"].to_numpy(dtype=np.float32)).to(device)

    def draw():
        k = _rng.choice([10, 20, 30])
        lr = _rng.choice([0.005, 0.01, 0.02])
        a = _rng.choice([0.5, 1.0, 1.5])
        prime = _rng.choice([0.5, 1.0, 2.0])
        return HPF_PyTorch_Config(n_factors=k, a=a, a_prime=prime, b_prime=prime, c=a, c_prime=prime, d_prime=prime,
                                  lr=lr, epochs=20, verbose=verbose)

    def trial(cfg):
        model = HPF_PyTorch(n_users, n_items, user_counts, item_counts, cfg).to(device)
        adam_epochs(model, u, i, r, cfg.lr, 4096, cfg.epochs, verbose=False)
        model.eval()
        preds = model.predict(val_s["u"].to_numpy(), val_s["i"].to_numpy()) - 1
        y = val_s["rating"].to_numpy() - 1
        return macro_mae(y, preds), rmse(y, preds)

    workers = globals()["_workers"]
    globals()["_workers"] = 1          # torch trials share one default stream: keep them sequential
    try:
        return _search("HPF (PyTorch)", "HPF PyTorch", n_trials, draw, trial,
                       lambda c: f"factors={c.n_factors}, lr={c.lr}, a={c.a}, prime={c.a_prime}")
    finally:
        globals()["_workers"] = workers


def write_best(best, path="best_hyperparams.txt"):
    """`BEST CONFIGURATIONS` header + one `Name: {dict}` line per tuned model
    (reference tune_all_models.py:311-317)."""
    with open(path, "w") as fh:
        fh.write("BEST CONFIGURATIONS\n===================\n")
        for name, cfg in best.items():
            if cfg:
                fh.write(f"{name}: {asdict(cfg)}\n")


def main():
    global _rng, _workers
    parser = argparse.ArgumentParser(description="Tune all models")
    parser.add_argument("--n_trials", type=int, default=5, help="Number of trials per model")
    parser.add_argument("--verbose", action="store_true", help="Enable verbose output")
    parser.add_argument("--workers", type=int, default=4, help="concurrent CAVI trials (engine contexts)")
    parser.add_argument("--seed", type=int, default=None, help="seed of the random search (unseeded by default)")
    args = parser.parse_args()
    _rng, _workers = random.Random(args.seed), max(1, args.workers)
    train_df, val_df = load_data()
    best = {
        "GaussianMF": tune_gaussian_mf(train_df, val_df, n_trials=args.n_trials, verbose=args.verbose),
        "PoissonMF": tune_poisson_mf(train_df, val_df, n_trials=args.n_trials, verbose=args.verbose),
        "HPF_CAVI": tune_hpf_cavi(train_df, val_df, n_trials=args.n_trials, verbose=args.verbose),
        "HPF_PyTorch": tune_hpf_pytorch(train_df, val_df, n_trials=args.n_trials, verbose=args.verbose),
    }
    print("\n\n=== TUNING COMPLETE. BEST CONFIGURATIONS ===")
    for name, cfg in best.items():
        if cfg:
            print(f"{name}: {asdict(cfg)}")
    write_best(best)


if __name__ == "__main__":
    main()
