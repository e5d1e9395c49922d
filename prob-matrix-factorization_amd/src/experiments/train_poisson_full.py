"""Full-data Poisson MF training driver (reference: src/experiments/train_poisson_full.py)."""
from src.experiments import _full_training as ft
from src.experiments.compare_models import load_best_hyperparams
from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig


def train_full_poisson(dataset_mode="train"):
    print(f"=== Training Full Poisson MF (CAVI) | Mode: {dataset_mode} ===")
    df, test_df = ft.load_frames(dataset_mode)     # raw ratings, no preprocessing
    print("Loading best hyperparameters...")
    loaded = load_best_hyperparams().get("PoissonMF", {})
    if loaded:
        print(f"Using loaded config: {loaded}")
        config = PoissonMFCAVIConfig(**loaded)
    else:
        print("Using default config (fallback)")
        config = PoissonMFCAVIConfig(n_factors=50, a0=0.1, b0=1.0, max_iter=100, tol=1e-4, random_state=42,
                                     verbose=True)
    model = PoissonMFCAVI(config)
    ft.timed_fit(lambda: model.fit(df))
    ft.write_embeddings("poisson_mf", model.E_theta, model.E_beta, config)
    print("Generating predictions on Test Set...")
    ft.write_test_predictions("poisson_mf", test_df, model.predict(test_df["u"].to_numpy(), test_df["i"].to_numpy()))
    print("Done.")


if __name__ == "__main__":
    train_full_poisson(dataset_mode=ft.mode_argument("Train Poisson MF"))
