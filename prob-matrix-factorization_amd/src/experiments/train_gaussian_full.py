"""Full-data Gaussian MF training driver (reference: src/experiments/train_gaussian_full.py)."""
from src.experiments import _full_training as ft
from src.experiments.compare_models import load_best_hyperparams
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig


def train_full_gaussian(dataset_mode="train"):
    print(f"=== Training Full Gaussian MF (CAVI) | Mode: {dataset_mode} ===")      # the driver's lines as the reference prints them
    df, test_df = ft.load_frames(dataset_mode)
    global_mean = df["rating"].mean()          # mean of whatever is trained on (:35-38)
    print(f"Centering data (Global Mean = {global_mean:.4f})...")
    centred = df.copy()
    centred["rating"] -= global_mean
    print("Loading best hyperparameters...")
    loaded = load_best_hyperparams().get("GaussianMF", {})
    if loaded:
        print(f"Using loaded config: {loaded}")
        config = GaussianMFCAVIConfig(**loaded)
    else:
        print("Using default config (fallback)")
        config = GaussianMFCAVIConfig(n_factors=50, sigma2=0.5, eta_theta2=0.1, eta_beta2=0.1, eta_bias2=0.1,
                                      max_iter=100, tol=1e-4, random_state=42, verbose=True)
    model = GaussianMFCAVI(config)
    ft.timed_fit(lambda: model.fit(centred, global_mean=global_mean))
    ft.write_embeddings("gaussian_mf", model.m_theta, model.m_beta, config, f"\nglobal_mean: {global_mean}")
    print("Generating predictions on Test Set...")
    y_pred = model.predict(test_df["u"].to_numpy(), test_df["i"].to_numpy(), global_mean=global_mean)
    ft.write_test_predictions("gaussian_mf", test_df, y_pred)
    print("Done.")


if __name__ == "__main__":
    train_full_gaussian(dataset_mode=ft.mode_argument("Train Gaussian MF"))
