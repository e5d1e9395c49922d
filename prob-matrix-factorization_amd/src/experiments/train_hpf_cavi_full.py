"""Full-data HPF (CAVI) training driver (reference: src/experiments/train_hpf_cavi_full.py)."""
from src.experiments import _full_training as ft
from src.experiments.compare_models import load_best_hyperparams
from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config


def train_full_hpf_cavi(dataset_mode="train"):
    print(f"=== Training Full HPF (CAVI) | Mode: {dataset_mode} ===")
    df, test_df = ft.load_frames(dataset_mode)
    print("Shifting ratings by +1 for HPF...")
    shifted = df.copy()
    shifted["rating"] += 1                        # :35-36
    print("Loading best hyperparameters...")
    loaded = load_best_hyperparams().get("HPF_CAVI", {})
    if loaded:
        print(f"Using loaded config: {loaded}")
        config = HPF_CAVI_Config(**loaded)
    else:
        print("Using default config (fallback)")
        config = HPF_CAVI_Config(n_factors=50, a=1.0, a_prime=1.0, b_prime=1.0, c=1.0, c_prime=1.0, d_prime=1.0,
                                 max_iter=100, tol=1e-4, random_state=42, verbose=True)
    model = HPF_CAVI(config)
    ft.timed_fit(lambda: model.fit(shifted))
    ft.write_embeddings("hpf_cavi", model.E_theta, model.E_beta, config)
    print("Generating predictions on Test Set...")
    y_pred = model.predict(test_df["u"].to_numpy(), test_df["i"].to_numpy()) - 1.0   # back to the 0..5 scale (:121-122)
    ft.write_test_predictions("hpf_cavi", test_df, y_pred)
    print("Done.")


if __name__ == "__main__":
    train_full_hpf_cavi(dataset_mode=ft.mode_argument("Train HPF CAVI"))
