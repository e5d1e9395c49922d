"""Grid search for the PyTorch HPF model (reference: src/experiments/tune_hpf_pytorch.py).

Same grid, same fixed hyper-parameters, same protocol -- 10 epochs of Adam at batch 4096 per
combination, validation RMSE after every epoch, the minimum over the epochs is the
combination's score -- and the same printed lines.  The module and the rating tensors live on
the GPU when there is one, and the batches come from a per-epoch permutation instead of a
DataLoader (see train_hpf_pytorch_full.adam_epochs)."""
import itertools

import numpy as np
import torch

from src.data.load_data import load_all_splits
from src.evaluation.metrics import rmse
from src.experiments._full_training import row_counts
from src.experiments.train_hpf_pytorch_full import adam_epochs, pick_device
from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config

PARAM_GRID = {          # tune_hpf_pytorch.py:47-52
    "n_factors": [20, 50],
    "lr": [0.001, 0.005],
    "a": [0.3, 1.0],
    "a_prime": [1.0, 3.0],
}
EPOCHS = 10             # "short run for tuning" (:74)
BATCH_SIZE = 4096       # (:32)


def run_tuning(splits=None, param_grid=None, epochs=EPOCHS, batch_size=BATCH_SIZE):
    """Returns (best_params, best_rmse, [(params, min_val_rmse), ...])."""
    print("Loading data...")
    train_df, val_df, test_df = splits if splits is not None else load_all_splits()
    train_df, val_df = train_df.copy(), val_df.copy()
    train_df["rating"] += 1          # shift ratings by +1 (:27-28)
    val_df["rating"] += 1
    # dimensions over all three splits here (:35-36), unlike the full-training driver
    n_users = int(max(train_df["u"].max(), val_df["u"].max(), test_df["u"].max())) + 1
    n_items = int(max(train_df["i"].max(), val_df["i"].max(), test_df["i"].max())) + 1
    user_counts = row_counts(train_df["u"].to_numpy(), n_users)
    item_counts = row_counts(train_df["i"].to_numpy(), n_items)
    device = pick_device("cpu")
    u = torch.from_numpy(train_df["u"].to_numpy()).long().to(device)
    i = torch.from_numpy(train_df["i"].to_numpy()).long().to(device)
    r = torch.from_numpy(train_df["rating"].to_numpy(dtype=np.float32)).to(device)
    val_u, val_i = val_df["u"].to_numpy(), val_df["i"].to_numpy()
    val_y = val_df["rating"].to_numpy() - 1

    grid = PARAM_GRID if param_grid is None else param_grid
    keys, values = zip(*grid.items())
    combinations = [dict(zip(keys, v)) for v in itertools.product(*values)]
    print(f"Total combinations to test: {len(combinations)}")
    best_rmse, best_config, results = float("inf"), None, []
    for n, params in enumerate(combinations):
        print(f"\n--- Run {n + 1}/{len(combinations)}: {params} ---")
        config = HPF_PyTorch_Config(n_factors=params["n_factors"], a=params["a"], a_prime=params["a_prime"], b_prime=1.0,
                                    c=0.3, c_prime=1.0, d_prime=1.0, lr=params["lr"], epochs=epochs, verbose=False)
        model = HPF_PyTorch(n_users, n_items, user_counts, item_counts, config).to(device)
        seen = []

        def validate(_epoch):
            model.eval()
            seen.append(rmse(val_y, model.predict(val_u, val_i) - 1))

        adam_epochs(model, u, i, r, config.lr, batch_size, config.epochs, verbose=False, on_epoch=validate)
        min_val_rmse = min(seen) if seen else float("inf")
        results.append((params, min_val_rmse))
        print(f"Result RMSE: {min_val_rmse:.4f}")
        if min_val_rmse < best_rmse:
            best_rmse, best_config = min_val_rmse, params
            print(f"*** New Best RMSE: {best_rmse:.4f} ***")
    print(f"\nBest Configuration: {best_config}")
    print(f"Best Validation RMSE: {best_rmse:.4f}")
    return best_config, best_rmse, results


if __name__ == "__main__":
    run_tuning()
