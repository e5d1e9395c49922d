"""Full-data HPF (PyTorch MAP, Adam) training driver
(reference: src/experiments/train_hpf_pytorch_full.py).  On a ROCm machine the
module and the rating tensors live on the GPU; batches are drawn from a
per-epoch random permutation (what DataLoader(shuffle=True) does) without the
per-sample Python overhead of a Dataset."""
from dataclasses import asdict

import numpy as np
import torch

from src.experiments import _full_training as ft
from src.experiments.compare_models import load_best_hyperparams
from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config


def pick_device(requested="cpu"):
    if torch.cuda.is_available():
        return torch.device("cuda")
    return torch.device(requested if requested else "cpu")


def adam_epochs(model, u, i, r, lr, batch_size, epochs, verbose=True, log_every=1):
    """The reference's external training loop (train_hpf_pytorch_full.py:96-108):
    Adam over all parameters, shuffled minibatches, one pass per epoch."""
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    n = len(r)
    for epoch in range(epochs):
        model.train()
        order = torch.randperm(n, device=r.device)
        total = 0.0
        for at in range(0, n, batch_size):
            idx = order[at:at + batch_size]
            optimizer.zero_grad()
            loss = model.loss(u[idx], i[idx], r[idx])
            loss.backward()
            optimizer.step()
            total += loss.item()
        if verbose and (epoch % log_every == 0 or epoch == epochs - 1):
            print(f"Epoch {epoch + 1}/{epochs} Loss: {total:.4f}")
    return model


def train_full_hpf_pytorch(dataset_mode="train"):
    print(f"=== Training Full HPF (PyTorch) | Mode: {dataset_mode} ===")
    df, test_df = ft.load_frames(dataset_mode)
    print("Shifting ratings by +1 for HPF...")
    shifted = df.copy()
    shifted["rating"] += 1
    n_users, n_items = int(shifted["u"].max()) + 1, int(shifted["i"].max()) + 1   # training frame only (:40-41)
    user_counts = ft.row_counts(shifted["u"].to_numpy(), n_users)
    item_counts = ft.row_counts(shifted["i"].to_numpy(), n_items)
    print("Loading best hyperparameters...")
    raw = load_best_hyperparams().get("HPF_PyTorch", {})
    if raw:
        fields = HPF_PyTorch_Config.__annotations__.keys()
        config = HPF_PyTorch_Config(**{k: v for k, v in raw.items() if k in fields})
        print(f"Using loaded config: {asdict(config)}")
    else:
        print("Using default config (fallback)")
        config = HPF_PyTorch_Config(n_factors=20, a=1.0, a_prime=1.0, b_prime=1.0, c=1.0, c_prime=1.0, d_prime=1.0,
                                    lr=0.01, epochs=50, verbose=True)
    batch_size = raw.get("batch_size", 4096)                                    # (:94)
    device = pick_device(config.device)
    model = HPF_PyTorch(n_users, n_items, user_counts, item_counts, config).to(device)
    u = torch.from_numpy(shifted["u"].to_numpy()).long().to(device)
    i = torch.from_numpy(shifted["i"].to_numpy()).long().to(device)
    r = torch.from_numpy(shifted["rating"].to_numpy(dtype=np.float32)).to(device)
    ft.timed_fit(lambda: adam_epochs(model, u, i, r, config.lr, batch_size, config.epochs, config.verbose))
    model.eval()
    ft.write_embeddings("hpf_pytorch", model.theta.detach().cpu().numpy(), model.beta.detach().cpu().numpy(), config)
    print("Generating predictions on Test Set...")
    keep = (test_df["u"] < n_users) & (test_df["i"] < n_items)
    y_pred = np.zeros(len(test_df))
    y_pred[keep.to_numpy()] = model.predict(test_df["u"].to_numpy()[keep], test_df["i"].to_numpy()[keep])
    ft.write_test_predictions("hpf_pytorch", test_df, y_pred - 1.0)
    print("Done.")


if __name__ == "__main__":
    train_full_hpf_pytorch(dataset_mode=ft.mode_argument("Train HPF PyTorch"))
