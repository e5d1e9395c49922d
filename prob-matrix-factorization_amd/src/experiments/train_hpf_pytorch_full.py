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


def _graphed_step(model, optimizer, batch_size, u, i, r):
    """Capture one full-batch Adam step (loss, backward, update: ~50 small kernels, launch-bound at
    the reference's sizes) in a HIP graph.  Returns (static index buffer, static loss, graph).
    The warm-up iterations PyTorch requires before a capture run on throw-away copies of the
    state, so the captured training is step-for-step the eager one."""
    dev = r.device
    quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if quiet is not None:   # the side-stream warm-up below is the documented capture recipe
        quiet(False)
    saved = [p.detach().clone() for p in model.parameters()]
    static_idx = torch.arange(batch_size, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            optimizer.zero_grad(set_to_none=True)
            model.loss(u[static_idx], i[static_idx], r[static_idx]).backward()
            optimizer.step()
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.no_grad():       # undo the warm-up: parameters and Adam moments back to the start
        for p, q in zip(model.parameters(), saved):
            p.copy_(q)
        for st in optimizer.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
    graph = torch.cuda.CUDAGraph()
    optimizer.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        static_loss = model.loss(u[static_idx], i[static_idx], r[static_idx])
        static_loss.backward()
        optimizer.step()
    # the capture itself executed nothing, but be explicit about the starting point
    return static_idx, static_loss, graph


def adam_epochs(model, u, i, r, lr, batch_size, epochs, verbose=True, log_every=1, graph=None, on_epoch=None, prefix=""):
    """The reference's external training loop (train_hpf_pytorch_full.py:96-108):
    Adam over all parameters, shuffled minibatches, one pass per epoch.

    On a GPU the full-size batches replay one captured HIP graph (`graph=False` or
    PMF_TORCH_GRAPH=0 keeps the eager loop) and the epoch loss is accumulated on the device
    (one read-back per epoch instead of one per step).  `on_epoch(epoch)` runs after every pass (the
    grid search's per-epoch validation, tune_hpf_pytorch.py:83-95)."""
    import os
    on_gpu = r.device.type == "cuda"
    n = len(r)
    if graph is None:
        graph = os.environ.get("PMF_TORCH_GRAPH", "1") != "0"
    graph = bool(graph) and on_gpu and n >= batch_size
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, capturable=graph)
    if graph:
        static_idx, static_loss, step_graph = _graphed_step(model, optimizer, batch_size, u, i, r)
    replays = 0
    for epoch in range(epochs):
        model.train()
        order = torch.randperm(n, device=r.device)
        total = torch.zeros((), device=r.device)
        for at in range(0, n, batch_size):
            idx = order[at:at + batch_size]
            if graph and len(idx) == batch_size:
                static_idx.copy_(idx)
                step_graph.replay()
                total += static_loss
                replays += 1
                continue
            optimizer.zero_grad()
            loss = model.loss(u[idx], i[idx], r[idx])
            loss.backward()
            optimizer.step()
            total += loss.detach()
        if verbose and (epoch % log_every == 0 or epoch == epochs - 1):
            print(f"{prefix}Epoch {epoch + 1}/{epochs} Loss: {total.item():.4f}", flush=bool(prefix))   # (compare_models.py:316 tags and flushes its lines)
        if on_epoch is not None:
            on_epoch(epoch)
    model.training_info_ = {"graph_replays": replays, "steps": epochs * ((n + batch_size - 1) // batch_size)}
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
    ft.timed_fit(lambda: adam_epochs(model, u, i, r, config.lr, batch_size, config.epochs, verbose=True))   # every epoch, whatever config.verbose says (:108)
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
