"""Shared body of the four `train_*_full` drivers: load splits -> model-specific
preprocessing -> config from best_hyperparams.txt -> fit -> embeddings / config /
test-prediction files with the reference's names and layouts
(reference: src/experiments/train_gaussian_full.py:16-137 and its three twins)."""
import os
import time
from dataclasses import asdict

import numpy as np
import pandas as pd

from src.data.load_data import load_all_splits
from src.evaluation.metrics import macro_mae, rmse
from src.utils.mapping import get_recipe_id_map

MODES = ("train", "train+val", "full")


def select_training_frame(dataset_mode, train_df, val_df, test_df):
    if dataset_mode == "train":
        return train_df[["u", "i", "rating"]]
    if dataset_mode == "train+val":
        print("Concatenating train and validation sets...")
        return pd.concat([train_df, val_df])[["u", "i", "rating"]]
    if dataset_mode == "full":
        print("Concatenating train, validation, and test sets...")
        return pd.concat([train_df, val_df, test_df])[["u", "i", "rating"]]
    raise ValueError(f"Invalid dataset_mode: {dataset_mode}. Choose from 'train', 'train+val', 'full'.")


def write_embeddings(model_dir, user_emb, item_emb, config, extra_config_text=""):
    """data/embeddings/<model>/{user,item}_embeddings.csv (+ optional leading
    recipe_id column) and config.txt = str(asdict(config))."""
    out = os.path.join("data", "embeddings", model_dir)
    os.makedirs(out, exist_ok=True)
    print(f"Saving embeddings to {out}...")
    pd.DataFrame(user_emb).to_csv(os.path.join(out, "user_embeddings.csv"), index=False)
    items = pd.DataFrame(item_emb)
    id_map = get_recipe_id_map()
    if id_map is not None:
        id_map = id_map[:len(items)]
        if len(id_map) == len(items):
            items.insert(0, "recipe_id", id_map)
        else:
            print("Skipping recipe_id insertion due to size mismatch.")
    items.to_csv(os.path.join(out, "item_embeddings.csv"), index=False)
    with open(os.path.join(out, "config.txt"), "w") as fh:
        fh.write(str(asdict(config)))
        fh.write(extra_config_text)


def write_test_predictions(model_dir, test_df, y_pred):
    """data/predictions/<model>/test_predictions.csv with columns u, i, y_true, y_pred."""
    out = os.path.join("data", "predictions", model_dir)
    os.makedirs(out, exist_ok=True)
    y_true = test_df["rating"].to_numpy()
    print(f"Test Set Metrics: MacroMAE={macro_mae(y_true, y_pred):.4f} | RMSE={rmse(y_true, y_pred):.4f}")
    pd.DataFrame({"u": test_df["u"].to_numpy(), "i": test_df["i"].to_numpy(), "y_true": y_true,
                  "y_pred": y_pred}).to_csv(os.path.join(out, "test_predictions.csv"), index=False)
    print(f"Saved test predictions to {out}")


def timed_fit(fit):
    print("Starting training...")
    t0 = time.time()
    out = fit()
    print(f"Training finished in {time.time() - t0:.1f}s")
    return out


def load_frames(dataset_mode):
    print("Loading data using load_all_splits...")
    train_df, val_df, test_df = load_all_splits()
    return select_training_frame(dataset_mode, train_df, val_df, test_df), test_df


def mode_argument(description):
    import argparse
    parser = argparse.ArgumentParser(description=description)
    parser.add_argument("--dataset_mode", type=str, default="train", choices=list(MODES),
                        help="Which dataset splits to use for training")
    return parser.parse_args().dataset_mode


def row_counts(ids, n):
    counts = np.zeros(n)
    vals, cnt = np.unique(ids, return_counts=True)
    counts[vals] = cnt
    return counts
