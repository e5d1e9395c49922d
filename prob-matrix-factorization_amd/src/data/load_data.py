"""Loaders for the processed interaction CSVs (columns u, i, rating) with the
reference's function names (reference: src/data/load_data.py:93-135).  The raw
Kaggle download and its one-off pandas preprocessing are out of scope here."""
import os

import pandas as pd

DATA_DIR = "data/processed"


def load_interactions(split):
    """split in {'train', 'validation', 'test'} -> DataFrame[u, i, rating]."""
    path = os.path.join(DATA_DIR, f"interactions_{split}.csv")
    if not os.path.exists(path):
        raise FileNotFoundError(f"File not found: {path}")
    return pd.read_csv(path)[["u", "i", "rating"]]


def load_all_splits():
    return tuple(load_interactions(s) for s in ("train", "validation", "test"))


def load_all_splits_centered():
    """(train, val, test) with the TRAIN mean subtracted from every rating, plus that mean."""
    train, val, test = load_all_splits()
    global_mean = train["rating"].mean()
    centred = []
    for df in (train, val, test):
        df = df.copy()
        df["rating"] = df["rating"] - global_mean
        centred.append(df)
    return centred[0], centred[1], centred[2], global_mean


def preprocess_data():
    raise NotImplementedError(
        "preprocess_data builds data/processed from the raw Kaggle dump (reference "
        "src/data/load_data.py:9-90); it is a one-off ETL outside this engine's scope -- run the "
        "reference's version and point DATA_DIR at its output.")
