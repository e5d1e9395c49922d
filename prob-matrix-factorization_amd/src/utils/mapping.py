"""Optional model-item-index -> Food.com recipe id column for
`item_embeddings.csv` (reference: src/utils/mapping.py:5-69).

Same contract as the reference: `data/processed/dict_i.csv` must hold columns
`i_new` (model index) and `i` (raw index), `data/raw/PP_recipes.csv` columns
`id` and `i`; anything missing -> None and the drivers skip the column (with the
reference's current preprocessing, which writes `recipe_id, i`, that is what
happens)."""
import os

import numpy as np
import pandas as pd


def get_recipe_id_map(data_dir="data"):
    index_path = os.path.join(data_dir, "processed", "dict_i.csv")
    recipes_path = os.path.join(data_dir, "raw", "PP_recipes.csv")
    for path in (index_path, recipes_path):
        if not os.path.exists(path):
            print(f"Error: {path} not found.")
            return None
    print("Loading mapping files...")
    index = pd.read_csv(index_path)
    if not {"i_new", "i"} <= set(index.columns):
        print("Error: dict_i.csv must contain 'i_new' and 'i' columns")
        return None
    joined = index.merge(pd.read_csv(recipes_path, usecols=["id", "i"]), on="i", how="left")
    missing = int(joined["id"].isnull().sum())
    if missing:
        print(f"Warning: {missing} items have no matching recipe_id in PP_recipes")
    joined["id"] = joined["id"].fillna(-1)
    joined = joined[joined["i_new"] >= 0]
    id_map = np.zeros(int(joined["i_new"].max()) + 1, dtype=int)
    id_map[joined["i_new"].to_numpy()] = joined["id"].astype(int).to_numpy()
    print(f"Mapping loaded. {len(joined)} items mapped.")
    return id_map
