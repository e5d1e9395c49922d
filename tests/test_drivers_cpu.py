"""CPU checks of the driver-side pieces that do not need the GPU: the
hyper-parameter file reader, CSV writers, loaders, the recipe-id hook, and the
PyTorch HPF model against the reference's loss/gradients."""
import os

import numpy as np
import pandas as pd
import pytest


def test_load_best_hyperparams_format(tmp_path):
    from src.experiments.compare_models import load_best_hyperparams
    p = tmp_path / "best_hyperparams.txt"
    p.write_text("BEST CONFIGURATIONS\n===================\n"
                 "GaussianMF: {'n_factors': 30, 'sigma2': 0.3, 'tol': 0.001, 'verbose': True}\n"
                 "PoissonMF: {'n_factors': 40, 'a0': 0.1, 'tol': None}\n\n"
                 "Broken: {not a dict\n")
    got = load_best_hyperparams(str(p))
    assert got == {"GaussianMF": {"n_factors": 30, "sigma2": 0.3, "tol": 0.001, "verbose": True},
                   "PoissonMF": {"n_factors": 40, "a0": 0.1, "tol": None}}
    assert load_best_hyperparams(str(tmp_path / "missing.txt")) == {}


def test_embedding_and_prediction_files_layout(tmp_path, monkeypatch):
    from src.experiments import _full_training as ft
    from src.models.poisson_mf_cavi import PoissonMFCAVIConfig
    monkeypatch.chdir(tmp_path)
    cfg = PoissonMFCAVIConfig(n_factors=3)
    U, I = np.arange(12.0).reshape(4, 3) / 7, np.arange(6.0).reshape(2, 3) / 3
    ft.write_embeddings("poisson_mf", U, I, cfg)
    ue = pd.read_csv("data/embeddings/poisson_mf/user_embeddings.csv", float_precision="round_trip")
    assert list(ue.columns) == ["0", "1", "2"] and ue.shape == (4, 3)
    np.testing.assert_allclose(ue.to_numpy(), U, rtol=0, atol=0)          # full float64 repr round-trips
    ie = pd.read_csv("data/embeddings/poisson_mf/item_embeddings.csv")
    assert list(ie.columns) == ["0", "1", "2"]                             # no mapping files -> no recipe_id column
    assert open("data/embeddings/poisson_mf/config.txt").read() == (
        "{'n_factors': 3, 'a0': 0.3, 'b0': 1.0, 'max_iter': 100, 'tol': 0.0001, 'random_state': 42, 'verbose': True}")
    test_df = pd.DataFrame({"u": [0, 1], "i": [1, 0], "rating": [5.0, 3.0]})
    ft.write_test_predictions("poisson_mf", test_df, np.array([4.5, 2.0]))
    tp = pd.read_csv("data/predictions/poisson_mf/test_predictions.csv")
    assert list(tp.columns) == ["u", "i", "y_true", "y_pred"] and tp["y_pred"].tolist() == [4.5, 2.0]


def test_recipe_id_hook(tmp_path):
    from src.utils.mapping import get_recipe_id_map
    assert get_recipe_id_map(str(tmp_path)) is None
    (tmp_path / "processed").mkdir(); (tmp_path / "raw").mkdir()
    pd.DataFrame({"recipe_id": [10, 11], "i": [0, 1]}).to_csv(tmp_path / "processed" / "dict_i.csv", index=False)
    pd.DataFrame({"id": [100, 101], "i": [7, 9]}).to_csv(tmp_path / "raw" / "PP_recipes.csv", index=False)
    assert get_recipe_id_map(str(tmp_path)) is None                      # needs i_new / i, as the reference does
    pd.DataFrame({"i_new": [1, 0], "i": [7, 9]}).to_csv(tmp_path / "processed" / "dict_i.csv", index=False)
    assert get_recipe_id_map(str(tmp_path)).tolist() == [101, 100]


def test_loaders(tmp_path, monkeypatch):
    from src.data import load_data
    d = tmp_path / "data" / "processed"
    d.mkdir(parents=True)
    for split, vals in (("train", [5.0, 3.0, 4.0]), ("validation", [1.0]), ("test", [2.0])):
        pd.DataFrame({"user_id": 0, "u": range(len(vals)), "i": range(len(vals)), "rating": vals,
                      "split": split}).to_csv(d / f"interactions_{split}.csv", index=False)
    monkeypatch.setattr(load_data, "DATA_DIR", str(d))
    tr, va, te = load_data.load_all_splits()
    assert list(tr.columns) == ["u", "i", "rating"] and len(tr) == 3 and len(va) == 1
    trc, vac, tec, gm = load_data.load_all_splits_centered()
    assert gm == pytest.approx(4.0) and vac["rating"].tolist() == [-3.0] and trc["rating"].sum() == pytest.approx(0)
    with pytest.raises(FileNotFoundError):
        load_data.load_interactions("nope")


def test_hpf_pytorch_matches_reference_loss_and_gradients(golden_dir):
    import json
    import torch
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    d = np.load(os.path.join(golden_dir, "hpf_torch.npz"))
    cfg = HPF_PyTorch_Config(verbose=False, **json.loads(str(d["cfg"])))
    torch.manual_seed(0)
    m = HPF_PyTorch(int(d["n_users"]), int(d["n_items"]), d["user_counts"], d["item_counts"], cfg)
    for name in ("theta_uncons", "beta_uncons", "xi_uncons", "eta_uncons"):       # same draw order
        assert np.array_equal(getattr(m, name).detach().numpy(), d[name]), name
    loss = m.loss(torch.from_numpy(d["batch_u"]), torch.from_numpy(d["batch_i"]), torch.from_numpy(d["batch_r"]))
    assert loss.item() == pytest.approx(float(d["loss"]), rel=1e-6)
    loss.backward()
    for name, key in (("theta_uncons", "grad_theta"), ("beta_uncons", "grad_beta"), ("xi_uncons", "grad_xi"),
                      ("eta_uncons", "grad_eta")):
        np.testing.assert_allclose(getattr(m, name).grad.numpy(), d[key], rtol=2e-5, atol=1e-6, err_msg=key)
    from helpers import GOLDEN  # noqa: F401
    pu = np.array([0, 1, 2, 299]) % int(d["n_users"]); pi = np.array([0, 79, 80, 3]) % int(d["n_items"])
    np.testing.assert_allclose(m.predict(pu, pi), d["predict"], rtol=1e-6)


def test_tune_hpf_pytorch_grid_protocol(capsys):
    """The grid search of tune_hpf_pytorch.py: every combination of the grid, the combination's score = its
    best per-epoch validation RMSE on the un-shifted scale, the reference's printed lines, input frames untouched."""
    import torch
    from src.experiments import tune_hpf_pytorch as tp
    assert tp.PARAM_GRID == {"n_factors": [20, 50], "lr": [0.001, 0.005], "a": [0.3, 1.0], "a_prime": [1.0, 3.0]}
    assert (tp.EPOCHS, tp.BATCH_SIZE) == (10, 4096)
    rng = np.random.default_rng(3)

    def frame(n):
        return pd.DataFrame({"u": rng.integers(0, 60, n), "i": rng.integers(0, 40, n),
                             "rating": rng.integers(0, 6, n).astype(float)})
    train, val, test = frame(1500), frame(200), frame(100)
    test.loc[0, "u"] = 77                      # the dimensions come from all three splits
    before = train["rating"].copy()
    torch.manual_seed(0)
    best, best_rmse, results = tp.run_tuning(splits=(train, val, test), epochs=2, batch_size=256,
                                             param_grid={"n_factors": [4, 6], "lr": [0.01], "a": [0.3], "a_prime": [1.0, 3.0]})
    assert train["rating"].equals(before)
    assert len(results) == 4 and all(np.isfinite(s) for _, s in results)
    assert best_rmse == min(s for _, s in results) and best in [p for p, _ in results]
    out = capsys.readouterr().out
    assert "Total combinations to test: 4" in out and "--- Run 4/4: " in out
    assert out.count("Result RMSE: ") == 4 and f"Best Validation RMSE: {best_rmse:.4f}" in out
    assert "*** New Best RMSE: " in out and f"Best Configuration: {best}" in out
