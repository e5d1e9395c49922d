"""End-to-end driver runs on the synthetic stand-in for the recipe data
(BASELINE config #5 shape: 11,780 users x 13,000 items, ~295k train+val
ratings; the real CSVs cannot be fetched offline)."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

HYPER = """BEST CONFIGURATIONS
===================
GaussianMF: {'n_factors': 30, 'sigma2': 0.3, 'eta_theta2': 0.5, 'eta_beta2': 0.5, 'eta_bias2': 1.0, 'max_iter': 6, 'tol': 0.001, 'random_state': 42, 'verbose': False}
PoissonMF: {'n_factors': 40, 'a0': 0.1, 'b0': 0.5, 'max_iter': 10, 'tol': None, 'random_state': 42, 'verbose': False}
HPF_CAVI: {'n_factors': 20, 'a': 0.3, 'a_prime': 5.0, 'b_prime': 5.0, 'c': 0.3, 'c_prime': 5.0, 'd_prime': 5.0, 'max_iter': 10, 'tol': None, 'random_state': 42, 'verbose': False}
HPF_PyTorch: {'n_factors': 10, 'a': 1.0, 'a_prime': 1.0, 'b_prime': 1.0, 'c': 1.0, 'c_prime': 1.0, 'd_prime': 1.0, 'lr': 0.0005, 'batch_size': 1024, 'epochs': 2, 'device': 'cpu', 'verbose': False}
"""


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    from helpers import recipe_standin
    root = tmp_path_factory.mktemp("recipes")
    d = root / "data" / "processed"
    d.mkdir(parents=True)
    for name, frame in zip(("train", "validation", "test"), recipe_standin()):
        frame.to_csv(d / f"interactions_{name}.csv", index=False)
    (root / "best_hyperparams.txt").write_text(HYPER)
    return root


def test_train_all_models_writes_reference_file_layout(workdir, monkeypatch, capsys):
    from src.experiments import train_all_models
    monkeypatch.chdir(workdir)
    monkeypatch.setattr("sys.argv", ["train_all_models", "--dataset_mode", "train+val"])
    train_all_models.main()
    out = capsys.readouterr().out
    assert "Failed" not in out, out[-3000:]
    K = {"gaussian_mf": 30, "poisson_mf": 40, "hpf_cavi": 20, "hpf_pytorch": 10}
    test = pd.read_csv("data/processed/interactions_test.csv")
    for name, k in K.items():
        ue = pd.read_csv(f"data/embeddings/{name}/user_embeddings.csv")
        ie = pd.read_csv(f"data/embeddings/{name}/item_embeddings.csv")
        assert list(ue.columns) == [str(c) for c in range(k)] and list(ie.columns) == list(ue.columns)
        assert len(ue) == 11_780 and len(ie) == 13_000 and np.isfinite(ue.to_numpy()).all()
        cfg = open(f"data/embeddings/{name}/config.txt").read()
        assert cfg.startswith("{'n_factors': %d" % k)
        assert ("global_mean:" in cfg) == (name == "gaussian_mf")
        tp = pd.read_csv(f"data/predictions/{name}/test_predictions.csv")
        assert list(tp.columns) == ["u", "i", "y_true", "y_pred"] and len(tp) == len(test)
        assert np.isfinite(tp["y_pred"]).all()
    # sanity: every model's test RMSE is of the order of the rating spread (std ~1.8 on this data)
    rm = {n: np.sqrt(np.mean((pd.read_csv(f"data/predictions/{n}/test_predictions.csv").eval("y_true - y_pred")) ** 2))
          for n in K}
    assert all(0.5 < v < 2.5 for v in rm.values()), rm


def test_compare_models_runs_all_four(workdir, monkeypatch, capsys):
    from src.experiments import compare_models
    monkeypatch.chdir(workdir)
    compare_models.main()
    out = capsys.readouterr().out
    assert "failed" not in out, out[-3000:]
    res = pd.read_csv("model_comparison_results.csv")
    assert res["Model"].tolist() == ["Gaussian MF (CAVI)", "Poisson MF (CAVI)", "HPF (CAVI)", "HPF (PyTorch)"]
    assert np.isfinite(res.drop(columns="Model").to_numpy()).all()
    params = open("model_comparison_params.txt").read()
    assert params.startswith("=== Gaussian MF (CAVI) ===\n{'n_factors': 30")


def test_config5_poisson_k64_rmse_parity_with_the_reference(workdir, golden_dir):
    """BASELINE config #5: Poisson MF K=64 on the recipe-shaped data (train+val), the 150 iterations of
    SURVEY.md section 8(d) (best_hyperparams.txt:4), against what THE REFERENCE computed on the same frames
    (`config5_poisson.npz`, tests/golden/make_golden.py config5; the CPU oracle is held to the same file in
    tests/test_oracle_golden.py): |dRMSE| <= 1e-4 on the test set (fp32 device arithmetic), factor rows within
    2e-3, and identical top-10 items for >= 99% of the sampled users (ties within 1e-5 of the k-th score are
    not counted as differences)."""
    from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
    import json
    ref = np.load(os.path.join(golden_dir, "config5_poisson.npz"))
    d = workdir / "data" / "processed"
    tr = pd.concat([pd.read_csv(d / "interactions_train.csv"), pd.read_csv(d / "interactions_validation.csv")])
    te = pd.read_csv(d / "interactions_test.csv")
    assert (len(tr), len(te)) == (int(ref["n_train_rows"]), int(ref["n_test_rows"]))       # the same frames
    cfg = json.loads(str(ref["cfg"]))
    assert cfg["max_iter"] == 150 and cfg["n_factors"] == 64
    m = PoissonMFCAVI(PoissonMFCAVIConfig(verbose=False, **cfg), dtype="f32").fit(tr)
    assert abs(float(ref["test_rmse"]) - m.evaluate_rmse(te)) <= 1e-4
    np.testing.assert_allclose(m.predict(te["u"].to_numpy(), te["i"].to_numpy()), ref["test_pred"], rtol=2e-3, atol=1e-4)
    users = ref["users"]
    assert np.max(np.abs(m.E_theta[users] - ref["E_theta_rows"]) / (np.abs(ref["E_theta_rows"]) + 1e-9)) <= 2e-3
    assert abs(m.E_theta.sum() / float(ref["E_theta_sum"]) - 1) <= 1e-4 and abs(m.E_beta.sum() / float(ref["E_beta_sum"]) - 1) <= 1e-4
    items, _ = m.top_k_items(users, 10)
    same = 0
    for row, (tb, sb) in enumerate(zip(ref["top11"], ref["top11_scores"])):
        kth = sb[9]
        tied = {int(j) for j, v in zip(tb, sb) if abs(v - kth) <= 1e-5 * abs(kth)}        # near the 10th score (incl. the 11th)
        ta = {int(j) for j in items[row]}
        same += ta == set(map(int, tb[:10])) or (ta - set(map(int, tb[:10]))) <= tied
    assert same >= 297


def test_config5_poisson_k64_f64_identical_rankings_with_the_reference(workdir, golden_dir):
    """The same run in the engine's float64 mode -- the reference's own arithmetic (poisson_mf_cavi.py:135-197 is
    float64 NumPy) -- where "identical top-k item rankings" is asserted literally: after the 150 iterations the
    top-10 list of every one of the 300 sampled users equals the reference's, in order, with NO tie tolerance;
    test predictions and factor rows to rtol 1e-8 (the two differ by summation order only)."""
    from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
    import json
    ref = np.load(os.path.join(golden_dir, "config5_poisson.npz"))
    d = workdir / "data" / "processed"
    tr = pd.concat([pd.read_csv(d / "interactions_train.csv"), pd.read_csv(d / "interactions_validation.csv")])
    te = pd.read_csv(d / "interactions_test.csv")
    assert (len(tr), len(te)) == (int(ref["n_train_rows"]), int(ref["n_test_rows"]))
    cfg = json.loads(str(ref["cfg"]))
    m = PoissonMFCAVI(PoissonMFCAVIConfig(verbose=False, **cfg), dtype="f64").fit(tr)
    assert m.history_["iterations"] == 150
    np.testing.assert_allclose(m.predict(te["u"].to_numpy(), te["i"].to_numpy()), ref["test_pred"], rtol=1e-8, atol=0)
    assert abs(float(ref["test_rmse"]) - m.evaluate_rmse(te)) <= 1e-10
    users = ref["users"]
    np.testing.assert_allclose(m.E_theta[users], ref["E_theta_rows"], rtol=1e-8, atol=0)
    assert abs(m.E_theta.sum() / float(ref["E_theta_sum"]) - 1) <= 1e-10 and abs(m.E_beta.sum() / float(ref["E_beta_sum"]) - 1) <= 1e-10
    items, scores = m.top_k_items(users, 10)
    assert len(users) == 300
    np.testing.assert_array_equal(items, ref["top11"][:, :10])          # 300 / 300, in order, no tolerance
    np.testing.assert_allclose(scores, ref["top11_scores"][:, :10], rtol=1e-8, atol=0)


def test_tune_all_models_concurrent_trials_write_a_loadable_file(workdir, monkeypatch, capsys):
    """Random search with 4 concurrent engine contexts; the file it writes must
    round-trip through load_best_hyperparams into the config dataclasses."""
    import shutil
    from src.experiments import tune_all_models
    from src.experiments.compare_models import load_best_hyperparams
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVIConfig
    from src.models.hpf_cavi import HPF_CAVI_Config
    from src.models.poisson_mf_cavi import PoissonMFCAVIConfig
    monkeypatch.chdir(workdir)
    shutil.copy("best_hyperparams.txt", "best_hyperparams.keep")
    try:
        monkeypatch.setattr("sys.argv", ["tune_all_models", "--n_trials", "4", "--workers", "4", "--seed", "3"])
        monkeypatch.setattr(tune_all_models, "tune_hpf_pytorch", lambda *a, **k: None)   # covered elsewhere; slow
        tune_all_models.main()
        out = capsys.readouterr().out
        assert "failed" not in out and out.count("Trial ") == 12, out[-2000:]
        got = load_best_hyperparams("best_hyperparams.txt")
        assert set(got) == {"GaussianMF", "PoissonMF", "HPF_CAVI"}
        GaussianMFCAVIConfig(**got["GaussianMF"]); PoissonMFCAVIConfig(**got["PoissonMF"]); HPF_CAVI_Config(**got["HPF_CAVI"])
        assert got["GaussianMF"]["n_factors"] in (30, 50, 70) and got["HPF_CAVI"]["a"] == got["HPF_CAVI"]["c"]
        # the same seed draws the same configurations regardless of the worker count
        first = open("best_hyperparams.txt").read()
        monkeypatch.setattr("sys.argv", ["tune_all_models", "--n_trials", "4", "--workers", "1", "--seed", "3"])
        tune_all_models.main()
        assert open("best_hyperparams.txt").read() == first
    finally:
        shutil.move("best_hyperparams.keep", "best_hyperparams.txt")


def test_engine_and_torch_share_a_process_when_torch_is_imported_first():
    """Load order contract: the CAVI engine never imports torch; torch bundles its own HIP runtime and has to
    be mapped before libpmf_hip.so pulls in /opt/rocm's.  torch first (what the mixed drivers and
    hpf_pytorch do at import time): both see the GPU.  Engine first: the engine keeps working, the late
    torch import is flagged (pmf_hip.loaded_before_torch) and PMF_HIP_TORCH_PRELOAD=1 restores the old order."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prob-matrix-factorization_amd")
    first = ("import sys; sys.path.insert(0, %r); import src.experiments.compare_models, pmf_hip; "
             "c = pmf_hip.Context(4, 4, 4); c.close(); import torch; "
             "assert torch.cuda.is_available() and not pmf_hip.loaded_before_torch(); print('ok')") % pkg
    out = subprocess.run([sys.executable, "-c", first], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    late = ("import sys; sys.path.insert(0, %r); import pmf_hip; c = pmf_hip.Context(4, 4, 4); c.close(); "
            "assert 'torch' not in sys.modules and pmf_hip.loaded_before_torch(); print('ok')") % pkg
    out = subprocess.run([sys.executable, "-c", late], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    pre = ("import sys; sys.path.insert(0, %r); import pmf_hip; c = pmf_hip.Context(4, 4, 4); c.close(); import torch; "
           "assert torch.cuda.is_available() and not pmf_hip.loaded_before_torch(); print('ok')") % pkg
    out = subprocess.run([sys.executable, "-c", pre], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PMF_HIP_TORCH_PRELOAD="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_hpf_pytorch_graph_replay_trains_like_the_eager_loop():
    """The HIP-graph captured Adam step (full batches) + eager tail batch against the plain eager
    loop: same seeds, same shuffles -> same parameters up to the atomic-add order of the
    embedding gradients; and the graph path must really have replayed the full batches."""
    import time
    import torch
    from src.experiments.train_hpf_pytorch_full import adam_epochs
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    U, I, N, K = 3000, 2000, 50_000, 10
    rng = np.random.default_rng(0)
    un, inn = rng.integers(0, U, N), rng.integers(0, I, N)
    dev = torch.device("cuda")
    u, i = torch.from_numpy(un).to(dev), torch.from_numpy(inn).to(dev)
    r = torch.from_numpy(rng.integers(1, 7, N).astype(np.float32)).to(dev)
    uc, ic = np.bincount(un, minlength=U), np.bincount(inn, minlength=I)
    out, secs, info = [], [], []
    for use_graph in (False, True):
        torch.manual_seed(0)
        m = HPF_PyTorch(U, I, uc, ic, HPF_PyTorch_Config(n_factors=K)).to(dev)
        torch.manual_seed(1)
        adam_epochs(m, u, i, r, 5e-3, 4096, 1, verbose=False, graph=use_graph)     # includes capture / warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        adam_epochs(m, u, i, r, 5e-3, 4096, 6, verbose=False, graph=use_graph)     # 12 full + 1 tail batch per epoch
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
        out.append([p.detach().cpu().numpy() for p in m.parameters()])
        info.append(m.training_info_)
    for a, b in zip(*out):
        assert np.allclose(a, b, rtol=2e-3, atol=2e-4)
    print(f"eager {secs[0]:.3f} s, graph {secs[1]:.3f} s")     # informational: 0.085 s against 0.062 s measured
    # the graph path really replayed the 12 full batches of each of the 6 epochs; the eager one none
    assert info[0] == {"graph_replays": 0, "steps": 78} and info[1] == {"graph_replays": 72, "steps": 78}


def test_hpf_pytorch_on_cuda_matches_the_reference_golden(golden_dir):
    """`HPF_PyTorch` placed on the MI355X against the vectors the reference produced on its CPU
    (tests/golden/hpf_torch.npz: hpf_pytorch.py:71-184): loss value, the four gradient tables and
    `predict`, at fp32 tolerance; then ONE Adam step of the graph-captured training loop against the
    same step of the eager loop, both starting from the golden parameters
    (train_hpf_pytorch_full.py:98-108)."""
    import copy
    import json
    import torch
    from src.experiments.train_hpf_pytorch_full import adam_epochs
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    d = np.load(os.path.join(golden_dir, "hpf_torch.npz"))
    cfg = HPF_PyTorch_Config(verbose=False, **json.loads(str(d["cfg"])))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = HPF_PyTorch(int(d["n_users"]), int(d["n_items"]), d["user_counts"], d["item_counts"], cfg)
    for name in ("theta_uncons", "beta_uncons", "xi_uncons", "eta_uncons"):       # drawn on the CPU: reference order
        assert np.array_equal(getattr(m, name).detach().numpy(), d[name]), name
    m = m.to(dev)
    bu, bi, br = (torch.from_numpy(d[k]).to(dev) for k in ("batch_u", "batch_i", "batch_r"))
    loss = m.loss(bu, bi, br)
    assert loss.device.type == "cuda"
    assert loss.item() == pytest.approx(float(d["loss"]), rel=2e-6)
    loss.backward()
    for name, key in (("theta_uncons", "grad_theta"), ("beta_uncons", "grad_beta"), ("xi_uncons", "grad_xi"),
                      ("eta_uncons", "grad_eta")):
        np.testing.assert_allclose(getattr(m, name).grad.cpu().numpy(), d[key], rtol=5e-5, atol=2e-6, err_msg=key)
    pu = np.array([0, 1, 2, 299]) % int(d["n_users"]); pi = np.array([0, 79, 80, 3]) % int(d["n_items"])
    np.testing.assert_allclose(m.predict(pu, pi), d["predict"], rtol=2e-6)
    # one Adam step over exactly one full batch: graph replay vs eager, from the golden parameters
    m.zero_grad(set_to_none=True)
    eager, graphed = copy.deepcopy(m), copy.deepcopy(m)
    n = len(br)
    for model, use_graph in ((eager, False), (graphed, True)):
        torch.manual_seed(5)                       # same shuffle
        adam_epochs(model, bu, bi, br, lr=1e-2, batch_size=n, epochs=1, verbose=False, graph=use_graph)
    assert graphed.training_info_["graph_replays"] == 1 and eager.training_info_["graph_replays"] == 0
    for name in ("theta_uncons", "beta_uncons", "xi_uncons", "eta_uncons"):
        a, b = getattr(eager, name).detach().cpu().numpy(), getattr(graphed, name).detach().cpu().numpy()
        assert np.abs(a - d[name]).max() > 1e-3, name                    # the step moved the parameters
        np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-6, err_msg=name)
        # Adam's first step moves every touched parameter by lr against the gradient's sign
        g = d["grad_" + name.split("_")[0]]
        moved = np.abs(g) > 1e-4
        np.testing.assert_allclose((a - d[name])[moved], -1e-2 * np.sign(g[moved]), rtol=1e-3, atol=1e-6, err_msg=name)


def test_tune_hpf_pytorch_grid_on_cuda(capsys):
    """The PyTorch HPF grid search with the module and the ratings on the GPU: the captured Adam step is replayed
    inside every combination, the per-epoch validation hook sees the updated parameters (the score improves over
    the epochs on learnable data), and the best combination is the minimum of the per-combination scores."""
    import torch
    from pmf_hip.synth import synth_ratings
    from src.experiments import tune_hpf_pytorch as tp
    u, i, r = synth_ratings(2000, 500, 60_000, seed=11)
    frames = []
    for lo, hi in ((0, 50_000), (50_000, 56_000), (56_000, 60_000)):
        frames.append(pd.DataFrame({"u": u[lo:hi].astype(np.int64), "i": i[lo:hi].astype(np.int64), "rating": r[lo:hi]}))
    torch.manual_seed(0)
    best, best_rmse, results = tp.run_tuning(splits=tuple(frames), epochs=3, batch_size=4096,
                                             param_grid={"n_factors": [8], "lr": [0.02, 0.0001], "a": [0.3], "a_prime": [1.0]})
    assert len(results) == 2 and best_rmse == min(s for _, s in results)
    scores = {p["lr"]: s for p, s in results}
    assert scores[0.02] < scores[0.0001]          # three epochs at lr 1e-4 barely move the parameters
    assert best["lr"] == 0.02
    assert "Total combinations to test: 2" in capsys.readouterr().out
