"""Random problems through the model classes' HOST logic against the reference itself, run live.

Only in the build container: `/root/reference` does not exist on the GPU box, and nothing of it travels --
`tests/golden/live_reference.py` runs the reference's classes in a subprocess (its `src` package and this repo's
never share an interpreter) and hands back arrays and the captured stdout.  This side runs the SAME model classes
the GPU uses, over the CPU stand-in of the engine (`tests/oracle_engine.py:OracleContext`), so what is compared is
everything above the kernels: dimension inference, the RNG draw order of the initial state, the iteration order,
the validation monitor, the early-stop rules, `predict` / `evaluate_*` semantics, and the verbose output --
character for character."""
import contextlib
import io
import json
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "models")),
                                reason="the reference exists in the build container only")

KEYS = {"hpf": ["gamma_a_theta", "gamma_b_theta", "gamma_a_beta", "gamma_b_beta", "E_theta", "E_beta", "E_xi", "E_eta"],
        "poisson": ["a_theta", "b_theta", "a_beta", "b_beta", "E_theta", "E_beta"],
        "poisson_ext": ["a_theta", "b_theta", "a_beta", "b_beta", "a_phi", "b_phi", "a_psi", "b_psi", "E_theta", "E_beta",
                        "E_phi", "E_psi"],
        "gauss_bias": ["m_theta", "m_beta", "V_theta", "V_beta", "m_user_bias", "m_item_bias"],
        "gauss": ["m_theta", "m_beta", "V_theta", "V_beta"]}


def _config(kind, rng):
    K = int(rng.choice([1, 2, 5, 8, 13]))
    common = dict(n_factors=K, max_iter=int(rng.integers(1, 7)), random_state=int(rng.integers(0, 1000)), verbose=True)
    if kind == "hpf":
        return dict(common, a=0.3, a_prime=2.0, b_prime=1.5, c=0.4, c_prime=3.0, d_prime=0.7,
                    tol=[None, 1e-3, 0.05][int(rng.integers(0, 3))])
    if kind in ("poisson", "poisson_ext"):
        return dict(common, a0=0.2, b0=0.6, tol=[None, 1e-3, 0.05][int(rng.integers(0, 3))])
    cfg = dict(common, sigma2=0.4, eta_theta2=0.6, eta_beta2=0.9, tol=[1e-3, 0.05, -1.0][int(rng.integers(0, 3))])
    if kind == "gauss_bias":
        cfg["eta_bias2"] = 1.3
    return cfg


def _build(kind, config):
    if kind == "hpf":
        from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
        return HPF_CAVI(HPF_CAVI_Config(**config), dtype="f64")
    if kind == "poisson":
        from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
        return PoissonMFCAVI(PoissonMFCAVIConfig(**config), dtype="f64")
    if kind == "poisson_ext":
        from src.models.poisson_mf_extended_cavi import PoissonMFExtendedCAVI, PoissonMFExtendedCAVIConfig
        return PoissonMFExtendedCAVI(PoissonMFExtendedCAVIConfig(**config), dtype="f64")
    if kind == "gauss_bias":
        from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
        return GaussianMFCAVI(GaussianMFCAVIConfig(**config), dtype="f64")
    from src.models.gaussian_mf_cavi import GaussianMFCAVI, GaussianMFCAVIConfig
    return GaussianMFCAVI(GaussianMFCAVIConfig(**config), dtype="f64")


def test_random_problems_host_logic_equals_the_live_reference(tmp_path, monkeypatch):
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import pmf_hip
    from fuzz_parity import problem
    from oracle_engine import OracleContext
    rng = np.random.default_rng(20251226)
    n_trials = 150
    inputs, metas = {"n_trials": np.asarray(n_trials)}, []
    for t in range(n_trials):
        kind = str(rng.choice(list(KEYS)))
        shape, u, i, x, (vu, vi, vx) = problem(rng)
        gauss = kind.startswith("gauss")
        gm = float(x.mean()) if gauss else 0.0
        shift = -gm if gauss else (1.0 if kind in ("hpf", "poisson_ext") else 0.0)      # the drivers' preprocessing
        U, I = int(u.max()) + 1, int(i.max()) + 1
        meta = {"kind": kind, "config": _config(kind, rng), "global_mean": gm, "validate": bool(rng.random() < 0.8)}
        metas.append((meta, shape))
        inputs.update({f"t{t}_u": u, f"t{t}_i": i, f"t{t}_x": x + shift, f"t{t}_vu": vu, f"t{t}_vi": vi, f"t{t}_vx": vx + shift,
                       f"t{t}_qu": rng.integers(0, U + 2, 20), f"t{t}_qi": rng.integers(0, I + 2, 20),
                       f"t{t}_cfg": np.asarray(json.dumps(meta))})
    src, dst = str(tmp_path / "problems.npz"), str(tmp_path / "reference_out.npz")
    np.savez(src, **inputs)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference.py"), src, dst], env=env,
                          capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert done.returncode == 0, done.stderr[-2000:]
    ref = np.load(dst, allow_pickle=False)

    monkeypatch.setattr(pmf_hip, "Context", OracleContext)      # the model classes look the context class up at run time
    seen = {"validated": 0, "stopped_early": 0, "warned": 0, "kinds": set()}
    for t, (meta, shape) in enumerate(metas):
        kind, gm = meta["kind"], meta["global_mean"]
        gauss = kind.startswith("gauss")
        tag = f"trial {t}: {kind} {shape} {meta['config']} validate={meta['validate']}"
        train = pd.DataFrame({"u": inputs[f"t{t}_u"], "i": inputs[f"t{t}_i"], "rating": inputs[f"t{t}_x"]})
        vdf = pd.DataFrame({"u": inputs[f"t{t}_vu"], "i": inputs[f"t{t}_vi"], "rating": inputs[f"t{t}_vx"]})
        val = vdf if meta["validate"] else None
        model = _build(kind, meta["config"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            if gauss:
                model.fit(train, val_df=val, global_mean=gm)
            else:
                model.fit(train, val_df=val)
        assert buf.getvalue() == str(ref[f"t{t}_stdout"]), tag
        seen["kinds"].add(kind)
        seen["validated"] += "Validation RMSE" in buf.getvalue()
        seen["stopped_early"] += model.history_["stopped_early"]
        seen["warned"] += "Warning" in buf.getvalue()
        for k in KEYS[kind]:
            np.testing.assert_allclose(np.asarray(getattr(model, k)), ref[f"t{t}_{k}"], rtol=1e-9, atol=1e-12, err_msg=f"{tag}: {k}")
        qu, qi = inputs[f"t{t}_qu"], inputs[f"t{t}_qi"]
        got = model.predict(qu, qi, gm) if gauss else model.predict(qu, qi)
        np.testing.assert_allclose(got, ref[f"t{t}_predict"], rtol=1e-9, atol=1e-12, err_msg=tag)
        ebuf = io.StringIO()
        want_ev = ref[f"t{t}_evaluate"]
        with contextlib.redirect_stdout(ebuf), np.errstate(all="ignore"):
            ev = [model.evaluate_rmse(vdf, gm) if gauss else model.evaluate_rmse(vdf)]
            if len(want_ev) == 2:      # (the reference's bias-free Gaussian class has no MacroMAE method)
                ev.append(model.evaluate_macro_mae(vdf, gm) if gauss else model.evaluate_macro_mae(vdf))
        np.testing.assert_allclose(np.asarray(ev, dtype=np.float64), want_ev, rtol=1e-9, equal_nan=True, err_msg=tag)
        assert ebuf.getvalue() == str(ref[f"t{t}_eval_stdout"]), tag
    # the sweep really went through the interesting branches
    assert seen["kinds"] == set(KEYS) and seen["validated"] >= 80 and seen["stopped_early"] >= 10, seen


def test_random_hpf_pytorch_cases_equal_the_live_reference(tmp_path):
    """The PyTorch HPF model (row a13) against the reference's, run live: identical initial parameters under the
    same torch seed, loss value and all four gradient tables on a random batch, then a few Adam steps and
    `predict` (ids outside the tables included)."""
    import torch
    from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
    rng = np.random.default_rng(7)
    n_trials = 25
    inputs, metas = {"n_trials": np.asarray(n_trials)}, []
    for t in range(n_trials):
        U, I, K = int(rng.integers(1, 60)), int(rng.integers(1, 40)), int(rng.choice([1, 3, 8, 20]))
        n = int(rng.integers(1, 300))
        u, i = rng.integers(0, U, n), rng.integers(0, I, n)
        config = dict(n_factors=K, a=float(rng.choice([0.3, 1.0])), a_prime=float(rng.choice([1.0, 3.0])), b_prime=1.0,
                      c=float(rng.choice([0.3, 1.0])), c_prime=1.0, d_prime=float(rng.choice([0.7, 1.0])), lr=0.01, verbose=False)
        meta = {"kind": "hpf_torch", "n_users": U, "n_items": I, "config": config, "seed": int(rng.integers(0, 1000)),
                "steps": int(rng.integers(1, 5))}
        metas.append(meta)
        inputs.update({f"t{t}_user_counts": np.bincount(u, minlength=U).astype(np.float64),
                       f"t{t}_item_counts": np.bincount(i, minlength=I).astype(np.float64),
                       f"t{t}_bu": u, f"t{t}_bi": i, f"t{t}_br": (rng.integers(0, 6, n) + 1).astype(np.float32),
                       f"t{t}_qu": rng.integers(0, U, 15), f"t{t}_qi": rng.integers(0, I, 15),
                       f"t{t}_cfg": np.asarray(json.dumps(meta))})
    src, dst = str(tmp_path / "torch_problems.npz"), str(tmp_path / "torch_reference_out.npz")
    np.savez(src, **inputs)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference.py"), src, dst], env=env,
                          capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert done.returncode == 0, done.stderr[-2000:]
    ref = np.load(dst, allow_pickle=False)
    names = ("theta_uncons", "beta_uncons", "xi_uncons", "eta_uncons")
    for t, meta in enumerate(metas):
        tag = f"trial {t}: {meta}"
        torch.manual_seed(meta["seed"])
        m = HPF_PyTorch(meta["n_users"], meta["n_items"], inputs[f"t{t}_user_counts"], inputs[f"t{t}_item_counts"],
                        HPF_PyTorch_Config(**meta["config"]))
        for name in names:          # the same draws, in the same order
            assert np.array_equal(getattr(m, name).detach().numpy(), ref[f"t{t}_{name}"]), f"{tag}: {name}"
        bu, bi = torch.from_numpy(inputs[f"t{t}_bu"]), torch.from_numpy(inputs[f"t{t}_bi"])
        br = torch.from_numpy(inputs[f"t{t}_br"])
        loss = m.loss(bu, bi, br)
        loss.backward()
        assert loss.item() == pytest.approx(float(ref[f"t{t}_loss"]), rel=2e-6, abs=1e-5), tag
        for name in names:
            np.testing.assert_allclose(getattr(m, name).grad.numpy(), ref[f"t{t}_grad_{name}"], rtol=1e-4, atol=2e-5,
                                       err_msg=f"{tag}: grad {name}")
        opt = torch.optim.Adam(m.parameters(), lr=meta["config"]["lr"])
        for _ in range(meta["steps"]):
            opt.zero_grad()
            m.loss(bu, bi, br).backward()
            opt.step()
        m.eval()
        np.testing.assert_allclose(m.predict(inputs[f"t{t}_qu"], inputs[f"t{t}_qi"]), ref[f"t{t}_predict"], rtol=1e-4,
                                   atol=1e-5, err_msg=tag)
        np.testing.assert_allclose(m.theta.detach().numpy(), ref[f"t{t}_theta_after"], rtol=1e-4, atol=1e-6, err_msg=tag)


def test_random_metric_inputs_equal_the_live_reference(tmp_path):
    """`src.evaluation.metrics` (rmse, mae, macro_mae and the two log predictive likelihoods) on random inputs."""
    from src.evaluation import metrics as M
    rng = np.random.default_rng(3)
    n_trials = 30
    inputs = {"n_trials": np.asarray(n_trials)}
    for t in range(n_trials):
        n, U, I, K = int(rng.integers(1, 200)), int(rng.integers(1, 20)), int(rng.integers(1, 15)), int(rng.integers(1, 6))
        y = rng.integers(0, 6, n).astype(np.float64) if t % 2 else rng.normal(size=n).round(1)
        meta = {"kind": "metrics", "sigma": float(rng.uniform(0.3, 2.0))}
        inputs.update({f"t{t}_y": y, f"t{t}_p": y + rng.normal(size=n), f"t{t}_u": rng.integers(0, U, n),
                       f"t{t}_i": rng.integers(0, I, n), f"t{t}_x": rng.integers(0, 6, n).astype(np.float64),
                       f"t{t}_theta": rng.normal(size=(U, K)), f"t{t}_beta": rng.normal(size=(I, K)),
                       f"t{t}_cfg": np.asarray(json.dumps(meta)), f"t{t}_sigma": np.asarray(meta["sigma"])})
    src, dst = str(tmp_path / "metric_inputs.npz"), str(tmp_path / "metric_reference_out.npz")
    np.savez(src, **inputs)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference.py"), src, dst], env=env,
                          capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert done.returncode == 0, done.stderr[-2000:]
    ref = np.load(dst, allow_pickle=False)
    for t in range(n_trials):
        y, p = inputs[f"t{t}_y"], inputs[f"t{t}_p"]
        frame = pd.DataFrame({"u": inputs[f"t{t}_u"], "i": inputs[f"t{t}_i"], "rating": inputs[f"t{t}_x"]})
        th, be = inputs[f"t{t}_theta"], inputs[f"t{t}_beta"]
        got = [M.rmse(y, p), M.mae(y, p), M.macro_mae(y, p),
               M.GaussianLogPredictiveLikelihood(frame, th, be, float(inputs[f"t{t}_sigma"])),
               M.PoissonLogPredictiveLikelihood(frame, np.abs(th), np.abs(be))]
        np.testing.assert_allclose(np.asarray(got, dtype=np.float64), ref[f"t{t}_values"], rtol=1e-12, err_msg=f"trial {t}")


DRIVER_HYPER = """BEST CONFIGURATIONS
===================
GaussianMF: {'n_factors': 6, 'sigma2': 0.3, 'eta_theta2': 0.5, 'eta_beta2': 0.5, 'eta_bias2': 1.0, 'max_iter': 4, 'tol': 0.001, 'random_state': 42, 'verbose': False}
PoissonMF: {'n_factors': 5, 'a0': 0.1, 'b0': 0.5, 'max_iter': 5, 'tol': None, 'random_state': 42, 'verbose': False}
HPF_CAVI: {'n_factors': 4, 'a': 0.3, 'a_prime': 5.0, 'b_prime': 5.0, 'c': 0.3, 'c_prime': 5.0, 'd_prime': 5.0, 'max_iter': 5, 'tol': None, 'random_state': 42, 'verbose': False}
HPF_PyTorch: {'n_factors': 3, 'a': 1.0, 'a_prime': 1.0, 'b_prime': 1.0, 'c': 1.0, 'c_prime': 1.0, 'd_prime': 1.0, 'lr': 0.0005, 'batch_size': 64, 'epochs': 1, 'device': 'cpu', 'verbose': False}
"""


def _tree(root):
    return sorted(os.path.relpath(os.path.join(d, f), root) for d, _, fs in os.walk(os.path.join(root, "data")) for f in fs
                  if "processed" not in d)


@pytest.mark.parametrize("mode", ["train", "train+val"])
def test_driver_outputs_on_disk_equal_the_live_reference(mode, tmp_path, monkeypatch):
    """The on-disk surface: the four full-training drivers of the reference and of this repo
    on the same tiny processed data and the same best_hyperparams.txt, each in a directory of its own -- same file
    tree, same `config.txt` text, same CSV headers / shapes / index columns, same values (CAVI models: rtol 1e-9;
    the PyTorch model's shuffled training is not pinned by the reference, so only its files' form is compared)."""
    import pmf_hip
    from oracle_engine import OracleContext
    rng = np.random.default_rng(11)
    n = 900
    u, i = rng.integers(0, 40, n), rng.integers(0, 25, n)
    u[0], i[0] = 39, 24
    r = rng.integers(0, 6, n).astype(float)
    part = rng.choice(3, size=n, p=[0.8, 0.1, 0.1])
    part[0] = 0
    roots = {}
    for who in ("reference", "ours"):
        root = tmp_path / who
        d = root / "data" / "processed"
        d.mkdir(parents=True)
        for k, name in enumerate(("train", "validation", "test")):
            pd.DataFrame({"u": u[part == k], "i": i[part == k], "rating": r[part == k]}).to_csv(d / f"interactions_{name}.csv", index=False)
        (root / "best_hyperparams.txt").write_text(DRIVER_HYPER)
        if mode == "train+val":
            # with the optional mapping files present (src/utils/mapping.py) item_embeddings.csv gets a leading
            # recipe_id column; one model index has no recipe (-> -1 and a warning)
            (root / "data" / "raw").mkdir()
            raw_ids = np.random.default_rng(1).permutation(200)[:25]
            pd.DataFrame({"i_new": np.arange(25), "i": raw_ids}).to_csv(d / "dict_i.csv", index=False)
            pd.DataFrame({"id": 5000 + raw_ids[:-1], "i": raw_ids[:-1], "other": 0}).to_csv(root / "data" / "raw" / "PP_recipes.csv", index=False)
        roots[who] = str(root)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference_drivers.py"), mode], env=env,
                          capture_output=True, text=True, timeout=900, cwd=roots["reference"])
    assert done.returncode == 0, done.stderr[-2000:]

    monkeypatch.setattr(pmf_hip, "Context", OracleContext)
    monkeypatch.chdir(roots["ours"])
    from src.experiments.train_gaussian_full import train_full_gaussian
    from src.experiments.train_hpf_cavi_full import train_full_hpf_cavi
    from src.experiments.train_hpf_pytorch_full import train_full_hpf_pytorch
    from src.experiments.train_poisson_full import train_full_poisson
    ours_out = {}
    for name, fn in (("gaussian_mf", train_full_gaussian), ("poisson_mf", train_full_poisson), ("hpf_cavi", train_full_hpf_cavi),
                     ("hpf_pytorch", train_full_hpf_pytorch)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            fn(dataset_mode=mode)
        ours_out[name] = buf.getvalue()

    assert _tree(roots["ours"]) == _tree(roots["reference"])
    for rel in _tree(roots["reference"]):
        a, b = os.path.join(roots["ours"], rel), os.path.join(roots["reference"], rel)
        if rel.endswith("config.txt"):
            assert open(a).read() == open(b).read(), rel
            continue
        fa, fb = pd.read_csv(a), pd.read_csv(b)
        assert list(fa.columns) == list(fb.columns) and fa.shape == fb.shape, rel
        if rel.endswith("item_embeddings.csv"):
            assert ("recipe_id" in fa.columns) == (mode == "train+val"), rel
        assert open(a).readline() == open(b).readline(), rel                     # the header line, byte for byte
        if "hpf_pytorch" in rel:
            keys = [c for c in fa.columns if c in ("u", "i", "y_true")]
            assert fa[keys].equals(fb[keys]), rel
            continue
        np.testing.assert_allclose(fa.to_numpy(dtype=float), fb.to_numpy(dtype=float), rtol=1e-9, atol=1e-12, err_msg=rel)
    # what the drivers print (timings differ; the PyTorch loop's loss values are not pinned)
    import difflib
    for name in ("gaussian_mf", "poisson_mf", "hpf_cavi", "hpf_pytorch"):
        mask = "=== Training Full HPF (PyTorch)" if name == "hpf_pytorch" else None
        theirs = _comparable(open(os.path.join(roots["reference"], f"stdout_{name}.txt")).read(), mask)
        mine = _comparable(ours_out[name], mask)
        delta = "\n".join(difflib.unified_diff(theirs, mine, "reference", "ours", lineterm="", n=0))
        assert mine == theirs, f"{name}\n{delta}"


def _comparable(text, mask_from=None):
    """Driver output without what cannot be equal: wall times are dropped, and from the line holding `mask_from` on
    (the PyTorch model's unseeded training) measured numbers are masked, so that the FORM of those lines still counts."""
    import re
    lines, masking = [], False
    for ln in text.splitlines():
        masking = masking or bool(mask_from and mask_from in ln)
        if any(word in ln for word in ("Time", "time", "seconds", " s)", "took")):
            continue
        lines.append(re.sub(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", "#", ln.rstrip()) if masking else ln.rstrip())
    return lines


def test_named_drivers_print_what_the_reference_prints(tmp_path, monkeypatch):
    """`train_all_models.main()` and `compare_models.main()` -- the two drivers the north star names -- against the
    reference's, live, on the same tiny data: the printed lines (timings aside; the PyTorch model's unseeded
    training numbers aside) and the files they leave behind."""
    import difflib
    import pmf_hip
    from oracle_engine import OracleContext
    rng = np.random.default_rng(12)
    n = 900
    u, i = rng.integers(0, 40, n), rng.integers(0, 25, n)
    u[0], i[0] = 39, 24
    r = rng.integers(0, 6, n).astype(float)
    part = rng.choice(3, size=n, p=[0.8, 0.1, 0.1])
    part[0] = 0
    roots = {}
    for who in ("reference", "ours"):
        root = tmp_path / who
        d = root / "data" / "processed"
        d.mkdir(parents=True)
        for k, name in enumerate(("train", "validation", "test")):
            pd.DataFrame({"u": u[part == k], "i": i[part == k], "rating": r[part == k]}).to_csv(d / f"interactions_{name}.csv", index=False)
        (root / "best_hyperparams.txt").write_text(DRIVER_HYPER)
        roots[who] = str(root)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference_mains.py")], env=env,
                          capture_output=True, text=True, timeout=900, cwd=roots["reference"])
    assert done.returncode == 0, done.stderr[-2000:]

    monkeypatch.setattr(pmf_hip, "Context", OracleContext)
    monkeypatch.chdir(roots["ours"])
    from src.experiments import compare_models, train_all_models
    mine = {}
    for name, fn, argv in (("train_all", train_all_models.main, ["train_all_models", "--dataset_mode", "train"]),
                           ("compare", compare_models.main, ["compare_models"])):
        monkeypatch.setattr(sys, "argv", argv)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            fn()
        mine[name] = buf.getvalue()
    for name, marker in (("train_all", ">>> 4/4 Starting HPF (PyTorch)"), ("compare", None)):
        theirs = open(os.path.join(roots["reference"], f"stdout_{name}.txt")).read()
        a, b = _comparable(theirs, marker), _comparable(mine[name], marker)
        if name == "compare":
            # not comparable: the PyTorch section's numbers (unseeded shuffle), the result table's rows (they end in
            # the wall time) and the plot message (this repo's counterpart draws no plots)
            def cut(ls):
                import re
                # the PyTorch section keeps its lines, with the measured numbers masked
                ls = [re.sub(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", "#", ln) if ("HPF PyTorch" in ln or "HPF_PyTorch" in ln or "Epoch" in ln)
                      else ln for ln in ls]
                keep = [ln for ln in ls if "Plots saved" not in ln and not ln[:1].isdigit()]
                while keep and keep[-1] == "":
                    keep.pop()
                return [ln for k, ln in enumerate(keep) if not (ln == "" and k + 1 < len(keep) and "Parameters saved" in keep[k + 1])]
            a, b = cut(a), cut(b)
        delta = "\\n".join(difflib.unified_diff(a, b, "reference", "ours", lineterm="", n=0))
        assert a == b, f"{name}\\n{delta}"
    # the metadata file compare_models leaves behind (compare_models.py:428-433)
    assert open(os.path.join(roots["ours"], "model_comparison_params.txt")).read() == \
        open(os.path.join(roots["reference"], "model_comparison_params.txt")).read()


def test_tuner_writes_the_file_the_reference_writes(tmp_path, monkeypatch):
    """`tune_all_models.main()` on both sides, one trial per model: the reference samples its grids with an unseeded
    `random.choice`, so the VALUES differ by construction -- the file's form must not: the two header lines, one
    line per model in the same order, and per model the same keys in the same order with values of the same types
    (the drivers build `Config(**dict)` from these lines)."""
    import ast
    import pmf_hip
    from oracle_engine import OracleContext
    rng = np.random.default_rng(13)
    n = 700
    u, i = rng.integers(0, 30, n), rng.integers(0, 20, n)
    u[0], i[0] = 29, 19
    r = rng.integers(0, 6, n).astype(float)
    part = rng.choice(3, size=n, p=[0.8, 0.1, 0.1])
    part[0] = 0
    roots = {}
    for who in ("reference", "ours"):
        d = tmp_path / who / "data" / "processed"
        d.mkdir(parents=True)
        for k, name in enumerate(("train", "validation", "test")):
            pd.DataFrame({"u": u[part == k], "i": i[part == k], "rating": r[part == k]}).to_csv(d / f"interactions_{name}.csv", index=False)
        roots[who] = str(tmp_path / who)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference_tuner.py")], env=env,
                          capture_output=True, text=True, timeout=900, cwd=roots["reference"])
    assert done.returncode == 0, done.stderr[-2000:]
    monkeypatch.setattr(pmf_hip, "Context", OracleContext)
    monkeypatch.chdir(roots["ours"])
    monkeypatch.setattr(sys, "argv", ["tune_all_models", "--n_trials", "1"])
    from src.experiments import tune_all_models
    captured = io.StringIO()
    with contextlib.redirect_stdout(captured):
        tune_all_models.main()
    import difflib
    import re

    def shape_of(text):      # every number masked: the sampled configurations and their scores differ by construction
        return [re.sub(r"[-+]?\d+(?:\.\d+)?(?:e[-+]?\d+)?", "#", ln.rstrip()) for ln in text.splitlines()
                if not any(w in ln for w in ("Time", "time", "seconds"))]
    said_theirs = shape_of(open(os.path.join(roots["reference"], "stdout_tune.txt")).read())
    said_mine = shape_of(captured.getvalue())
    delta = "\n".join(difflib.unified_diff(said_theirs, said_mine, "reference", "ours", lineterm="", n=0))
    assert said_mine == said_theirs, delta
    theirs = open(os.path.join(roots["reference"], "best_hyperparams.txt")).read().splitlines()
    mine = open(os.path.join(roots["ours"], "best_hyperparams.txt")).read().splitlines()
    assert mine[:2] == theirs[:2] == ["BEST CONFIGURATIONS", "==================="]
    assert len(mine) == len(theirs) == 6
    for a, b in zip(mine[2:], theirs[2:]):
        (name_a, text_a), (name_b, text_b) = a.split(":", 1), b.split(":", 1)
        da, db = ast.literal_eval(text_a.strip()), ast.literal_eval(text_b.strip())
        assert name_a == name_b and list(da) == list(db), (a, b)
        assert [type(v) for v in da.values()] == [type(v) for v in db.values()], (a, b)


def test_torch_grid_search_prints_the_reference_protocol(tmp_path, monkeypatch):
    """`tune_hpf_pytorch.run_tuning()` on both sides (the full 16-combination grid, 10 epochs each, tiny data): the
    same protocol lines in the same order; the RMSE values are not pinned (unseeded shuffles / initialisation)."""
    rng = np.random.default_rng(14)
    n = 400
    u, i = rng.integers(0, 25, n), rng.integers(0, 15, n)
    r = rng.integers(0, 6, n).astype(float)
    part = rng.choice(3, size=n, p=[0.8, 0.1, 0.1])
    roots = {}
    for who in ("reference", "ours"):
        d = tmp_path / who / "data" / "processed"
        d.mkdir(parents=True)
        for k, name in enumerate(("train", "validation", "test")):
            pd.DataFrame({"u": u[part == k], "i": i[part == k], "rating": r[part == k]}).to_csv(d / f"interactions_{name}.csv", index=False)
        roots[who] = str(tmp_path / who)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "live_reference_tune_torch.py")], env=env,
                          capture_output=True, text=True, timeout=900, cwd=roots["reference"])
    assert done.returncode == 0, done.stderr[-2000:]
    monkeypatch.chdir(roots["ours"])
    from src.experiments import tune_hpf_pytorch
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tune_hpf_pytorch.run_tuning()

    def protocol(text):      # the lines that do not carry a measured value
        return [ln for ln in text.splitlines() if ln.strip() and not ln.startswith(("Result RMSE", "*** New Best", "Best Validation",
                                                                                     "Best Configuration"))]
    theirs = protocol(open(os.path.join(roots["reference"], "stdout_tune_torch.txt")).read())
    assert protocol(buf.getvalue()) == theirs and len(theirs) == 2 + 16
    for text in (buf.getvalue(), open(os.path.join(roots["reference"], "stdout_tune_torch.txt")).read()):
        assert text.count("Result RMSE: ") == 16 and "Best Configuration: {" in text and "Best Validation RMSE: " in text
