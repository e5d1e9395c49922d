"""Whole `fit` calls at BASELINE config C2 / C3 size through the model classes (the reference's own API):
DataFrames in, `fit(train_df, val_df)` with the per-iteration validation monitor, attributes out.
Wall time of each leg on one MI355X.    python tools/fit_c2.py [--small]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "prob-matrix-factorization_amd")]
import numpy as np, pandas as pd
from pmf_hip.synth import synth_ratings, train_val_split
from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig

small = "--small" in sys.argv
U, I, N = (100_000, 10_000, 5_000_000) if small else (1_000_000, 100_000, 50_000_000)
t0 = time.time()
u, i, r = synth_ratings(U, I, N)
(tu, ti, tr), (vu, vi, vr) = train_val_split(u, i, r)
train = pd.DataFrame({"u": tu, "i": ti, "rating": tr})
val = pd.DataFrame({"u": vu, "i": vi, "rating": vr})
print(f"data: {len(train)} train / {len(val)} val ratings, {time.time() - t0:.1f} s to generate", flush=True)


def report(name, model, t_fit, iters):
    sec = model.history_["seconds"]
    t0 = time.time()
    shape = tuple(np.asarray(model.m_theta if hasattr(model, "m_theta") else model.E_theta).shape)
    t_pull = time.time() - t0
    print(f"{name}: fit {t_fit:.2f} s for {iters} iterations incl. validation each "
          f"(setup {t_fit - sum(sec):.2f} s = frame -> arrays, initial state, upload, index build; iterations "
          f"{np.mean(sec[1:]) * 1e3:.1f} ms each after the first {sec[0] * 1e3:.0f} ms); user factors {shape} to host {t_pull:.2f} s; "
          f"val RMSE {model.history_['val_rmse'][-1]:.4f}", flush=True)


# HPF (compare_models.py:180-185 shifts the ratings by +1)
tr1, va1 = train.assign(rating=train["rating"] + 1.0), val.assign(rating=val["rating"] + 1.0)
m = HPF_CAVI(HPF_CAVI_Config(n_factors=64, a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0, max_iter=20,
                             tol=None, random_state=42, verbose=False))
t0 = time.time(); m.fit(tr1, va1); report("HPF_CAVI K=64", m, time.time() - t0, 20)
del m
# Gaussian (compare_models.py:54-65 centres by the train mean)
mu = float(train["rating"].mean())
trc, vac = train.assign(rating=train["rating"] - mu), val.assign(rating=val["rating"] - mu)
g = GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=64, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0, max_iter=10,
                                        tol=-1e9, random_state=42, verbose=False))
t0 = time.time(); g.fit(trc, vac, global_mean=mu); report("GaussianMFCAVI K=64", g, time.time() - t0, 10)
