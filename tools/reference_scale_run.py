"""End-to-end `compare_models` run at the reference data's shape (11,780 users x 13,000 items,
~271k / 23.6k / 11.8k train / val / test ratings -- synthetic stand-in, the real CSVs are not
obtainable offline) with the reference's best_hyperparams.txt, to put wall-clock `fit` times next to
the reference's published ones."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "prob-matrix-factorization_amd")]
import numpy as np, pandas as pd
from pmf_hip.synth import synth_ratings

HYPER = """BEST CONFIGURATIONS
===================
GaussianMF: {'n_factors': 30, 'sigma2': 0.3, 'eta_theta2': 0.5, 'eta_beta2': 0.5, 'eta_bias2': 1.0, 'max_iter': 100, 'tol': 0.001, 'random_state': 42, 'verbose': False}
PoissonMF: {'n_factors': 40, 'a0': 0.1, 'b0': 0.5, 'max_iter': 150, 'tol': None, 'random_state': 42, 'verbose': False}
HPF_CAVI: {'n_factors': 20, 'a': 0.3, 'a_prime': 5.0, 'b_prime': 5.0, 'c': 0.3, 'c_prime': 5.0, 'd_prime': 5.0, 'max_iter': 100, 'tol': None, 'random_state': 42, 'verbose': False}
HPF_PyTorch: {'n_factors': 10, 'a': 1.0, 'a_prime': 1.0, 'b_prime': 1.0, 'c': 1.0, 'c_prime': 1.0, 'd_prime': 1.0, 'lr': 0.0005, 'batch_size': 1024, 'epochs': 50, 'device': 'cpu', 'verbose': False}
"""
work = tempfile.mkdtemp()
os.chdir(work)
os.makedirs("data/processed")
u, i, r = synth_ratings(11_780, 13_000, 306_400, seed=5)
u[0], i[0] = 11_779, 12_999
part = np.random.default_rng(0).choice(3, size=len(u), p=[0.8845, 0.0770, 0.0385]); part[0] = 0
for k, name in enumerate(("train", "validation", "test")):
    s = part == k
    pd.DataFrame({"u": u[s], "i": i[s], "rating": r[s]}).to_csv(f"data/processed/interactions_{name}.csv", index=False)
open("best_hyperparams.txt", "w").write(HYPER)
from src.experiments import compare_models
compare_models.main()
print(open("model_comparison_results.csv").read())
