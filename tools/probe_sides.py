"""Per-side launch time of the Gaussian factor half-sweep at C2 (user side gathers the 832 MB item
table, item side the 8.3 GB user table)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd")]
import pmf_hip  # noqa: E402
from pmf_hip import ARR_BIAS, ARR_FACTOR, ITEM, USER  # noqa: E402
from pmf_hip.synth import BASE_SEED, synth_ratings  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
I = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000      # item table = I x 8.5 KB at K=64 (Infinity Cache: 256 MB)
U, N = 1_000_000, 50_000_000
u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
ctx = pmf_hip.Context(U, I, K)
rng = np.random.default_rng(1)
ctx.set_ratings(u, i, r - r.mean())
ctx.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K)))
ctx.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))
for _ in range(2):
    ctx.gauss_factor_sweep(USER, 0.5, 1.0); ctx.gauss_factor_sweep(ITEM, 0.5, 1.0)
ctx.sync()
for side, name in ((USER, "user sweep (gathers item rows)"), (ITEM, "item sweep (gathers user rows)")):
    ts = []
    for _ in range(5):
        ctx.sync(); t0 = time.perf_counter()
        ctx.gauss_factor_sweep(side, 0.5, 1.0)
        ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.gauss_factor_sweep(1 - side, 0.5, 1.0)
    print(f"K={K} I={I} {name}: {min(ts):.2f} ms (min of 5), median {sorted(ts)[2]:.2f} ms", flush=True)
